// GroupNorm (+ fused ReLU) on the native [M, C] lattice-value layout, forward and backward.
// The reference runs torch.nn.GroupNorm on a transposed [1, C, M] view (lattice_modules.py:585-616), which on a
// row-major value matrix costs two transposed copies in each direction; the LNN blocks (GnRelu1x1, GnReluConv,
// GnReluFinefy, ...) apply it before every operator, so it sits between any two lattice kernels of the U-Net.
//   statistics : per-channel partial sums over a slab of rows per workgroup (lanes run along the channels of a row:
//                coalesced), combined across workgroups with one fp64 atomic per (workgroup, channel)
//   apply      : every workgroup rebuilds the per-channel scale/shift from the C channel sums in LDS, then one
//                float4 pass over its slab
// Backward uses the same two shapes: per-channel sums of gy*x and gy, then dx = gy*gamma*rstd + x*c2[g] + c3[g].
#include "ln_common.h"

#define LN_GN_MAX_C 1024
#define LN_GN_PASSES 16
#define LN_GN_REPLICAS 32  // accumulator copies: memory-side atomics serialise per cache line, so spread the workgroups

// acc[c*2 + 0] += sum_rows p(row, c), acc[c*2 + 1] += sum_rows q(row, c)
//   forward  (gy == nullptr): p = x,  q = x*x
//   backward               : p = gy' * x, q = gy'   with gy' = gy masked by (x*a[c] + b[c] > 0) when relu
// A thread owns one float4 (4 channels) of a row; 256 / (c/4) rows are read per pass and LN_GN_PASSES independent
// passes are in flight per thread.
__global__ void __launch_bounds__(256)
    k_gn_stats(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ scale_shift, int relu, int m, int c,
               double* __restrict__ acc, const int* __restrict__ rows_dev) {
    if (rows_dev) m = min(m, *rows_dev);  // static-rows mode: the tensors are taller than the lattice, only its rows count
    __shared__ float4 s_p[256], s_q[256];
    const int tid = threadIdx.x;
    const int quads = c >> 2;                 // c % 4 == 0, c <= 1024  ->  quads <= 256
    const int rows_per_pass = 256 / quads;
    const int rp = tid / quads;
    const int qi = tid - rp * quads;
    const bool live = rp < rows_per_pass;
    const long long r0 = (long long)blockIdx.x * rows_per_pass * LN_GN_PASSES;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
    if (live) {
        float4 a = p, b = p;
        if (gy && relu) {
            a = reinterpret_cast<const float4*>(scale_shift)[qi];
            b = reinterpret_cast<const float4*>(scale_shift + c)[qi];
        }
        float4 xv[LN_GN_PASSES], gv[LN_GN_PASSES];
#pragma unroll
        for (int k = 0; k < LN_GN_PASSES; ++k) {
            const long long row = r0 + (long long)k * rows_per_pass + rp;
            const bool ok = row < m;
            xv[k] = ok ? reinterpret_cast<const float4*>(x + row * c)[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy) gv[k] = ok ? reinterpret_cast<const float4*>(gy + row * c)[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < LN_GN_PASSES; ++k) {
            if (gy) {
                float4 g = gv[k];
                if (relu) {
                    if (!(xv[k].x * a.x + b.x > 0.f)) g.x = 0.f;
                    if (!(xv[k].y * a.y + b.y > 0.f)) g.y = 0.f;
                    if (!(xv[k].z * a.z + b.z > 0.f)) g.z = 0.f;
                    if (!(xv[k].w * a.w + b.w > 0.f)) g.w = 0.f;
                }
                p.x += g.x * xv[k].x; p.y += g.y * xv[k].y; p.z += g.z * xv[k].z; p.w += g.w * xv[k].w;
                q.x += g.x; q.y += g.y; q.z += g.z; q.w += g.w;
            } else {
                p.x += xv[k].x; p.y += xv[k].y; p.z += xv[k].z; p.w += xv[k].w;
                q.x += xv[k].x * xv[k].x; q.y += xv[k].y * xv[k].y; q.z += xv[k].z * xv[k].z; q.w += xv[k].w * xv[k].w;
            }
        }
    }
    s_p[tid] = p;
    s_q[tid] = q;
    __syncthreads();
    if (live && rp == 0) {  // fold the threads that share a channel quad, then one fp64 atomic per channel sum
        for (int r = 1; r < rows_per_pass; ++r) {
            const float4 pp = s_p[r * quads + qi], qq = s_q[r * quads + qi];
            p.x += pp.x; p.y += pp.y; p.z += pp.z; p.w += pp.w;
            q.x += qq.x; q.y += qq.y; q.z += qq.z; q.w += qq.w;
        }
        double* dst = acc + (size_t)(blockIdx.x % LN_GN_REPLICAS) * 2 * c + 8 * qi;
        atomicAdd(dst + 0, (double)p.x); atomicAdd(dst + 1, (double)q.x);
        atomicAdd(dst + 2, (double)p.y); atomicAdd(dst + 3, (double)q.y);
        atomicAdd(dst + 4, (double)p.z); atomicAdd(dst + 5, (double)q.z);
        atomicAdd(dst + 6, (double)p.w); atomicAdd(dst + 7, (double)q.w);
    }
}

// per-channel scale a[c] = gamma*rstd[g], shift b[c] = beta - mean[g]*a[c] from the channel sums; block 0 also
// publishes mean/rstd per group and scale/shift per channel for the backward pass
__device__ __forceinline__ void ln_gn_channel_affine(const double* __restrict__ acc, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int m, int c, int groups, float eps, float* s_a,
                                                     float* s_b, float* __restrict__ mean_rstd, float* __restrict__ scale_shift) {
    if (m < 1) m = 1;  // (an empty lattice under a static row bound: all sums are zero)
    const int cg = c / groups;
    __shared__ double s_sum[LN_GN_MAX_C], s_sq[LN_GN_MAX_C];
    for (int col = threadIdx.x; col < c; col += 256) {  // channel sums over the accumulator replicas
        double v0[LN_GN_REPLICAS], v1[LN_GN_REPLICAS];
#pragma unroll
        for (int r = 0; r < LN_GN_REPLICAS; ++r) {
            v0[r] = acc[(size_t)r * 2 * c + 2 * col];
            v1[r] = acc[(size_t)r * 2 * c + 2 * col + 1];
        }
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int r = 0; r < LN_GN_REPLICAS; ++r) {
            a0 += v0[r];
            a1 += v1[r];
        }
        s_sum[col] = a0;
        s_sq[col] = a1;
    }
    __syncthreads();
    for (int col = threadIdx.x; col < c; col += 256) {
        const int g = col / cg;
        double s = 0.0, ss = 0.0;
        for (int k = 0; k < cg; ++k) {
            s += s_sum[g * cg + k];
            ss += s_sq[g * cg + k];
        }
        const double cnt = (double)m * cg;
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float a = (gamma ? gamma[col] : 1.f) * rstd;
        const float b = (beta ? beta[col] : 0.f) - (float)mean * a;
        s_a[col] = a;
        s_b[col] = b;
        if (blockIdx.x == 0) {
            scale_shift[col] = a;
            scale_shift[c + col] = b;
            if (col == g * cg) {
                mean_rstd[g] = (float)mean;
                mean_rstd[groups + g] = rstd;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
    k_gn_apply(const float* __restrict__ x, const double* __restrict__ acc, const float* __restrict__ gamma, const float* __restrict__ beta,
               int m, int c, int groups, float eps, int relu, float* __restrict__ y, float* __restrict__ mean_rstd,
               float* __restrict__ scale_shift, double* __restrict__ zero_next, int zero_count, const int* __restrict__ rows_dev) {
    __shared__ float s_a[LN_GN_MAX_C], s_b[LN_GN_MAX_C];
    const int m_tensor = m;
    if (rows_dev) m = min(m, *rows_dev);  // rows beyond the lattice: excluded from the statistics, written as zeros
    if (zero_next && blockIdx.x == 0)  // the accumulators of the NEXT call (nobody is using them now: stream order)
        for (int i = threadIdx.x; i < zero_count; i += 256) zero_next[i] = 0.0;
    ln_gn_channel_affine(acc, gamma, beta, m, c, groups, eps, s_a, s_b, mean_rstd, scale_shift);
    __syncthreads();
    const long long total4 = (long long)m * c / 4;  // c % 4 == 0 checked by the host
    const long long tensor4 = (long long)m_tensor * c / 4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = total4 + (long long)blockIdx.x * 256 + threadIdx.x; i < tensor4; i += stride)
        reinterpret_cast<float4*>(y)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const int col = int((i * 4) % c);
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = v.x * s_a[col] + s_b[col];
        v.y = v.y * s_a[col + 1] + s_b[col + 1];
        v.z = v.z * s_a[col + 2] + s_b[col + 2];
        v.w = v.w * s_a[col + 3] + s_b[col + 3];
        if (relu) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

// dx = gy' * gamma * rstd + x * c2[g] + c3[g];   block 0 writes dgamma = (ds - db*mean)*rstd, dbeta = db
__global__ void __launch_bounds__(256)
    k_gn_backward_apply(const float* __restrict__ x, const float* __restrict__ gy, const double* __restrict__ acc,
                        const float* __restrict__ gamma, const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift, int m,
                        int c, int groups, int relu, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                        double* __restrict__ zero_next, int zero_count, const int* __restrict__ rows_dev) {
    const int m_tensor = m;
    if (rows_dev) m = min(m, *rows_dev);
    if (m < 1) m = 1;
    if (zero_next && blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_count; i += 256) zero_next[i] = 0.0;
    __shared__ float s_gr[LN_GN_MAX_C], s_c2[LN_GN_MAX_C], s_c3[LN_GN_MAX_C], s_a[LN_GN_MAX_C], s_b[LN_GN_MAX_C];
    const int cg = c / groups;
    __shared__ double s_ds[LN_GN_MAX_C], s_db[LN_GN_MAX_C];
    for (int col = threadIdx.x; col < c; col += 256) {  // channel sums over the accumulator replicas
        double v0[LN_GN_REPLICAS], v1[LN_GN_REPLICAS];
#pragma unroll
        for (int r = 0; r < LN_GN_REPLICAS; ++r) {
            v0[r] = acc[(size_t)r * 2 * c + 2 * col];
            v1[r] = acc[(size_t)r * 2 * c + 2 * col + 1];
        }
        double ds = 0.0, db = 0.0;
#pragma unroll
        for (int r = 0; r < LN_GN_REPLICAS; ++r) {
            ds += v0[r];
            db += v1[r];
        }
        s_ds[col] = ds;
        s_db[col] = db;
    }
    __syncthreads();
    for (int col = threadIdx.x; col < c; col += 256) {
        const int g = col / cg;
        const float mean = mean_rstd[g], rstd = mean_rstd[groups + g];
        double sum1 = 0.0, sum2 = 0.0;  // sum over the group's channels of ds*gamma, db*gamma
        for (int k = 0; k < cg; ++k) {
            const int cc = g * cg + k;
            const double gm = gamma ? (double)gamma[cc] : 1.0;
            sum1 += s_ds[cc] * gm;
            sum2 += s_db[cc] * gm;
        }
        const double cnt = (double)m * cg;
        const double c2 = (sum2 * mean - sum1) * (double)rstd * rstd * rstd / cnt;
        const double c3 = -c2 * mean - sum2 * (double)rstd / cnt;
        s_gr[col] = (gamma ? gamma[col] : 1.f) * rstd;
        s_c2[col] = (float)c2;
        s_c3[col] = (float)c3;
        s_a[col] = scale_shift[col];
        s_b[col] = scale_shift[c + col];
        if (blockIdx.x == 0) {
            if (dgamma) dgamma[col] = (float)((s_ds[col] - s_db[col] * mean) * rstd);
            if (dbeta) dbeta[col] = (float)s_db[col];
        }
    }
    __syncthreads();
    const long long total4 = (rows_dev ? (long long)min(m_tensor, *rows_dev) : (long long)m) * c / 4;
    const long long tensor4 = (long long)m_tensor * c / 4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = total4 + (long long)blockIdx.x * 256 + threadIdx.x; i < tensor4; i += stride)
        reinterpret_cast<float4*>(dx)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const int col = int((i * 4) % c);
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 g = reinterpret_cast<const float4*>(gy)[i];
        if (relu) {
            if (!(xv.x * s_a[col] + s_b[col] > 0.f)) g.x = 0.f;
            if (!(xv.y * s_a[col + 1] + s_b[col + 1] > 0.f)) g.y = 0.f;
            if (!(xv.z * s_a[col + 2] + s_b[col + 2] > 0.f)) g.z = 0.f;
            if (!(xv.w * s_a[col + 3] + s_b[col + 3] > 0.f)) g.w = 0.f;
        }
        float4 o;
        o.x = g.x * s_gr[col] + xv.x * s_c2[col] + s_c3[col];
        o.y = g.y * s_gr[col + 1] + xv.y * s_c2[col + 1] + s_c3[col + 1];
        o.z = g.z * s_gr[col + 2] + xv.z * s_c2[col + 2] + s_c3[col + 2];
        o.w = g.w * s_gr[col + 3] + xv.w * s_c2[col + 3] + s_c3[col + 3];
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

static int ln_gn_check(const char* who, int m, int c, int groups) {
    LN_REQUIRE(m >= 1 && c >= 4 && c <= LN_GN_MAX_C && c % 4 == 0, LN_ERR_UNSUPPORTED, "%s: need 1 <= rows, channels %% 4 == 0 and <= %d (got %d x %d)",
               who, LN_GN_MAX_C, m, c);
    LN_REQUIRE(groups >= 1 && c % groups == 0, LN_ERR_ARG, "%s: %d groups do not divide %d channels", who, groups, c);
    return LN_OK;
}

static int ln_gn_stats_grid(int m, int c) {
    const int rows_per_pass = 256 / (c / 4);
    return ln_div_up(m, rows_per_pass * LN_GN_PASSES);
}

static int ln_gn_apply_grid(int m, int c) {
    const long long total4 = (long long)m * c / 4;
    int grid = ln_div_up(total4, 256 * 4);
    if (grid > 2048) grid = 2048;
    return grid < 1 ? 1 : grid;
}

extern "C" size_t ln_group_norm_workspace_bytes(int channels) { return (size_t)LN_GN_REPLICAS * 2 * channels * sizeof(double); }

extern "C" int ln_group_norm_forward_rows(const float* x, const float* gamma, const float* beta, int m, int channels, int groups, float eps,
                                          int relu, float* y, float* mean_rstd, float* scale_shift, void* workspace, size_t workspace_bytes,
                                          void* next_workspace, size_t next_workspace_bytes, const int* rows_device, void* stream);
extern "C" int ln_group_norm_forward(const float* x, const float* gamma, const float* beta, int m, int channels, int groups, float eps,
                                     int relu, float* y, float* mean_rstd, float* scale_shift, void* workspace, size_t workspace_bytes,
                                     void* next_workspace, size_t next_workspace_bytes, void* stream) {
    return ln_group_norm_forward_rows(x, gamma, beta, m, channels, groups, eps, relu, y, mean_rstd, scale_shift, workspace, workspace_bytes,
                                      next_workspace, next_workspace_bytes, nullptr, stream);
}
extern "C" int ln_group_norm_forward_rows(const float* x, const float* gamma, const float* beta, int m, int channels, int groups, float eps,
                                          int relu, float* y, float* mean_rstd, float* scale_shift, void* workspace, size_t workspace_bytes,
                                          void* next_workspace, size_t next_workspace_bytes, const int* rows_device, void* stream) {
    int rc = ln_gn_check("ln_group_norm_forward", m, channels, groups);
    if (rc) return rc;
    LN_REQUIRE(x && y && mean_rstd && scale_shift && workspace && workspace_bytes >= ln_group_norm_workspace_bytes(channels), LN_ERR_ARG,
               "ln_group_norm_forward: null buffer or workspace too small");
    LN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
               LN_ERR_ARG, "ln_group_norm_forward: x / y must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    double* acc = static_cast<double*>(workspace);
    // next_workspace != NULL: the caller alternates two workspaces and promises `workspace` is zero (it was `next_workspace` of
    // the previous call on this stream, or freshly zeroed); this call zero-fills `next_workspace` on its way out.
    if (!next_workspace && ln_zero_async(acc, ln_group_norm_workspace_bytes(channels), st) != LN_OK)
        return ln_check_launch("ln_group_norm_forward(memset)");
    LN_LAUNCH("k_gn_stats", k_gn_stats, dim3(ln_gn_stats_grid(m, channels)), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, 0, m,
              channels, acc, rows_device);
    LN_LAUNCH("k_gn_apply", k_gn_apply, dim3(ln_gn_apply_grid(m, channels)), dim3(256), 0, st, x, acc, gamma, beta, m, channels, groups, eps, relu, y,
              mean_rstd, scale_shift, static_cast<double*>(next_workspace), int(next_workspace_bytes / sizeof(double)), rows_device);
    return ln_check_launch("ln_group_norm_forward");
}

extern "C" int ln_group_norm_backward_rows(const float* x, const float* grad_y, const float* gamma, const float* mean_rstd,
                                           const float* scale_shift, int m, int channels, int groups, int relu, float* grad_x, float* grad_gamma,
                                           float* grad_beta, void* workspace, size_t workspace_bytes, void* next_workspace,
                                           size_t next_workspace_bytes, const int* rows_device, void* stream);
extern "C" int ln_group_norm_backward(const float* x, const float* grad_y, const float* gamma, const float* mean_rstd,
                                      const float* scale_shift, int m, int channels, int groups, int relu, float* grad_x, float* grad_gamma,
                                      float* grad_beta, void* workspace, size_t workspace_bytes, void* next_workspace, size_t next_workspace_bytes, void* stream) {
    return ln_group_norm_backward_rows(x, grad_y, gamma, mean_rstd, scale_shift, m, channels, groups, relu, grad_x, grad_gamma, grad_beta, workspace,
                                       workspace_bytes, next_workspace, next_workspace_bytes, nullptr, stream);
}
extern "C" int ln_group_norm_backward_rows(const float* x, const float* grad_y, const float* gamma, const float* mean_rstd,
                                           const float* scale_shift, int m, int channels, int groups, int relu, float* grad_x, float* grad_gamma,
                                           float* grad_beta, void* workspace, size_t workspace_bytes, void* next_workspace,
                                           size_t next_workspace_bytes, const int* rows_device, void* stream) {
    int rc = ln_gn_check("ln_group_norm_backward", m, channels, groups);
    if (rc) return rc;
    LN_REQUIRE(x && grad_y && mean_rstd && scale_shift && grad_x && workspace && workspace_bytes >= ln_group_norm_workspace_bytes(channels),
               LN_ERR_ARG, "ln_group_norm_backward: null buffer or workspace too small");
    LN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x)) & 15) == 0, LN_ERR_ARG,
               "ln_group_norm_backward: x / grad_y / grad_x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    double* acc = static_cast<double*>(workspace);
    if (!next_workspace && ln_zero_async(acc, ln_group_norm_workspace_bytes(channels), st) != LN_OK)
        return ln_check_launch("ln_group_norm_backward(memset)");
    LN_LAUNCH("k_gn_stats", k_gn_stats, dim3(ln_gn_stats_grid(m, channels)), dim3(256), 0, st, x, grad_y, scale_shift, relu, m, channels, acc, rows_device);
    LN_LAUNCH("k_gn_backward_apply", k_gn_backward_apply, dim3(ln_gn_apply_grid(m, channels)), dim3(256), 0, st, x, grad_y, acc, gamma, mean_rstd,
              scale_shift, m, channels, groups, relu, grad_x, grad_gamma, grad_beta, static_cast<double*>(next_workspace),
              int(next_workspace_bytes / sizeof(double)), rows_device);
    return ln_check_launch("ln_group_norm_backward");
}
