// Launch diet of the glue around the lattice operators in a training step: pieces the reference writes as
// chains of torch broadcasting operators on tiny or token-sized tensors, each a launch of its own.
//
//  * weight normalisation (utils.py:72-158 weight_norm_wrapper with v_dim=None, used by LinearWN / ConvLatticeIm2RowWN / ...):
//        w = v * g / ||v||_F                   (g: one magnitude per row or per column of v)
//    forward = norm + div + mul, backward ~10 elementwise / reduction launches on a parameter of a few thousand numbers.
//    Here: one workgroup forward, one workgroup backward, sums in a fixed order (deterministic).
//        grad_g[j] = sum_k gw[j,k] v[j,k] / n
//        grad_v    = gw * g / n  -  v * (sum_j g[j] sum_k gw[j,k] v[j,k]) / n^3
//
//  * DistributeLatticeModule (lattice_modules.py:72-94): per token, positions minus the mean position of the token's vertex, rows
//    of the tokens of vertex 0 (the "invalid" bucket, also every token that found no vertex) zeroed:
//        out[t, :D] = d[t, :D] - sums[idx[t]] / max(count[idx[t]], 1),  out[t, D:] = d[t, D:]     (idx[t] > 0; zeros otherwise)
//    in one pass instead of div / clamp / index_select / sub / cat / eq / masked_fill.
#include "ln_common.h"

#define LN_WN_THREADS 1024
#define LN_WN_MAX_G 1024

// element (j, k): j indexes g, k the other dimension of v [R, C]
//   g_dim == 0 (g per row):    address j * C + k, K = C
//   g_dim == 1 (g per column): address k * C + j, K = R
struct LnWnShape {
    int J, K, sj, sk;
};

__device__ __forceinline__ float ln_wn_block_sum(float x, float* s_red) {
    // fixed tree over the 1024 threads: wave sums by DPP-free shuffles, then 16 wave totals in LDS
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // s_red may still be read from a previous call
    if ((threadIdx.x & 63) == 0) s_red[wave] = x;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < LN_WN_THREADS / 64; ++w) t += s_red[w];
    return t;
}

__global__ void __launch_bounds__(LN_WN_THREADS)
    k_weight_norm_forward(const float* __restrict__ v, const float* __restrict__ g, LnWnShape s, float* __restrict__ w,
                          float* __restrict__ norm_out) {
    __shared__ float s_red[LN_WN_THREADS / 64];
    const long long total = (long long)s.J * s.K;
    float acc = 0.f;
    for (long long i = threadIdx.x; i < total; i += LN_WN_THREADS) {
        const float x = v[i];
        acc += x * x;
    }
    const float n = sqrtf(ln_wn_block_sum(acc, s_red));
    if (threadIdx.x == 0) norm_out[0] = n;
    // plain [R, C] walk: i = row * C + col; j = row (g_dim 0: sj = C) or col (g_dim 1: sj = 1)
    const int C = s.sj == 1 ? s.J : s.K;
    for (long long i = threadIdx.x; i < total; i += LN_WN_THREADS) {
        const int row = int(i / C), col = int(i - (long long)row * C);
        const int j = s.sj == 1 ? col : row;
        w[i] = v[i] * (g[j] / n);
    }
}

__global__ void __launch_bounds__(LN_WN_THREADS)
    k_weight_norm_backward(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ gw,
                           const float* __restrict__ norm, LnWnShape s, float* __restrict__ gv, float* __restrict__ gg) {
    __shared__ float s_part[LN_WN_THREADS];
    __shared__ float s_dot[LN_WN_MAX_G];
    __shared__ float s_red[LN_WN_THREADS / 64];
    const float n = norm[0];
    const int lanes = LN_WN_THREADS / s.J;  // threads per g entry
    const int kl = threadIdx.x / s.J;
    const int j = threadIdx.x - kl * s.J;
    float acc = 0.f;
    if (kl < lanes)
        for (int k = kl; k < s.K; k += lanes) {
            const long long a = (long long)j * s.sj + (long long)k * s.sk;
            acc += gw[a] * v[a];
        }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    float mine = 0.f;
    if (threadIdx.x < s.J) {
        float d = 0.f;
        for (int l = 0; l < lanes; ++l) d += s_part[l * s.J + threadIdx.x];
        s_dot[threadIdx.x] = d;
        gg[threadIdx.x] = d / n;
        mine = d * g[threadIdx.x];
    }
    const float t = ln_wn_block_sum(mine, s_red);  // sum_j g[j] * dot[j]
    const float dn_over_n = -t / (n * n * n);       // (dL/dn) / n
    const long long total = (long long)s.J * s.K;
    const int C = s.sj == 1 ? s.J : s.K;
    for (long long i = threadIdx.x; i < total; i += LN_WN_THREADS) {
        const int row = int(i / C), col = int(i - (long long)row * C);
        const int jj = s.sj == 1 ? col : row;
        gv[i] = gw[i] * (g[jj] / n) + v[i] * dn_over_n;
    }
}

static int ln_wn_shape(const char* who, int rows, int cols, int g_dim, LnWnShape& s) {
    LN_REQUIRE(rows >= 1 && cols >= 1 && (g_dim == 0 || g_dim == 1), LN_ERR_ARG, "%s: bad sizes", who);
    s = g_dim == 0 ? LnWnShape{rows, cols, cols, 1} : LnWnShape{cols, rows, 1, cols};
    LN_REQUIRE(s.J <= LN_WN_MAX_G, LN_ERR_UNSUPPORTED, "%s: at most %d magnitudes (got %d)", who, LN_WN_MAX_G, s.J);
    LN_REQUIRE((long long)rows * cols <= (1ll << 24), LN_ERR_UNSUPPORTED, "%s: parameter too large for the one-workgroup form", who);
    return LN_OK;
}

extern "C" int ln_weight_norm_forward(const float* v, const float* g, int rows, int cols, int g_dim, float* w, float* norm, void* stream) {
    LnWnShape s;
    int rc = ln_wn_shape("ln_weight_norm_forward", rows, cols, g_dim, s);
    if (rc) return rc;
    LN_REQUIRE(v && g && w && norm, LN_ERR_ARG, "ln_weight_norm_forward: null buffer");
    LN_LAUNCH("k_weight_norm_forward", k_weight_norm_forward, dim3(1), dim3(LN_WN_THREADS), 0, (hipStream_t)stream, v, g, s, w, norm);
    return ln_check_launch("ln_weight_norm_forward");
}

extern "C" int ln_weight_norm_backward(const float* v, const float* g, const float* grad_w, const float* norm, int rows, int cols, int g_dim,
                                       float* grad_v, float* grad_g, void* stream) {
    LnWnShape s;
    int rc = ln_wn_shape("ln_weight_norm_backward", rows, cols, g_dim, s);
    if (rc) return rc;
    LN_REQUIRE(v && g && grad_w && norm && grad_v && grad_g, LN_ERR_ARG, "ln_weight_norm_backward: null buffer");
    LN_LAUNCH("k_weight_norm_backward", k_weight_norm_backward, dim3(1), dim3(LN_WN_THREADS), 0, (hipStream_t)stream, v, g, grad_w, norm, s,
              grad_v, grad_g);
    return ln_check_launch("ln_weight_norm_backward");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// thread = (token, column); 256 threads walk consecutive elements of out (coalesced); the per-vertex rows come from L2
__global__ void __launch_bounds__(256)
    k_distribute_centre(const float* __restrict__ d, const int* __restrict__ idx, const float* __restrict__ sums,
                        const int* __restrict__ counts, long long tokens, int width, int pos_dim, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= tokens * width) return;
    const long long t = i / width;
    const int c = int(i - t * width);
    const int row = idx[t];
    float x = 0.f;
    if (row > 0) {
        x = d[i];
        if (c < pos_dim) x -= sums[(size_t)row * pos_dim + c] / (float)max(counts[row], 1);
    }
    out[i] = x;
}

extern "C" int ln_distribute_centre(const float* distributed, const int* splat_idx, const float* position_sums, const int* counts,
                                    long long tokens, int width, int pos_dim, float* out, void* stream) {
    LN_REQUIRE(tokens >= 0 && width >= 1 && pos_dim >= 0 && pos_dim <= width, LN_ERR_ARG, "ln_distribute_centre: bad sizes");
    if (tokens == 0) return LN_OK;
    LN_REQUIRE(distributed && splat_idx && position_sums && counts && out, LN_ERR_ARG, "ln_distribute_centre: null buffer");
    LN_LAUNCH("k_distribute_centre", k_distribute_centre, dim3(ln_div_up(tokens * width, 256)), dim3(256), 0, (hipStream_t)stream, distributed,
              splat_idx, position_sums, counts, tokens, width, pos_dim, out);
    return ln_check_launch("ln_distribute_centre");
}
