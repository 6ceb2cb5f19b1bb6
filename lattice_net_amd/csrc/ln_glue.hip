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
    int C, per_col;  // row width of v [R, C] and "g runs along the columns" — stated, not inferred from the strides (sj == 1 also holds for
                     // g_dim 0 with one column)
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
    // plain [R, C] walk: i = row * C + col; j = row (g_dim 0) or col (g_dim 1)
    const int C = s.C;
    for (long long i = threadIdx.x; i < total; i += LN_WN_THREADS) {
        const int row = int(i / C), col = int(i - (long long)row * C);
        const int j = s.per_col ? col : row;
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
    const int C = s.C;
    for (long long i = threadIdx.x; i < total; i += LN_WN_THREADS) {
        const int row = int(i / C), col = int(i - (long long)row * C);
        const int jj = s.per_col ? col : row;
        gv[i] = gw[i] * (g[jj] / n) + v[i] * dn_over_n;
    }
}

static int ln_wn_shape(const char* who, int rows, int cols, int g_dim, LnWnShape& s) {
    LN_REQUIRE(rows >= 1 && cols >= 1 && (g_dim == 0 || g_dim == 1), LN_ERR_ARG, "%s: bad sizes", who);
    s = g_dim == 0 ? LnWnShape{rows, cols, cols, 1, cols, 0} : LnWnShape{cols, rows, 1, cols, cols, 1};
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Mean negative log-likelihood of the training loop (ln_train.py:130 torch.nn.NLLLoss(ignore_index=...)) over log-probabilities
// [n, C] and labels [n]:  loss = -sum_i [y_i != ignore] lp[i, y_i] / max(#{y_i != ignore}, 1).
// torch's own nll_loss reduces in one workgroup (0.14 ms at n = 120 k); written with gather / mean / their autograd it is ten
// launches.  Here: per-workgroup partial sums in a fixed order + one finishing workgroup forward, one fully written [n, C]
// gradient pass backward (no zero fill, no scatter).
#define LN_NLL_BLOCKS 512

__global__ void __launch_bounds__(256)
    k_nll_partials(const float* __restrict__ lp, const long long* __restrict__ target, long long n, int C, long long ignore_index,
                   float* __restrict__ partial) {
    __shared__ float s_sum[4], s_cnt[4];
    float acc = 0.f, cnt = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t = target[i];
        if (t != ignore_index) {
            const long long tc = t < 0 ? 0 : (t >= C ? C - 1 : t);  // (the torch formulation clamps out-of-range labels)
            acc += lp[i * C + tc];
            cnt += 1.f;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_down(acc, off, 64);
        cnt += __shfl_down(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_sum[threadIdx.x >> 6] = acc;
        s_cnt[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        partial[2 * blockIdx.x + 1] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
    }
}

__global__ void __launch_bounds__(LN_NLL_BLOCKS)
    k_nll_finish(const float* __restrict__ partial, int blocks, float* __restrict__ loss_count) {
    __shared__ float s_sum[LN_NLL_BLOCKS / 64], s_cnt[LN_NLL_BLOCKS / 64];
    float acc = threadIdx.x < blocks ? partial[2 * threadIdx.x] : 0.f;
    float cnt = threadIdx.x < blocks ? partial[2 * threadIdx.x + 1] : 0.f;
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_down(acc, off, 64);
        cnt += __shfl_down(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_sum[threadIdx.x >> 6] = acc;
        s_cnt[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int w = 0; w < LN_NLL_BLOCKS / 64; ++w) {
            a += s_sum[w];
            c += s_cnt[w];
        }
        loss_count[0] = -a / fmaxf(c, 1.f);
        loss_count[1] = fmaxf(c, 1.f);
    }
}

__global__ void __launch_bounds__(256)
    k_nll_backward(const long long* __restrict__ target, const float* __restrict__ grad_loss, const float* __restrict__ loss_count, long long n,
                   int C, long long ignore_index, float* __restrict__ grad_lp) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * C) return;
    const long long i = g / C;
    const int c = int(g - i * C);
    const long long t = target[i];
    const long long tc = t < 0 ? 0 : (t >= C ? C - 1 : t);
    grad_lp[g] = (t != ignore_index && c == tc) ? -grad_loss[0] / loss_count[1] : 0.f;
}

extern "C" size_t ln_nll_workspace_bytes(void) { return (size_t)LN_NLL_BLOCKS * 2 * sizeof(float); }

extern "C" int ln_nll_forward(const float* log_probs, const long long* target, long long n, int classes, long long ignore_index,
                              void* workspace, size_t workspace_bytes, float* loss_count, void* stream) {
    LN_REQUIRE(n >= 0 && classes >= 1, LN_ERR_ARG, "ln_nll_forward: bad sizes");
    LN_REQUIRE(loss_count && workspace && workspace_bytes >= ln_nll_workspace_bytes(), LN_ERR_ARG, "ln_nll_forward: null buffer / small workspace");
    LN_REQUIRE(n == 0 || (log_probs && target), LN_ERR_ARG, "ln_nll_forward: null buffer");
    hipStream_t st = (hipStream_t)stream;
    int blocks = ln_div_up(n, 256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > LN_NLL_BLOCKS ? LN_NLL_BLOCKS : blocks);
    float* partial = static_cast<float*>(workspace);
    LN_LAUNCH("k_nll_partials", k_nll_partials, dim3(blocks), dim3(256), 0, st, log_probs, target, n, classes, ignore_index, partial);
    LN_LAUNCH("k_nll_finish", k_nll_finish, dim3(1), dim3(LN_NLL_BLOCKS), 0, st, partial, blocks, loss_count);
    return ln_check_launch("ln_nll_forward");
}

extern "C" int ln_nll_backward(const long long* target, const float* grad_loss, const float* loss_count, long long n, int classes,
                               long long ignore_index, float* grad_log_probs, void* stream) {
    LN_REQUIRE(n >= 0 && classes >= 1, LN_ERR_ARG, "ln_nll_backward: bad sizes");
    if (n == 0) return LN_OK;
    LN_REQUIRE(target && grad_loss && loss_count && grad_log_probs, LN_ERR_ARG, "ln_nll_backward: null buffer");
    LN_LAUNCH("k_nll_backward", k_nll_backward, dim3(ln_div_up(n * classes, 256)), dim3(256), 0, (hipStream_t)stream, target, grad_loss, loss_count,
              n, classes, ignore_index, grad_log_probs);
    return ln_check_launch("ln_nll_backward");
}
