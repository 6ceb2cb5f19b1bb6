// Max-centring of the gathered simplex rows in the DeformSlice head (lattice_modules.py:525-529):
//   x [N, K, C]  (K = d+1 vertex rows of C values per point)
//   out[n,k,c] = x[n,k,c] - (gamma[c] * max_k x[n,k,c] + beta[c])
// The reference writes it with torch broadcasting; its backward then reduces two [N, 1, C] tensors over the N points
// with torch's generic column reduction (two passes of ~150 us each at N = 120 k, C = 9).  Here: one pass forward
// (max, arg-max and the centred rows), one pass backward
//   s[n,c]      = sum_k g[n,k,c]
//   gx[n,k,c]   = g[n,k,c] - [k == argmax[n,c]] * gamma[c] * s[n,c]
//   ggamma[c]   = -sum_n s[n,c] * max[n,c]          gbeta[c] = -sum_n s[n,c]
// with the two parameter gradients summed per workgroup in a fixed order, written as slabs and folded by
// ln_k_sum_slabs (deterministic).
#include "ln_common.h"

#define LN_MC_MAX_K 8
#define LN_MC_MAX_C 64
#ifndef LN_MC_ITERS
#define LN_MC_ITERS 4  // points per thread.  (16 until round 5: 268 workgroups for 120 k points, one wave per SIMD walking 16 dependent
                       // iterations: forward 23.2 / backward 37.4 us; 4: 12.7 / 15.5 us + 1.6 us more in the slab sum; 2: 13.5 / 18.2 + 6.5)
#endif

// thread t of a workgroup: channel c = t % C, point lane pl = t / C; lanes = 256 / C point lanes are live
__global__ void __launch_bounds__(256)
    k_max_centre_forward(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, long long n, int K, int C,
                         float* __restrict__ out, float* __restrict__ max_vals, unsigned char* __restrict__ arg_max) {
    const int lanes = 256 / C;
    const int pl = threadIdx.x / C;
    const int c = threadIdx.x - pl * C;
    if (pl >= lanes) return;
    const float gm = gamma[c], bt = beta[c];
    const long long p0 = (long long)blockIdx.x * lanes * LN_MC_ITERS;
    for (int it = 0; it < LN_MC_ITERS; ++it) {
        const long long p = p0 + (long long)it * lanes + pl;
        if (p >= n) break;
        const float* xr = x + p * K * C + c;
        float v[LN_MC_MAX_K];
        float mx = -INFINITY;
        int am = 0;
#pragma unroll
        for (int k = 0; k < LN_MC_MAX_K; ++k)
            if (k < K) {
                v[k] = xr[(size_t)k * C];
                if (v[k] > mx) {  // first maximum wins
                    mx = v[k];
                    am = k;
                }
            }
        const float shift = gm * mx + bt;
        float* o = out + p * K * C + c;
#pragma unroll
        for (int k = 0; k < LN_MC_MAX_K; ++k)
            if (k < K) o[(size_t)k * C] = v[k] - shift;
        max_vals[p * C + c] = mx;
        arg_max[p * C + c] = (unsigned char)am;
    }
}

__global__ void __launch_bounds__(256)
    k_max_centre_backward(const float* __restrict__ g, const float* __restrict__ max_vals, const unsigned char* __restrict__ arg_max,
                          const float* __restrict__ gamma, long long n, int K, int C, float* __restrict__ gx, float* __restrict__ slabs) {
    __shared__ float s_gg[256], s_gb[256];
    const int lanes = 256 / C;
    const int pl = threadIdx.x / C;
    const int c = threadIdx.x - pl * C;
    float acc_gg = 0.f, acc_gb = 0.f;
    if (pl < lanes) {
        const float gm = gamma[c];
        const long long p0 = (long long)blockIdx.x * lanes * LN_MC_ITERS;
        for (int it = 0; it < LN_MC_ITERS; ++it) {
            const long long p = p0 + (long long)it * lanes + pl;
            if (p >= n) break;
            const float* gr = g + p * K * C + c;
            float v[LN_MC_MAX_K];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LN_MC_MAX_K; ++k)
                if (k < K) {
                    v[k] = gr[(size_t)k * C];
                    s += v[k];
                }
            const int am = arg_max[p * C + c];
            float* o = gx + p * K * C + c;
#pragma unroll
            for (int k = 0; k < LN_MC_MAX_K; ++k)
                if (k < K) o[(size_t)k * C] = (k == am) ? v[k] - gm * s : v[k];
            acc_gg -= s * max_vals[p * C + c];
            acc_gb -= s;
        }
    }
    s_gg[threadIdx.x] = acc_gg;
    s_gb[threadIdx.x] = acc_gb;
    __syncthreads();
    if (threadIdx.x < C) {  // fold the point lanes of channel c in a fixed order
        float a = 0.f, b = 0.f;
        for (int l = 0; l < lanes; ++l) {
            a += s_gg[l * C + threadIdx.x];
            b += s_gb[l * C + threadIdx.x];
        }
        slabs[(size_t)blockIdx.x * 2 * C + threadIdx.x] = a;
        slabs[(size_t)blockIdx.x * 2 * C + C + threadIdx.x] = b;
    }
}

static int ln_mc_check(const char* who, long long n, int k, int c) {
    LN_REQUIRE(n >= 0 && k >= 1 && k <= LN_MC_MAX_K && c >= 1 && c <= LN_MC_MAX_C, LN_ERR_UNSUPPORTED,
               "%s: need 1 <= vertices per simplex <= %d and 1 <= channels <= %d (got %d, %d)", who, LN_MC_MAX_K, LN_MC_MAX_C, k, c);
    return LN_OK;
}

static int ln_mc_blocks(long long n, int c) { return ln_div_up(n, (long long)(256 / c) * LN_MC_ITERS); }

extern "C" int ln_max_centre_forward(const float* x, const float* gamma, const float* beta, long long n, int k, int c, float* out,
                                     float* max_vals, unsigned char* arg_max, void* stream) {
    int rc = ln_mc_check("ln_max_centre_forward", n, k, c);
    if (rc) return rc;
    if (n == 0) return LN_OK;
    LN_REQUIRE(x && gamma && beta && out && max_vals && arg_max, LN_ERR_ARG, "ln_max_centre_forward: null buffer");
    hipStream_t st = (hipStream_t)stream;
    LN_LAUNCH("k_max_centre_forward", k_max_centre_forward, dim3(ln_mc_blocks(n, c)), dim3(256), 0, st, x, gamma, beta, n, k, c, out, max_vals, arg_max);
    return ln_check_launch("ln_max_centre_forward");
}

extern "C" size_t ln_max_centre_backward_workspace_bytes(long long n, int k, int c) {
    (void)k;
    if (n < 1 || c < 1 || c > LN_MC_MAX_C) return 256;
    return (size_t)ln_mc_blocks(n, c) * 2 * c * sizeof(float) + 256;
}

extern "C" int ln_max_centre_backward(const float* grad_out, const float* max_vals, const unsigned char* arg_max, const float* gamma,
                                      long long n, int k, int c, float* grad_x, float* grad_gamma_beta, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    int rc = ln_mc_check("ln_max_centre_backward", n, k, c);
    if (rc) return rc;
    LN_REQUIRE(grad_gamma_beta, LN_ERR_ARG, "ln_max_centre_backward: null parameter-gradient buffer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (ln_zero_async(grad_gamma_beta, (size_t)2 * c * sizeof(float), st) != LN_OK)
            return ln_check_launch("ln_max_centre_backward(memset)");
        return LN_OK;
    }
    LN_REQUIRE(grad_out && max_vals && arg_max && gamma && grad_x, LN_ERR_ARG, "ln_max_centre_backward: null buffer");
    LN_REQUIRE(workspace && workspace_bytes >= ln_max_centre_backward_workspace_bytes(n, k, c), LN_ERR_WORKSPACE,
               "ln_max_centre_backward: workspace too small");
    const int blocks = ln_mc_blocks(n, c);
    float* slabs = static_cast<float*>(workspace);
    LN_LAUNCH("k_max_centre_backward", k_max_centre_backward, dim3(blocks), dim3(256), 0, st, grad_out, max_vals, arg_max, gamma, n, k, c, grad_x, slabs);
    // [blocks][2C] -> [2, C]: row 0 = d gamma, row 1 = d beta
    LN_LAUNCH("k_max_centre_sum", ln_k_sum_slabs<false>, dim3(ln_div_up(2 * c, 16)), dim3(256), 0, st, slabs, blocks, (long long)2 * c, 2 * c,
              grad_gamma_beta);
    return ln_check_launch("ln_max_centre_backward");
}
