// Per-token linear layer + LeakyReLU for tall-and-skinny activations: y[T, Cout] = act(x[T, Cin] @ W^T + b) with
// T ~ 10^5..10^6 tokens and Cin, Cout <= 128 (PointNetModule's per-token MLP on the distributed rows,
// lattice_modules.py:636-676: [4N, 4] -> 16 -> 32).  A BLAS GEMM with K = 4..32 runs at a few percent of the memory
// roofline (0.4 ms per layer at C3); these kernels are plain streaming passes with the weight matrix in LDS.
//   forward       : thread = (token, 4 output channels); the token's input row is a broadcast read
//   backward / x  : thread = (token, 4 input channels)
//   backward / W,b: workgroups walk token tiles staged in LDS, (o, i) pairs in registers, one slab per workgroup,
//                   deterministic slab sum
#include "ln_common.h"

#define LN_MLP_MAX_C 128

__device__ __forceinline__ float ln_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// slope < 0: identity (no activation)
__global__ void __launch_bounds__(256)
    k_linear_act_forward(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, long long rows, int cin,
                         int cout, float slope, float* __restrict__ y) {
    extern __shared__ float s_w[];  // [cout, cin + 1]
    const int tid = threadIdx.x;
    for (int i = tid; i < cout * cin; i += 256) {
        const int o = i / cin;
        s_w[o * (cin + 1) + (i - o * cin)] = w[i];
    }
    __syncthreads();
    const int q = cout >> 2;  // output quads per token (cout % 4 == 0)
    const long long total = rows * q;
    const long long stride = (long long)gridDim.x * 256;
    for (long long g = (long long)blockIdx.x * 256 + tid; g < total; g += stride) {
        const long long t = g / q;
        const int o0 = int(g - t * q) * 4;
        const float* xr = x + t * cin;
        float a0 = b ? b[o0] : 0.f, a1 = b ? b[o0 + 1] : 0.f, a2 = b ? b[o0 + 2] : 0.f, a3 = b ? b[o0 + 3] : 0.f;
        const float* w0 = s_w + o0 * (cin + 1);
        for (int i = 0; i < cin; ++i) {
            const float xv = xr[i];
            a0 = fmaf(xv, w0[i], a0);
            a1 = fmaf(xv, w0[(cin + 1) + i], a1);
            a2 = fmaf(xv, w0[2 * (cin + 1) + i], a2);
            a3 = fmaf(xv, w0[3 * (cin + 1) + i], a3);
        }
        if (slope >= 0.f) {
            a0 = ln_lrelu(a0, slope);
            a1 = ln_lrelu(a1, slope);
            a2 = ln_lrelu(a2, slope);
            a3 = ln_lrelu(a3, slope);
        }
        *reinterpret_cast<float4*>(y + t * cout + o0) = make_float4(a0, a1, a2, a3);
    }
}

// gx[t, i] = sum_o g'[t, o] W[o, i],  g' = gy * act'(y)   (act' from the sign of the post-activation value)
__global__ void __launch_bounds__(256)
    k_linear_act_backward_x(const float* __restrict__ w, const float* __restrict__ y, const float* __restrict__ gy, long long rows, int cin,
                            int cout, float slope, float* __restrict__ gx) {
    extern __shared__ float s_w[];  // [cout, cin]  (lanes run over i: conflict-free)
    const int tid = threadIdx.x;
    for (int i = tid; i < cout * cin; i += 256) s_w[i] = w[i];
    __syncthreads();
    const int q = cin >> 2;  // cin % 4 == 0 on this path
    const long long total = rows * q;
    const long long stride = (long long)gridDim.x * 256;
    for (long long g = (long long)blockIdx.x * 256 + tid; g < total; g += stride) {
        const long long t = g / q;
        const int i0 = int(g - t * q) * 4;
        const float* gr = gy + t * cout;
        const float* yr = y + t * cout;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int o = 0; o < cout; ++o) {
            float gv = gr[o];
            if (slope >= 0.f && !(yr[o] > 0.f)) gv *= slope;
            const float* wr = s_w + o * cin + i0;
            a0 = fmaf(gv, wr[0], a0);
            a1 = fmaf(gv, wr[1], a1);
            a2 = fmaf(gv, wr[2], a2);
            a3 = fmaf(gv, wr[3], a3);
        }
        *reinterpret_cast<float4*>(gx + t * cin + i0) = make_float4(a0, a1, a2, a3);
    }
}

// slab[blockIdx.x][o*cin + i] = sum over the workgroup's tokens of g'[t, o] x[t, i];  slab[..][cout*cin + o] = sum g'[t, o]
#define LN_MLP_TILE 64
#define LN_MLP_MAXP 16  // (o, i) pairs per thread: cout * cin <= 4096
__global__ void __launch_bounds__(256)
    k_linear_act_backward_w(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy, long long rows, int cin,
                            int cout, float slope, float* __restrict__ slabs) {
    extern __shared__ float s_mem[];
    float* s_x = s_mem;                        // [TILE, cin]
    float* s_g = s_x + LN_MLP_TILE * cin;      // [TILE, cout + 1]
    const int tid = threadIdx.x;
    const int pairs = cout * cin;
    float acc[LN_MLP_MAXP];
#pragma unroll
    for (int k = 0; k < LN_MLP_MAXP; ++k) acc[k] = 0.f;
    float acc_b = 0.f;
    const long long tiles = (rows + LN_MLP_TILE - 1) / LN_MLP_TILE;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long t0 = tile * LN_MLP_TILE;
        __syncthreads();
        for (int i = tid; i < LN_MLP_TILE * cin; i += 256) {
            const long long t = t0 + i / cin;
            s_x[i] = t < rows ? x[t0 * cin + i] : 0.f;
        }
        for (int i = tid; i < LN_MLP_TILE * cout; i += 256) {
            const int lt = i / cout, o = i - lt * cout;
            const long long t = t0 + lt;
            float gv = 0.f;
            if (t < rows) {
                gv = gy[t0 * cout + i];
                if (slope >= 0.f && !(y[t0 * cout + i] > 0.f)) gv *= slope;
            }
            s_g[lt * (cout + 1) + o] = gv;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LN_MLP_MAXP; ++k) {
            const int j = tid + k * 256;
            if (j < pairs) {
                const int o = j / cin, i = j - o * cin;
                float a = acc[k];
#pragma unroll 8
                for (int lt = 0; lt < LN_MLP_TILE; ++lt) a = fmaf(s_g[lt * (cout + 1) + o], s_x[lt * cin + i], a);
                acc[k] = a;
            }
        }
        if (tid < cout)
            for (int lt = 0; lt < LN_MLP_TILE; ++lt) acc_b += s_g[lt * (cout + 1) + tid];
    }
    float* slab = slabs + (size_t)blockIdx.x * (pairs + cout);
#pragma unroll
    for (int k = 0; k < LN_MLP_MAXP; ++k) {
        const int j = tid + k * 256;
        if (j < pairs) slab[j] = acc[k];
    }
    if (tid < cout) slab[pairs + tid] = acc_b;
}

static int ln_mlp_check(const char* who, long long rows, int cin, int cout) {
    LN_REQUIRE(rows >= 0 && cin >= 1 && cout >= 4 && cout % 4 == 0 && cin <= LN_MLP_MAX_C && cout <= LN_MLP_MAX_C, LN_ERR_UNSUPPORTED,
               "%s: need cin <= %d, cout %% 4 == 0 and <= %d (got %d -> %d)", who, LN_MLP_MAX_C, LN_MLP_MAX_C, cin, cout);
    return LN_OK;
}

static int ln_mlp_grid(long long work_items) {
    long long g = (work_items + 255) / 256;
    if (g > 4096) g = 4096;
    return g < 1 ? 1 : int(g);
}

#define LN_MLP_W_GRID 512

extern "C" int ln_linear_act_forward(const float* x, const float* w, const float* b, long long rows, int cin, int cout, float slope, float* y,
                                     void* stream) {
    int rc = ln_mlp_check("ln_linear_act_forward", rows, cin, cout);
    if (rc) return rc;
    LN_REQUIRE(rows == 0 || (x && w && y), LN_ERR_ARG, "ln_linear_act_forward: null buffer");
    LN_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0, LN_ERR_ARG, "ln_linear_act_forward: y must be 16-byte aligned");
    if (rows == 0) return LN_OK;
    const size_t lds = sizeof(float) * (size_t)cout * (cin + 1);
    LN_LAUNCH("k_linear_act_forward", k_linear_act_forward, dim3(ln_mlp_grid(rows * (cout / 4))), dim3(256), lds, (hipStream_t)stream, x, w, b, rows,
              cin, cout, slope, y);
    return ln_check_launch("ln_linear_act_forward");
}

extern "C" size_t ln_linear_act_backward_workspace_bytes(int cin, int cout) {
    return (size_t)LN_MLP_W_GRID * ((size_t)cin * cout + cout) * sizeof(float) + 256;
}

extern "C" int ln_linear_act_backward(const float* x, const float* w, const float* y, const float* grad_y, long long rows, int cin, int cout,
                                      float slope, float* grad_x, float* grad_w, float* grad_b, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    int rc = ln_mlp_check("ln_linear_act_backward", rows, cin, cout);
    if (rc) return rc;
    LN_REQUIRE(grad_w && (rows == 0 || (x && w && y && grad_y)), LN_ERR_ARG, "ln_linear_act_backward: null buffer");
    LN_REQUIRE((long long)cin * cout <= 256 * LN_MLP_MAXP, LN_ERR_UNSUPPORTED, "ln_linear_act_backward: cin*cout = %d exceeds %d", cin * cout,
               256 * LN_MLP_MAXP);
    LN_REQUIRE(workspace && workspace_bytes >= ln_linear_act_backward_workspace_bytes(cin, cout), LN_ERR_WORKSPACE,
               "ln_linear_act_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int pairs = cin * cout;
    if (rows == 0) {
        (void)hipMemsetAsync(grad_w, 0, sizeof(float) * pairs, st);
        if (grad_b) (void)hipMemsetAsync(grad_b, 0, sizeof(float) * cout, st);
        return ln_check_launch("ln_linear_act_backward");
    }
    if (grad_x) {
        LN_REQUIRE(cin % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0, LN_ERR_UNSUPPORTED,
                   "ln_linear_act_backward: grad_x needs cin %% 4 == 0 and 16-byte alignment");
        LN_LAUNCH("k_linear_act_backward_x", k_linear_act_backward_x, dim3(ln_mlp_grid(rows * (cin / 4))), dim3(256),
                  sizeof(float) * (size_t)cout * cin, st, w, y, grad_y, rows, cin, cout, slope, grad_x);
    }
    float* slabs = static_cast<float*>(workspace);
    long long tiles = (rows + LN_MLP_TILE - 1) / LN_MLP_TILE;
    const int grid = int(tiles < LN_MLP_W_GRID ? tiles : LN_MLP_W_GRID);
    const size_t lds = sizeof(float) * LN_MLP_TILE * ((size_t)cin + cout + 1);
    LN_LAUNCH("k_linear_act_backward_w", k_linear_act_backward_w, dim3(grid), dim3(256), lds, st, x, y, grad_y, rows, cin, cout, slope, slabs);
    LN_LAUNCH("k_linear_reduce_slabs", ln_k_sum_slabs<false>, dim3(ln_div_up(pairs, 16)), dim3(256), 0, st, slabs, grid, (long long)(pairs + cout),
              pairs, grad_w);
    if (grad_b)
        LN_LAUNCH("k_linear_reduce_slabs", ln_k_sum_slabs<false>, dim3(ln_div_up(cout, 16)), dim3(256), 0, st, slabs + pairs, grid,
                  (long long)(pairs + cout), cout, grad_b);
    return ln_check_launch("ln_linear_act_backward");
}
