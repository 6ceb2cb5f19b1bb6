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

// any cout (e.g. the 9 -> 1 layer of the DeformSlice head): thread = (token, output channel)
__global__ void __launch_bounds__(256)
    k_linear_act_forward_scalar(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, long long rows, int cin,
                                int cout, float slope, float* __restrict__ y) {
    extern __shared__ float s_w[];  // [cout, cin + 1]
    const int tid = threadIdx.x;
    for (int i = tid; i < cout * cin; i += 256) {
        const int o = i / cin;
        s_w[o * (cin + 1) + (i - o * cin)] = w[i];
    }
    __syncthreads();
    const long long total = rows * cout;
    const long long stride = (long long)gridDim.x * 256;
    for (long long g = (long long)blockIdx.x * 256 + tid; g < total; g += stride) {
        const long long t = g / cout;
        const int o = int(g - t * cout);
        const float* xr = x + t * cin;
        const float* wr = s_w + o * (cin + 1);
        float a = b ? b[o] : 0.f;
        for (int i = 0; i < cin; ++i) a = fmaf(xr[i], wr[i], a);
        y[g] = slope >= 0.f ? ln_lrelu(a, slope) : a;
    }
}

// gx for any cin: thread = (token, input channel)
__global__ void __launch_bounds__(256)
    k_linear_act_backward_x_scalar(const float* __restrict__ w, const float* __restrict__ y, const float* __restrict__ gy, long long rows, int cin,
                                   int cout, float slope, float* __restrict__ gx) {
    extern __shared__ float s_w[];  // [cout, cin]
    const int tid = threadIdx.x;
    for (int i = tid; i < cout * cin; i += 256) s_w[i] = w[i];
    __syncthreads();
    const long long total = rows * cin;
    const long long stride = (long long)gridDim.x * 256;
    for (long long g = (long long)blockIdx.x * 256 + tid; g < total; g += stride) {
        const long long t = g / cin;
        const int i = int(g - t * cin);
        float a = 0.f;
        for (int o = 0; o < cout; ++o) {
            float gv = gy[t * cout + o];
            if (slope >= 0.f && !(y[t * cout + o] > 0.f)) gv *= slope;
            a = fmaf(gv, s_w[o * cin + i], a);
        }
        gx[g] = a;
    }
}

// gx[t, i] = sum_o g'[t, o] W[o, i],  g' = gy * act'(y)   (act' from the sign of the post-activation value)
__global__ void __launch_bounds__(256)
    k_linear_act_backward_x(const float* __restrict__ w, const float* __restrict__ y, const float* __restrict__ gy, long long rows, int cin,
                            int cout, float slope, float* __restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) float s_w4[];  // [cout, cin]  (lanes run over i: conflict-free)
    const int tid = threadIdx.x;
    for (int i = tid; i < cout * cin; i += 256) s_w4[i] = w[i];
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
        auto step = [&](float gv, float yv, int o) {
            if (slope >= 0.f && !(yv > 0.f)) gv *= slope;
            const float4 wv = *reinterpret_cast<const float4*>(s_w4 + o * cin + i0);
            a0 = fmaf(gv, wv.x, a0);
            a1 = fmaf(gv, wv.y, a1);
            a2 = fmaf(gv, wv.z, a2);
            a3 = fmaf(gv, wv.w, a3);
        };
        if ((cout & 3) == 0) {  // rows of gy / y are 16-byte aligned: four output channels per load
#pragma unroll 2
            for (int o = 0; o < cout; o += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gr + o);
                const float4 y4 = slope >= 0.f ? *reinterpret_cast<const float4*>(yr + o) : make_float4(1.f, 1.f, 1.f, 1.f);
                step(g4.x, y4.x, o);
                step(g4.y, y4.y, o + 1);
                step(g4.z, y4.z, o + 2);
                step(g4.w, y4.w, o + 3);
            }
        } else {
            for (int o = 0; o < cout; ++o) step(gr[o], slope >= 0.f ? yr[o] : 1.f, o);
        }
        *reinterpret_cast<float4*>(gx + t * cin + i0) = make_float4(a0, a1, a2, a3);
    }
}

// slab[blockIdx.x][o*cin + i] = sum over the workgroup's tokens of g'[t, o] x[t, i];  slab[..][cout*cin + o] = sum g'[t, o]
#define LN_MLP_TILE 64
#define LN_MLP_MAXP 40  // (o, i) pairs per thread: cout * cin <= 10240
__global__ void __launch_bounds__(256)
    k_linear_act_backward_w(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy, long long rows, int cin,
                            int cout, float slope, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_x = s_mem;                        // [TILE, cin]
    float* s_g = s_x + LN_MLP_TILE * cin;      // [TILE, cout]  (lanes that share an output channel read one address: broadcast)
    __shared__ float s_red[256];
    const int tid = threadIdx.x;
    const int pairs = cout * cin;
    // Layers with fewer than 256 (o, i) pairs: the tokens of a tile are split over 256 / pairs thread groups, folded at the end.
    const int P = pairs < 256 ? pairs : 256;
    const int G = 256 / P;
    const int grp = tid / P;
    const int j0 = tid - grp * P;
    const bool active = grp < G;
    float acc[LN_MLP_MAXP];
#pragma unroll
    for (int k = 0; k < LN_MLP_MAXP; ++k) acc[k] = 0.f;
    float acc_b = 0.f;
    const long long tiles = (rows + LN_MLP_TILE - 1) / LN_MLP_TILE;
    const long long x_elems = rows * cin, g_elems = rows * cout;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long t0 = tile * LN_MLP_TILE;
        __syncthreads();
        // contiguous copies of the tile's rows (16-byte aligned: a tile starts at a multiple of 64 rows), zero beyond the last row
        {
            const long long base = t0 * cin;
            for (int v = tid; v < LN_MLP_TILE * cin / 4; v += 256) {
                const long long e = base + (long long)v * 4;
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e + 3 < x_elems) {
                    xv = *reinterpret_cast<const float4*>(x + e);
                } else {
                    if (e < x_elems) xv.x = x[e];
                    if (e + 1 < x_elems) xv.y = x[e + 1];
                    if (e + 2 < x_elems) xv.z = x[e + 2];
                }
                *reinterpret_cast<float4*>(s_x + v * 4) = xv;
            }
        }
        {
            const long long base = t0 * cout;
            for (int v = tid; v < LN_MLP_TILE * cout / 4; v += 256) {
                const long long e = base + (long long)v * 4;
                float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), yv = make_float4(1.f, 1.f, 1.f, 1.f);
                if (e + 3 < g_elems) {
                    gv = *reinterpret_cast<const float4*>(gy + e);
                    if (slope >= 0.f) yv = *reinterpret_cast<const float4*>(y + e);
                } else {
                    if (e < g_elems) { gv.x = gy[e]; if (slope >= 0.f) yv.x = y[e]; }
                    if (e + 1 < g_elems) { gv.y = gy[e + 1]; if (slope >= 0.f) yv.y = y[e + 1]; }
                    if (e + 2 < g_elems) { gv.z = gy[e + 2]; if (slope >= 0.f) yv.z = y[e + 2]; }
                }
                if (slope >= 0.f) {
                    if (!(yv.x > 0.f)) gv.x *= slope;
                    if (!(yv.y > 0.f)) gv.y *= slope;
                    if (!(yv.z > 0.f)) gv.z *= slope;
                    if (!(yv.w > 0.f)) gv.w *= slope;
                }
                *reinterpret_cast<float4*>(s_g + v * 4) = gv;
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < LN_MLP_MAXP; ++k) {
                const int j = j0 + k * 256;
                if (j < pairs && (k == 0 || G == 1)) {
                    const int o = j / cin, i = j - o * cin;
                    float a = acc[k];
#pragma unroll 8
                    for (int lt = grp; lt < LN_MLP_TILE; lt += G) a = fmaf(s_g[lt * cout + o], s_x[lt * cin + i], a);
                    acc[k] = a;
                }
            }
        }
        if (tid < cout)
            for (int lt = 0; lt < LN_MLP_TILE; ++lt) acc_b += s_g[lt * cout + tid];
    }
    float* slab = slabs + (size_t)blockIdx.x * (pairs + cout);
    if (G == 1) {
#pragma unroll
        for (int k = 0; k < LN_MLP_MAXP; ++k) {
            const int j = tid + k * 256;
            if (j < pairs) slab[j] = acc[k];
        }
    } else {  // fold the token groups of each pair in a fixed order
        s_red[tid] = active ? acc[0] : 0.f;
        __syncthreads();
        if (tid < P) {
            float a = 0.f;
            for (int g = 0; g < G; ++g) a += s_red[g * P + tid];
            slab[tid] = a;
        }
    }
    if (tid < cout) slab[pairs + tid] = acc_b;
}

// Same slabs for wide layers (cin, cout multiples of 4, more than 1024 pairs): a thread owns up to LN_MLP_TILED_BLOCKS
// 4x4 blocks of (o, i) pairs, so one row of a token tile costs two float4 LDS reads per 16 FMAs instead of two reads per FMA.
#define LN_MLP_TILED_BLOCKS 3  // 256 threads x 3 blocks x 16 pairs = 12288 pairs
__global__ void __launch_bounds__(256)
    k_linear_act_backward_w_tiled(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy, long long rows, int cin,
                                  int cout, float slope, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_x = s_mem;                        // [TILE, cin]
    float* s_g = s_x + LN_MLP_TILE * cin;      // [TILE, cout]
    const int tid = threadIdx.x;
    const int pairs = cout * cin;
    const int ib = cin >> 2, ob = cout >> 2;   // 4-wide blocks per dimension
    const int nblocks = ib * ob;
    float acc[LN_MLP_TILED_BLOCKS][16];
#pragma unroll
    for (int k = 0; k < LN_MLP_TILED_BLOCKS; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
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
            const long long t = t0 + i / cout;
            float gv = 0.f;
            if (t < rows) {
                gv = gy[t0 * cout + i];
                if (slope >= 0.f && !(y[t0 * cout + i] > 0.f)) gv *= slope;
            }
            s_g[i] = gv;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LN_MLP_TILED_BLOCKS; ++k) {
            const int blk = tid + k * 256;
            if (blk < nblocks) {
                const int o4 = blk / ib, i4 = blk - o4 * ib;
                const float* pg = s_g + o4 * 4;
                const float* px = s_x + i4 * 4;
#pragma unroll 4
                for (int lt = 0; lt < LN_MLP_TILE; ++lt) {
                    const float4 g4 = *reinterpret_cast<const float4*>(pg + lt * cout);
                    const float4 x4 = *reinterpret_cast<const float4*>(px + lt * cin);
                    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
                    const float xx[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[k][a * 4 + b] = fmaf(gg[a], xx[b], acc[k][a * 4 + b]);
                }
            }
        }
        if (tid < cout)
            for (int lt = 0; lt < LN_MLP_TILE; ++lt) acc_b += s_g[lt * cout + tid];
    }
    float* slab = slabs + (size_t)blockIdx.x * (pairs + cout);
#pragma unroll
    for (int k = 0; k < LN_MLP_TILED_BLOCKS; ++k) {
        const int blk = tid + k * 256;
        if (blk < nblocks) {
            const int o4 = blk / ib, i4 = blk - o4 * ib;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) slab[(o4 * 4 + a) * cin + i4 * 4 + b] = acc[k][a * 4 + b];
        }
    }
    if (tid < cout) slab[pairs + tid] = acc_b;
}

static int ln_mlp_check(const char* who, long long rows, int cin, int cout) {
    LN_REQUIRE(rows >= 0 && cin >= 1 && cout >= 1 && cin <= LN_MLP_MAX_C && cout <= LN_MLP_MAX_C, LN_ERR_UNSUPPORTED,
               "%s: need 1 <= cin, cout <= %d (got %d -> %d)", who, LN_MLP_MAX_C, cin, cout);
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
    if (rows == 0) return LN_OK;
    const size_t lds = sizeof(float) * (size_t)cout * (cin + 1);
    if (cout % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
        LN_LAUNCH("k_linear_act_forward", k_linear_act_forward, dim3(ln_mlp_grid(rows * (cout / 4))), dim3(256), lds, (hipStream_t)stream, x, w, b,
                  rows, cin, cout, slope, y);
    else
        LN_LAUNCH("k_linear_act_forward", k_linear_act_forward_scalar, dim3(ln_mlp_grid(rows * cout)), dim3(256), lds, (hipStream_t)stream, x, w, b,
                  rows, cin, cout, slope, y);
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
        (void)ln_zero_async(grad_w, sizeof(float) * pairs, st);
        if (grad_b) (void)ln_zero_async(grad_b, sizeof(float) * cout, st);
        return ln_check_launch("ln_linear_act_backward");
    }
    if (grad_x) {
        if (cin % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0)
            LN_LAUNCH("k_linear_act_backward_x", k_linear_act_backward_x, dim3(ln_mlp_grid(rows * (cin / 4))), dim3(256),
                      sizeof(float) * (size_t)cout * cin, st, w, y, grad_y, rows, cin, cout, slope, grad_x);
        else
            LN_LAUNCH("k_linear_act_backward_x", k_linear_act_backward_x_scalar, dim3(ln_mlp_grid(rows * cin)), dim3(256),
                      sizeof(float) * (size_t)cout * cin, st, w, y, grad_y, rows, cin, cout, slope, grad_x);
    }
    float* slabs = static_cast<float*>(workspace);
    long long tiles = (rows + LN_MLP_TILE - 1) / LN_MLP_TILE;
    const int grid = int(tiles < LN_MLP_W_GRID ? tiles : LN_MLP_W_GRID);  // measured: 256 and 2048 workgroups are both slower for the narrow layers
    if (cin % 4 == 0 && cout % 4 == 0 && pairs > 1024 && pairs <= 256 * LN_MLP_TILED_BLOCKS * 16) {
        const size_t lds = sizeof(float) * LN_MLP_TILE * ((size_t)cin + cout);
        LN_LAUNCH("k_linear_act_backward_w", k_linear_act_backward_w_tiled, dim3(grid), dim3(256), lds, st, x, y, grad_y, rows, cin, cout, slope,
                  slabs);
    } else {
        const size_t lds = sizeof(float) * LN_MLP_TILE * ((size_t)cin + cout + 1);
        LN_LAUNCH("k_linear_act_backward_w", k_linear_act_backward_w, dim3(grid), dim3(256), lds, st, x, y, grad_y, rows, cin, cout, slope, slabs);
    }
    LN_LAUNCH("k_linear_reduce_slabs", ln_k_sum_slabs<false>, dim3(ln_div_up(pairs, 16)), dim3(256), 0, st, slabs, grid, (long long)(pairs + cout),
              pairs, grad_w);
    if (grad_b)
        LN_LAUNCH("k_linear_reduce_slabs", ln_k_sum_slabs<false>, dim3(ln_div_up(cout, 16)), dim3(256), 0, st, slabs + pairs, grid,
                  (long long)(pairs + cout), cout, grad_b);
    return ln_check_launch("ln_linear_act_backward");
}
