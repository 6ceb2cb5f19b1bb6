// Fused slice + linear classifier (LatticeGPU.cuh:3387-3464, 3628-3756), wave-tiled kernels for the widths the reference's
// networks use (V a multiple of 32, up to 32 classes, d in {2, 3}); ln_rows.hip keeps the general kernels and calls these first.
//
// Both kernels give every WAVE its own tile of points and its own slice of LDS: no workgroup barrier anywhere, a wave that
// waits for its row gathers does not hold up the three others of its workgroup.
//
// Forward (k_slice_classify_forward_wave): tile = 64 points, channels in chunks of 32 (= one 128-byte line per vertex row).
//   gather : 8 lanes per point read the chunk of each of the d+1 vertex rows as one line (float4 per lane), the sliced
//            features h = sum_r row_r * (w_r + delta_w_r) go to LDS [64][33];
//   linear : lane p owns point p and ALL classes in registers; the classifier weights are wave-uniform and are broadcast from
//            LDS (staged once per workgroup; an s_load variant through the scalar cache measured slower and was dropped):
//            per (class, channel) one v_mul + one v_add.
//   Every logit is still the serial sum over v ascending of W[c,v] * h[p,v] with separate multiply and add, then + b[c]
//   (compiled with -ffp-contract=off): bit-identical to the golden vectors.
//
// Backward (k_slice_classify_backward_wave): tile = 16 points (one MFMA row block), fp32 matrix instructions for both dense
// products (v_mfma_f32_16x16x4_f32; exact fp32 products, fp32 accumulate):
//   gh = g @ W            [16, C] x [C, V]   -> LDS, grad_sliced (HBM)    KS * NT instructions
//   gather                8 lanes per point: h (-> LDS), d(delta_w)[p, r] = row_r . gh[p]  (in-lane partial dots, 3 xor steps)
//   gW += g^T @ [h | 1]   [C, 16] x [16, V+1] accumulated in registers over all tiles of the wave; the extra column of ones
//                         yields the bias gradient.  One slab per workgroup, summed by ln_k_sum_slabs (ln_rows.hip).
#include "ln_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ln_wave_sync() {
    // LDS operations of one wave execute in program order; this only stops the compiler from moving them across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef LN_SCW_PROBE
#define LN_SCW_PROBE 0
#endif
#ifndef LN_SCW_FWD_WAVES
#define LN_SCW_FWD_WAVES 3  // waves per SIMD the register budget is held to (LDS admits three workgroups per CU)
#endif
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {  // the value of another lane of the row (DPP control CTRL), no LDS round trip
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

#ifndef LN_SCB_PROBE
#define LN_SCB_PROBE 0
#endif
#ifndef LN_SCW_BWD_WAVES
#define LN_SCW_BWD_WAVES 2
#endif
#define LN_SCW_SH 33  // row stride of the forward's h tile (floats): lanes = points read conflict-free

template <int DP1, int CT>
__global__ void __launch_bounds__(256, LN_SCW_FWD_WAVES)
    k_slice_classify_forward_wave(const float* __restrict__ values, const float* __restrict__ delta_w, const float* __restrict__ lin_w,
                                  const float* __restrict__ lin_b, const int* __restrict__ idx, const float* __restrict__ w, int n,
                                  int V, int C, float* __restrict__ logits) {
    __shared__ float s_h_all[4][64 * LN_SCW_SH];
    __shared__ float s_we_all[4][64 * DP1];
    __shared__ int s_idx_all[4][64 * DP1];
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // [C][V] classifier weights
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* s_h = s_h_all[wave];
    float* s_we = s_we_all[wave];
    int* s_idx = s_idx_all[wave];
    const int tiles = (n + 63) >> 6;
    const int chunks = V >> 5;
    const int s = lane & 7;
    for (int e = threadIdx.x; e < C * V; e += 256) s_w[e] = lin_w[e];
    __syncthreads();
    // one batch = 2 points per 8-lane group x (d+1) vertex rows: 2 (d+1) line reads per lane; two batches in flight
    auto issue = [&](int ch, int bt, float4 (&x)[2][DP1], int (&rows)[2][DP1]) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int lp = (bt * 2 + ps) * 8 + (lane >> 3);
#pragma unroll
            for (int r = 0; r < DP1; ++r) {
                rows[ps][r] = s_idx[lp * DP1 + r];
#if LN_SCW_PROBE == 2
                x[ps][r] = make_float4(1.f, 2.f, 3.f, float(rows[ps][r]));
#else
                x[ps][r] = reinterpret_cast<const float4*>(values + (size_t)(rows[ps][r] >= 0 ? rows[ps][r] : 0) * V + ch * 32)[s];
#endif
            }
        }
    };
    auto finish = [&](int bt, const float4 (&x)[2][DP1], const int (&rows)[2][DP1]) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int lp = (bt * 2 + ps) * 8 + (lane >> 3);
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < DP1; ++r) {
                // LatticeGPU.cuh:3418-3428: rows r ascending, absent vertices skipped — here a zero row (h + 0 * w == h bit for bit;
                // a select instead of a branch per row)
                const bool ok = rows[ps][r] >= 0;
                const float wt = s_we[lp * DP1 + r];
                const float4 xv = x[ps][r];
                h.x = h.x + (ok ? xv.x : 0.0f) * wt; h.y = h.y + (ok ? xv.y : 0.0f) * wt;
                h.z = h.z + (ok ? xv.z : 0.0f) * wt; h.w = h.w + (ok ? xv.w : 0.0f) * wt;
            }
            float* d = s_h + lp * LN_SCW_SH + s * 4;
            d[0] = h.x; d[1] = h.y; d[2] = h.z; d[3] = h.w;
        }
    };
    for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
        const long long p0 = (long long)tile * 64;
        const long long tok_end = (long long)n * DP1;
#pragma unroll
        for (int k = 0; k < DP1; ++k) {
            const int i = lane + 64 * k;
            const long long t = p0 * DP1 + i;
            const bool ok = t < tok_end;
            s_idx[i] = ok ? idx[t] : -1;
            s_we[i] = ok ? w[t] + delta_w[t] : 0.0f;
        }
        ln_wave_sync();
        float acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.0f;
        for (int ch = 0; ch < chunks; ++ch) {
            {
                float4 xa[2][DP1], xb[2][DP1];
                int ra[2][DP1], rb[2][DP1];
                issue(ch, 0, xa, ra);
                issue(ch, 1, xb, rb);
                finish(0, xa, ra);
                issue(ch, 2, xa, ra);
                finish(1, xb, rb);
                issue(ch, 3, xb, rb);
                finish(2, xa, ra);
                finish(3, xb, rb);
            }
            ln_wave_sync();
            float hv[32];
#pragma unroll
            for (int v = 0; v < 32; ++v) hv[v] = s_h[lane * LN_SCW_SH + v];
            // Classifier weights are wave-uniform: staged once per workgroup in LDS and read with 16-byte loads at a uniform
            // address (an LDS broadcast: four weights per instruction), four classes' chains interleaved.  Measured alternatives:
            // s_load through the scalar cache (~0.35 us per 16 weights with the one or two loads in flight the SGPR file has room
            // for: the kernel ran at the scalar cache's latency), lane-distributed weights handed over by v_readlane (~22 cycles per
            // multiply-add instead of 8).  No branch on c < C: padding classes recompute class C - 1 and are dropped at the store.
            const float* wch = s_w + ch * 32;
#pragma unroll
            for (int c0 = 0; c0 < CT; c0 += 4) {
                const float* wc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) wc[k] = wch + (c0 + k < C ? c0 + k : C - 1) * V;
#if LN_SCW_PROBE == 1
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c0 + k] = acc[c0 + k] + wc[k][0] * hv[k];
#else
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 w4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) w4[k] = *reinterpret_cast<const float4*>(wc[k] + 4 * j);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[c0 + k] = acc[c0 + k] + w4[k].x * hv[4 * j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[c0 + k] = acc[c0 + k] + w4[k].y * hv[4 * j + 1];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[c0 + k] = acc[c0 + k] + w4[k].z * hv[4 * j + 2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[c0 + k] = acc[c0 + k] + w4[k].w * hv[4 * j + 3];
                }
#endif
            }
            ln_wave_sync();  // the next chunk overwrites s_h
        }
        // logits of the tile are contiguous in HBM: stage [64][C] in LDS, store lane-linear
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c < C) s_h[lane * C + c] = acc[c] + lin_b[c];
        ln_wave_sync();
        const long long left = (long long)n - p0;
        const int cnt = int(left < 64 ? left : 64) * C;
        float* out = logits + p0 * C;
        for (int i = lane; i < cnt; i += 64) out[i] = s_h[i];
        ln_wave_sync();
    }
}

int ln_sc_forward_wave(const float* values, const float* delta_w, const float* lin_w, const float* lin_b, const int* idx, const float* w,
                       int n, int pos_dim, int val_dim, int nr_classes, float* logits, hipStream_t st) {
    // returns 1 when launched, 0 when the shape is not covered (the caller falls back to the general kernels)
    if (val_dim % 32 != 0 || nr_classes > 32 || (pos_dim != 2 && pos_dim != 3) || (reinterpret_cast<uintptr_t>(values) & 15) != 0) return 0;
    const size_t lds = sizeof(float) * (size_t)nr_classes * val_dim;
    if (lds > 16 * 1024) return 0;
    const int tiles = ln_div_up(n, 64);
    const int grid = ln_div_up(tiles, 4);
#define LN_SCW_FWD(DP1, CT)                                                                                                        \
    LN_LAUNCH("k_slice_classify_forward", (k_slice_classify_forward_wave<DP1, CT>), dim3(grid), dim3(256), lds, st, values, delta_w, lin_w, \
              lin_b, idx, w, n, val_dim, nr_classes, logits)
#define LN_SCW_FWD_C(DP1)                            \
    switch ((nr_classes + 3) / 4) {                  \
        case 1: LN_SCW_FWD(DP1, 4); break;           \
        case 2: LN_SCW_FWD(DP1, 8); break;           \
        case 3: LN_SCW_FWD(DP1, 12); break;          \
        case 4: LN_SCW_FWD(DP1, 16); break;          \
        case 5: LN_SCW_FWD(DP1, 20); break;          \
        case 6: LN_SCW_FWD(DP1, 24); break;          \
        case 7: LN_SCW_FWD(DP1, 28); break;          \
        default: LN_SCW_FWD(DP1, 32); break;         \
    }
    if (pos_dim == 3) {
        LN_SCW_FWD_C(4);
    } else {
        LN_SCW_FWD_C(3);
    }
#undef LN_SCW_FWD_C
#undef LN_SCW_FWD
    return 1;
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// U = V / 32 (float4 per lane and vertex row in the gather), NT = V / 16 = 2U channel tiles, CTL = class tiles of 16.
// A wave walks its tiles with everything that only depends on the tile index fetched ONE TILE AHEAD (tokens, the gradient rows
// in both fragment layouts), so that a tile's own chain is LDS -> row gathers -> LDS -> matrix instructions.
template <int DP1, int U, int CTL>
__global__ void __launch_bounds__(256, LN_SCW_BWD_WAVES)
    k_slice_classify_backward_wave(const float* __restrict__ grad_logits, const float* __restrict__ values,
                                   const float* __restrict__ delta_w, const float* __restrict__ lin_w, const int* __restrict__ idx,
                                   const float* __restrict__ w, int n, int C, float* __restrict__ g_delta_w,
                                   float* __restrict__ grad_sliced, float* __restrict__ w_eff, float* __restrict__ slabs) {
    constexpr int V = 32 * U;
    constexpr int NT = 2 * U;
    constexpr int KSMAX = 8;     // C <= 32
    // ONE tile buffer per wave: gh = g @ W is written in D-fragment order, read back per point by the gather, which overwrites it IN
    // PLACE with the sliced features h (same lane, same 16 bytes), read in B-fragment order by the classifier gradient.  Stride
    // V + 4: the four row groups of a D fragment land on banks 16 apart, rows stay 16-byte aligned (the B-fragment reads of the
    // last phase then overlap on 12 of 32 banks: two cycles instead of one for NT reads per K step)
    constexpr int SG = V + 4;
    constexpr int WAVE_LDS = 16 * SG;  // floats
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KS = (C + 3) >> 2;  // K steps of gh = g @ W
    float* s_wb = smem;                                   // [KS][NT][64]  classifier in B-fragment order
    float* s_wave = s_wb + KS * NT * 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* s_gh = s_wave + wave * WAVE_LDS;               // [16][SG]
    __shared__ float s_we_all[4][16 * DP1];
    __shared__ int s_idx_all[4][16 * DP1];
    __shared__ float s_dw_all[4][16 * DP1];
    float* s_we = s_we_all[wave];
    int* s_idx = s_idx_all[wave];
    float* s_dw = s_dw_all[wave];
    const int i = lane & 15, q = lane >> 4;
    for (int e = threadIdx.x; e < KS * NT * 64; e += 256) {
        const int l = e & 63, f = e >> 6;
        const int ks = f / NT, nt = f - ks * NT;
        const int c = ks * 4 + (l >> 4), v = nt * 16 + (l & 15);
        s_wb[e] = c < C ? lin_w[(size_t)c * V + v] : 0.0f;
    }
    __syncthreads();
    floatx4 acc_w[CTL][NT];
    floatx4 acc_b[CTL];
#pragma unroll
    for (int ct = 0; ct < CTL; ++ct) {
        acc_b[ct] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc_w[ct][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const int tiles = (n + 15) >> 4;
    const long long tok_end = (long long)n * DP1;
    static_assert(16 * DP1 <= 64, "one token per lane");
    // what a tile needs from HBM before its first LDS access
    int t_idx = -1;
    float t_we = 0.0f, t_gdw = 0.0f;
    float a1[KSMAX];        // A fragments of gh = g @ W:    A[i = point][k = class]
    float a3[4][CTL];       // A fragments of gW += g^T h:   A[i = class][k = point]
    auto fetch = [&](int tile) {
        const long long p0 = (long long)tile * 16;
        const long long t = p0 * DP1 + lane;
        const bool ok = lane < 16 * DP1 && t < tok_end && tile < tiles;
        t_idx = ok ? idx[t] : -1;
        t_we = ok ? w[t] + delta_w[t] : 0.0f;
        t_gdw = ok ? g_delta_w[t] : 0.0f;
        const bool prow = p0 + i < n && tile < tiles;
        const float* grow = grad_logits + (p0 + i) * C;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int c = ks * 4 + q;
            a1[ks] = (prow && c < C) ? grow[c] : 0.0f;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const long long p = p0 + ks * 4 + q;
#pragma unroll
            for (int ct = 0; ct < CTL; ++ct) {
                const int c = ct * 16 + i;
                a3[ks][ct] = (p < n && c < C && tile < tiles) ? grad_logits[p * C + c] : 0.0f;
            }
        }
    };
    const int stride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    fetch(tile);
    for (; tile < tiles; tile += stride) {
        const long long p0 = (long long)tile * 16;
        const float t_gdw_cur = t_gdw;
        if (lane < 16 * DP1) {
            s_idx[lane] = t_idx;
            s_we[lane] = t_we;
            if (p0 * DP1 + lane < tok_end) w_eff[p0 * DP1 + lane] = t_we;
        }
        // (1) gh = g @ W
        {
            floatx4 gh[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gh[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {
                if (ks < KS) {  // wave-uniform
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        gh[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], s_wb[(ks * NT + nt) * 64 + lane], gh[nt], 0, 0, 0);
                }
            }
            // D: column = lane & 15 (channel), row = q * 4 + reg (point)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_gh[(q * 4 + r) * SG + nt * 16 + i] = gh[nt][r];
        }
        float b3[4][CTL];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < CTL; ++ct) b3[ks][ct] = a3[ks][ct];
        ln_wave_sync();
        // (2) gather: 8 lanes per point, U float4 per lane and vertex row; two passes of 8 points, all row reads of a tile in flight
        // together (U <= 2) before the next tile's prefetch is issued behind them
        {
            const int s = lane & 7;
            int rows[2][DP1];
            float4 x[2][DP1][U];
            auto issue = [&](int ps) {
                const int lp = ps * 8 + (lane >> 3);
#pragma unroll
                for (int r = 0; r < DP1; ++r) {
                    rows[ps][r] = s_idx[lp * DP1 + r];
                    const float4* src = reinterpret_cast<const float4*>(values + (size_t)(rows[ps][r] >= 0 ? rows[ps][r] : 0) * V);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
#if LN_SCB_PROBE & 1
                        x[ps][r][u] = make_float4(1.f, 2.f, float(rows[ps][r]), 3.f);
#else
                        x[ps][r][u] = src[s + 8 * u];
#endif
                    }
                }
            };
            auto finish = [&](int ps) {
                const int lp = ps * 8 + (lane >> 3);
                const long long p = p0 + lp;
                float dot[DP1];
#pragma unroll
                for (int r = 0; r < DP1; ++r) dot[r] = 0.0f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float* cell = s_gh + lp * SG + 4 * (s + 8 * u);
                    const float4 g4 = *reinterpret_cast<const float4*>(cell);
                    float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int r = 0; r < DP1; ++r) {
                        const bool ok = rows[ps][r] >= 0;
                        const float wt = s_we[lp * DP1 + r];
                        float4 xv = x[ps][r][u];
                        xv.x = ok ? xv.x : 0.0f; xv.y = ok ? xv.y : 0.0f; xv.z = ok ? xv.z : 0.0f; xv.w = ok ? xv.w : 0.0f;
                        h.x = h.x + xv.x * wt; h.y = h.y + xv.y * wt; h.z = h.z + xv.z * wt; h.w = h.w + xv.w * wt;
                        dot[r] = dot[r] + (xv.x * g4.x + xv.y * g4.y + xv.z * g4.z + xv.w * g4.w);
                    }
                    *reinterpret_cast<float4*>(cell) = h;  // in place: gh of this cell has been read by this lane only
#if !(LN_SCB_PROBE & 2)
                    if (p < n) reinterpret_cast<float4*>(grad_sliced + (size_t)p * V)[s + 8 * u] = g4;
#endif
                }
#pragma unroll
                for (int r = 0; r < DP1; ++r) {  // sum over the 8 lanes of the point: DPP within the quad, then across the two quads
                    float d = dot[r];
                    d += ln_dpp<0xB1>(d);   // quad_perm [1,0,3,2]
                    d += ln_dpp<0x4E>(d);   // quad_perm [2,3,0,1]
                    d += ln_dpp<0x141>(d);  // row_half_mirror: lane i <-> 7 - i of each group of 8
                    if (s == r) s_dw[lp * DP1 + r] = rows[ps][r] >= 0 ? d : 0.0f;
                }
            };
            issue(0);
            if (U <= 2) issue(1);
            fetch(tile + stride);  // (behind this tile's row reads; past the last tile: zeros, nothing is loaded)
            finish(0);
            if (U > 2) issue(1);
            finish(1);
        }
        ln_wave_sync();
        // accumulated into (Lattice.cu:1091-1115): old value fetched with the tile's tokens, one lane-linear store
        if (lane < 16 * DP1 && p0 * DP1 + lane < tok_end) g_delta_w[p0 * DP1 + lane] = t_gdw_cur + s_dw[lane];
        // (3) gW += g^T @ [h | 1]: A[i = class][k = point], B[k = point][j = channel]
#if !(LN_SCB_PROBE & 4)
#pragma unroll
#endif
        for (int ks = (LN_SCB_PROBE & 4) ? 3 : 0; ks < 4; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float b = s_gh[(ks * 4 + q) * SG + nt * 16 + i];
#pragma unroll
                for (int ct = 0; ct < CTL; ++ct) acc_w[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b3[ks][ct], b, acc_w[ct][nt], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < CTL; ++ct) acc_b[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b3[ks][ct], 1.0f, acc_b[ct], 0, 0, 0);
        }
        ln_wave_sync();  // the next tile overwrites s_gh / the token arrays
    }
    // one slab per workgroup: the four waves' accumulators summed through LDS (wave w adds after wave w - 1)
    __syncthreads();
    float* s_slab = s_wave;  // [C * V + C] floats <= 4 * WAVE_LDS = 64 (V + 4) (C <= 32)
    const int CV = C * V;
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int ct = 0; ct < CTL; ++ct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = ct * 16 + q * 4 + r;
                    if (c < C) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            float* d = s_slab + c * V + nt * 16 + i;
                            *d = (wv ? *d : 0.0f) + acc_w[ct][nt][r];
                        }
                        if (i == 0) s_slab[CV + c] = (wv ? s_slab[CV + c] : 0.0f) + acc_b[ct][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    float* slab = slabs + (size_t)blockIdx.x * (CV + C);
    for (int e = threadIdx.x; e < CV + C; e += 256) slab[e] = s_slab[e];
}

int ln_sc_backward_wave_grid(int n) {
    const int g = ln_div_up(ln_div_up(n, 16), 4);
    return g > 512 ? 512 : g;  // two workgroups per CU (the register budget admits two waves per SIMD)
}

int ln_sc_backward_wave(const float* grad_logits, const float* values, const float* delta_w, const float* lin_w, const int* idx,
                        const float* w, int n, int pos_dim, int val_dim, int nr_classes, float* g_delta_w, float* grad_sliced,
                        float* w_eff, float* slabs, int* grid_out, hipStream_t st) {
    // returns 1 when launched (one slab of C*V + C floats per workgroup, *grid_out of them), 0 when the shape is not covered
    if (val_dim % 32 != 0 || val_dim > 128 || nr_classes > 32 || (pos_dim != 2 && pos_dim != 3)) return 0;
    if (((reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(grad_sliced)) & 15) != 0) return 0;
    const int grid = ln_sc_backward_wave_grid(n);
    const int ks = (nr_classes + 3) / 4, nt = val_dim / 16;
    const size_t lds = sizeof(float) * ((size_t)ks * nt * 64 + 4 * (size_t)(16 * (val_dim + 4)));
    const int ctl = nr_classes > 16 ? 2 : 1;
#define LN_SCW_BWD(DP1, U, CTL)                                                                                                       \
    LN_LAUNCH("k_slice_classify_backward", (k_slice_classify_backward_wave<DP1, U, CTL>), dim3(grid), dim3(256), lds, st, grad_logits, \
              values, delta_w, lin_w, idx, w, n, nr_classes, g_delta_w, grad_sliced, w_eff, slabs)
#define LN_SCW_BWD_U(DP1, CTL)                    \
    switch (val_dim / 32) {                       \
        case 1: LN_SCW_BWD(DP1, 1, CTL); break;   \
        case 2: LN_SCW_BWD(DP1, 2, CTL); break;   \
        case 3: LN_SCW_BWD(DP1, 3, CTL); break;   \
        default: LN_SCW_BWD(DP1, 4, CTL); break;  \
    }
    if (pos_dim == 3) {
        if (ctl == 2) { LN_SCW_BWD_U(4, 2) } else { LN_SCW_BWD_U(4, 1) }
    } else {
        if (ctl == 2) { LN_SCW_BWD_U(3, 2) } else { LN_SCW_BWD_U(3, 1) }
    }
#undef LN_SCW_BWD_U
#undef LN_SCW_BWD
    *grid_out = grid;
    return 1;
}
