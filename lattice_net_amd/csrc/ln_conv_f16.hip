// Half-precision feature path of the lattice convolution (BASELINE.json config 5 / SURVEY.md §8d C5: "features fp16,
// accumulate fp32"): the same gather-GEMM as ln_conv.hip on gfx950's v_mfma_f32_16x16x32_f16 (K = 32 per instruction: twice the
// flop per issue slot of the CDNA3-era 16x16x16 form, which remains for gathered widths that are not multiples of 32) — fp16 operands, fp32 accumulators
// — for the forward pass and the gradient wrt the values, and the filter gradient accumulated in fp32 (fmaf) from fp16
// activations and gradients.
//
// Forward:  out[m, :] = sum_e values[nbr[m,e], :] @ W[e*V:(e+1)*V, :]        values / W / out in fp16
//   * a wave owns 16 vertices; lane (i = lane&15, q = lane>>4) reads the q-th quarter of neighbour row nbr[m0+i, e]
//     as V/4 contiguous halfs.  One MFMA consumes 16 k-values: lane group q supplies its halfs [4s, 4s+4) at step s, so
//     the K order inside a neighbour is permuted (k = q*V/4 + 4s + j), which only reorders the fp32 accumulation.
//   * W_e is staged per slot into LDS in that fragment order: 8 bytes per lane per MFMA, lane-linear.
#include "ln_common.h"

#include <hip/hip_fp16.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
template <int V, int NT, bool FLIP, bool WT>
__global__ void __launch_bounds__(256)
    k_conv_mfma_f16(const int* __restrict__ nbr, const _Float16* __restrict__ values, const _Float16* __restrict__ filter, int m, int E,
                    _Float16* __restrict__ out, int f_total, int f_off) {
    constexpr int F = 16 * NT;
    constexpr int KQ = V / 4;            // halfs per lane per neighbour
    constexpr bool K32 = V % 32 == 0;    // v_mfma_f32_16x16x32_f16: lane group q supplies 8 halfs per step
    constexpr int KS = K32 ? 8 : 4;      // halfs per lane and step
    constexpr int S = KQ / KS;           // MFMA steps per neighbour
    static_assert(V % 16 == 0, "V must be a multiple of 16");
    __shared__ __attribute__((aligned(16))) _Float16 s_b[V * F];  // [((s*NT + nt)*64 + lane)*KS + j]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int m0 = blockIdx.x * 64 + wave * 16;
    const int my_row = m0 + i;

    floatx4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int e = 0; e < E; ++e) {
        _Float16 a[KQ];
        const int es = (FLIP && e < E - 1) ? (e ^ 1) : e;
        const int nb = (my_row < m) ? nbr[(size_t)my_row * E + es] : -1;
        if (nb >= 0) {
            const halfx4* src = reinterpret_cast<const halfx4*>(values + (size_t)nb * V + q * KQ);
#pragma unroll
            for (int s = 0; s < KQ / 4; ++s) {
                const halfx4 w4 = src[s];
                a[4 * s] = w4[0], a[4 * s + 1] = w4[1], a[4 * s + 2] = w4[2], a[4 * s + 3] = w4[3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < KQ; ++k) a[k] = (_Float16)0;
        }
        __syncthreads();  // previous iteration's reads of s_b are done
        for (int x = tid; x < V * F; x += 256) {
            const int k = WT ? (x % V) : (x / F);
            const int f = WT ? (x / V) : (x - k * F);
            const int qq = k / KQ;
            const int r = k - qq * KQ;
            const size_t src = WT ? ((size_t)e * f_total + f_off + f) * V + k : ((size_t)e * V + k) * f_total + f_off + f;
            s_b[((((r / KS) * NT) + (f >> 4)) * 64 + qq * 16 + (f & 15)) * KS + (r % KS)] = filter[src];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (K32) {
                    const halfx8 b = *reinterpret_cast<const halfx8*>(s_b + ((s * NT + nt) * 64 + lane) * 8);
                    const halfx8 a8 = {a[8 * s], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3], a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7]};
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b, acc[nt], 0, 0, 0);
                } else {
                    const halfx4 b = *reinterpret_cast<const halfx4*>(s_b + ((s * NT + nt) * 64 + lane) * 4);
                    const halfx4 a4 = {a[4 * s], a[4 * s + 1], a[4 * s + 2], a[4 * s + 3]};
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) out[(size_t)row * f_total + f_off + nt * 16 + i] = (_Float16)acc[nt][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// pos_dim 3 (E = 9), V and F in {32, 64}: the whole filter bank staged ONCE per workgroup, in MFMA fragment order, for T sub-tiles
// of 64 vertices (blockDim.x = 256 T, the shape of ln_conv.hip's k_conv_forward_b3): 72 KB of LDS at V = F = 64, two workgroups
// per CU.  The per-slot kernel above re-stages 8 KB per slot with two barriers and 2-byte global loads, and starts each
// neighbour-row gather only after the previous slot's matrix instructions: 42 us at C5 (70 k vertices) against 2 x 21 here.
//   fragment unit (e, s, nt, lane = (qq, fi)) = the 8 halfs W_e[k = qq * V/4 + 8 s + j][f = 16 nt + fi], j = 0..7 (16 bytes: one
//                     operand of v_mfma_f32_16x16x32_f16)
//   bank [E*V, F]   : a thread reads 8 rows x 8 columns (eight 16-byte loads) and writes the 8 units of its columns;
//   bank^T [E*F, V] : a thread reads 8 consecutive k of one column (one 16-byte load) = one unit.
// ------------------------------------------------------------------------------------------
#ifndef LN_F16_LINE
#define LN_F16_LINE 1  // 0: fragment-shaped gathers at 64 channels too (A/B)
#endif
template <int V, int NT, bool FLIP, bool WT>
__global__ void __launch_bounds__(1024)
    k_conv_f16_tiled(const int* __restrict__ nbr, const _Float16* __restrict__ values, const _Float16* __restrict__ filter, int m,
                     _Float16* __restrict__ out) {
    constexpr int E = 9, F = 16 * NT, KQ = V / 4, S = KQ / 8, G8 = KQ / 8;
    static_assert(V % 32 == 0, "a lane's quarter row is read in 16-byte words, each one operand of a K = 32 step");
    __shared__ __attribute__((aligned(16))) halfx8 s_frag[E * S * NT * 64];
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int m0 = blockIdx.x * (nthreads / 4) + (tid >> 6) * 16;
    const int my_row = m0 + i;
    // V = 64 (128-byte rows): line-shaped gathers as in k_conv_forward_b3 (ln_conv.hip) — lane l loads piece l & 7 (16 bytes) of rows
    // l >> 3 and 8 + (l >> 3) of the wave's 16, and the wave re-shapes its 2 KB through a private, XOR-swizzled LDS region into the
    // fragment shape (lane (i, q) = pieces 2q, 2q + 1 of row i).  The fragment-shaped loads gathered 81 MB in 23 us = 3.5 TB/s at C5:
    // the rate of that shape whatever the hit rate (tools/probes/gather_layout_probe.cpp).
    constexpr bool LINE = (V == 64) && LN_F16_LINE;
    constexpr int DEPTH = 4;  // ring of gathered quarter rows: DEPTH - 1 gathers in flight
    __shared__ __attribute__((aligned(16))) halfx8 s_x[LINE ? 16 : 1][16 * 8];
    __shared__ int s_nbr[LINE ? 256 * E : 1];
    int nb[LINE ? 2 : 1][E];
    halfx8 g[DEPTH][G8];
    const int wv_ = tid >> 6, lr = lane >> 3;
    auto fsw = [](int r) -> int { return ((r >> 1) & 7) ^ (((r >> 2) & 1) << 1); };
    if constexpr (LINE) {
        const int rows_wg = nthreads / 4;
        const size_t g0 = (size_t)blockIdx.x * rows_wg * E, g_end = (size_t)m * E;
        for (int x = tid; x < rows_wg * E; x += nthreads) s_nbr[x] = (g0 + x < g_end) ? nbr[g0 + x] : -1;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) nb[j][e] = s_nbr[(wv_ * 16 + 8 * j + lr) * E + ((FLIP && e < E - 1) ? (e ^ 1) : e)];
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) nb[0][e] = (my_row < m) ? nbr[(size_t)my_row * E + ((FLIP && e < E - 1) ? (e ^ 1) : e)] : -1;
    }
    auto gather = [&](int e, halfx8 (&dst)[G8]) {
        if constexpr (LINE) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                dst[j] = *reinterpret_cast<const halfx8*>(values + (size_t)(nb[j][e] >= 0 ? nb[j][e] : 0) * V + (lane & 7) * 8);
        } else {
            const halfx8* src = reinterpret_cast<const halfx8*>(values + (size_t)(nb[0][e] >= 0 ? nb[0][e] : 0) * V + q * KQ);
#pragma unroll
            for (int k = 0; k < G8; ++k) dst[k] = src[k];
        }
    };
#pragma unroll
    for (int k = 0; k < DEPTH - 1; ++k) gather(k, g[k]);
    if constexpr (!WT) {
        constexpr int ITEMS = E * (V / 8) * (F / 8);
        for (int x = tid; x < ITEMS; x += nthreads) {
            const int fo = x % (F / 8);
            const int t2 = x / (F / 8);
            const int k8 = t2 % (V / 8);
            const int e = t2 / (V / 8);
            const int k0 = k8 * 8, f0 = fo * 8;
            halfx8 r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const halfx8*>(filter + ((size_t)(e * V + k0 + j)) * F + f0);
            const int qq = k0 / KQ, s = (k0 - qq * KQ) >> 3;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int f = f0 + c;
                s_frag[((e * S + s) * NT + (f >> 4)) * 64 + qq * 16 + (f & 15)] =
                    halfx8{r[0][c], r[1][c], r[2][c], r[3][c], r[4][c], r[5][c], r[6][c], r[7][c]};
            }
        }
    } else {
        constexpr int ITEMS = E * F * (V / 8);
        for (int x = tid; x < ITEMS; x += nthreads) {
            const int ko = x % (V / 8);
            const int t2 = x / (V / 8);
            const int f = t2 % F;
            const int e = t2 / F;
            const int k0 = ko * 8;
            const int qq = k0 / KQ, s = (k0 - qq * KQ) >> 3;
            s_frag[((e * S + s) * NT + (f >> 4)) * 64 + qq * 16 + (f & 15)] = *reinterpret_cast<const halfx8*>(filter + ((size_t)(e * F + f)) * V + k0);
        }
    }
    floatx4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e + DEPTH - 1 < E) gather(e + DEPTH - 1, g[(e + DEPTH - 1) % DEPTH]);
        halfx8 ge[G8];
        bool present;
        if constexpr (LINE) {
            // rows of absent neighbours are zeroed by the lane that loaded them; LDS operations of one wave execute in program order
#pragma unroll
            for (int j = 0; j < 2; ++j)
                s_x[wv_][(8 * j + lr) * 8 + ((lane & 7) ^ fsw(8 * j + lr))] = (nb[j][e] >= 0) ? g[e % DEPTH][j] : halfx8{0, 0, 0, 0, 0, 0, 0, 0};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            ge[0] = s_x[wv_][i * 8 + ((2 * q) ^ fsw(i))];
            ge[1] = s_x[wv_][i * 8 + ((2 * q + 1) ^ fsw(i))];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            present = true;
        } else {
#pragma unroll
            for (int k = 0; k < G8; ++k) ge[k] = g[e % DEPTH][k];
            present = nb[0][e] >= 0;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            halfx8 a = ge[s];
            if (!present) a = halfx8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, s_frag[((e * S + s) * NT + nt) * 64 + lane], acc[nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) out[(size_t)row * F + nt * 16 + i] = (_Float16)acc[nt][r];
        }
}

// sub-tiles per workgroup of k_conv_f16_tiled: the launch runs in rounds of 256 CUs x (workgroups that fit a CU's LDS); the T with the
// cheapest rounds(T) * T wins, larger T (fewer bank stagings) on ties.  LN_F16_T = 1..4 forces it (experiments).
static int ln_f16_subtiles(int m, size_t lds_bytes) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("LN_F16_T");
        forced = e ? atoi(e) : 0;
    }
    if (forced >= 1 && forced <= 4) return forced;
    const int per_cu = lds_bytes * 2 <= 150 * 1024 ? 2 : 1;
    const int tiles = (m + 63) / 64;
    int best = 1, best_cost = 1 << 30;
    for (int t = 1; t <= 4; ++t) {
        if (per_cu * t * 4 > 32) break;  // wave slots of a CU
        // (one workgroup per CU: a single sub-tile would leave the CU with four waves — 39 us instead of 21 at C5 — unless the lattice
        // is too small to give every CU more)
        if (per_cu == 1 && t < 3 && tiles >= 3 * 256) continue;
        const int wgs = (tiles + t - 1) / t;
        const int cost = ((wgs + 256 * per_cu - 1) / (256 * per_cu)) * t;
        if (cost <= best_cost) {
            best = t;
            best_cost = cost;
        }
    }
    return best;
}

template <bool FLIP, bool WT>
static bool ln_conv_f16_tiled(const int* nbr, const _Float16* values, const _Float16* filter, int m, int E, int val_dim, int nr_filters,
                              _Float16* out, hipStream_t st) {
    static const bool off = getenv("LN_F16_PER_SLOT") != nullptr;  // A/B: the per-slot kernel for every shape
    if (off || E != 9 || ((reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(filter)) & 15) != 0) return false;
#define LN_F16_TILED(VV, NN)                                                                                                         \
    if (val_dim == VV && nr_filters == 16 * NN) {                                                                                    \
        /* (at 64 channels: + the re-shaping regions of up to 16 waves and the ids, see the kernel) */                               \
        const int t = ln_f16_subtiles(m, (size_t)9 * VV * 16 * NN * 2 + ((VV == 64 && LN_F16_LINE) ? 16 * 2048 + 256 * 9 * 4 : 0));  \
        LN_LAUNCH("k_conv_mfma_f16", (k_conv_f16_tiled<VV, NN, FLIP, WT>), dim3(ln_div_up(m, 64 * t)), dim3(256 * t), 0, st, nbr, values, \
                  filter, m, out);                                                                                                   \
        return true;                                                                                                                 \
    }
    LN_F16_TILED(32, 2) LN_F16_TILED(64, 4) LN_F16_TILED(32, 4) LN_F16_TILED(64, 2)
#undef LN_F16_TILED
    return false;
}

// any shape: one thread per output element, fp32 accumulation
__global__ void __launch_bounds__(256)
    k_conv_generic_f16(const int* __restrict__ nbr, const _Float16* __restrict__ values, const _Float16* __restrict__ filter, long long work,
                       int E, int V, int F, int flip, int wt, _Float16* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long mrow = g / F;
    const int f = int(g - mrow * F);
    float acc = 0.0f;
    for (int e = 0; e < E; ++e) {
        const int nb = nbr[mrow * E + ((flip && e < E - 1) ? (e ^ 1) : e)];
        if (nb < 0) continue;
        const _Float16* vr = values + (size_t)nb * V;
        for (int v = 0; v < V; ++v) {
            const float wv = wt ? (float)filter[((size_t)e * F + f) * V + v] : (float)filter[((size_t)e * V + v) * F + f];
            acc = fmaf((float)vr[v], wv, acc);
        }
    }
    out[g] = (_Float16)acc;
}

template <int V, bool FLIP, bool WT>
static void ln_conv_f16_launch_v(int nr_filters, const int* nbr, const _Float16* values, const _Float16* filter, int m, int E, _Float16* out,
                                 hipStream_t st) {
    const dim3 grid(ln_div_up(m, 64)), block(256);
    constexpr int NT_MAX = (V * 128 * 2 <= 32 * 1024) ? 8 : 4;  // LDS of one slot's filter slice
    int f_off = 0;
    while (f_off < nr_filters) {
        const int left = (nr_filters - f_off) / 16;
        if (NT_MAX >= 8 && left >= 8) {
            if constexpr (NT_MAX >= 8) LN_LAUNCH("k_conv_mfma_f16", (k_conv_mfma_f16<V, 8, FLIP, WT>), grid, block, 0, st, nbr, values, filter, m, E, out, nr_filters, f_off);
            f_off += 128;
        } else if (left >= 4) {
            LN_LAUNCH("k_conv_mfma_f16", (k_conv_mfma_f16<V, 4, FLIP, WT>), grid, block, 0, st, nbr, values, filter, m, E, out, nr_filters, f_off);
            f_off += 64;
        } else if (left >= 2) {
            LN_LAUNCH("k_conv_mfma_f16", (k_conv_mfma_f16<V, 2, FLIP, WT>), grid, block, 0, st, nbr, values, filter, m, E, out, nr_filters, f_off);
            f_off += 32;
        } else {
            LN_LAUNCH("k_conv_mfma_f16", (k_conv_mfma_f16<V, 1, FLIP, WT>), grid, block, 0, st, nbr, values, filter, m, E, out, nr_filters, f_off);
            f_off += 16;
        }
    }
}

template <bool FLIP, bool WT>
static int ln_conv_f16_dispatch(const int* nbr, const _Float16* values, const _Float16* filter, int m, int E, int val_dim, int nr_filters,
                                _Float16* out, hipStream_t st) {
    bool done = ln_conv_f16_tiled<FLIP, WT>(nbr, values, filter, m, E, val_dim, nr_filters, out, st);
    if (done) return ln_check_launch("ln_conv_forward_f16");
    if (nr_filters % 16 == 0 && (reinterpret_cast<uintptr_t>(values) & 7) == 0) {
        done = true;
        switch (val_dim) {
            case 16: ln_conv_f16_launch_v<16, FLIP, WT>(nr_filters, nbr, values, filter, m, E, out, st); break;
            case 32: ln_conv_f16_launch_v<32, FLIP, WT>(nr_filters, nbr, values, filter, m, E, out, st); break;
            case 64: ln_conv_f16_launch_v<64, FLIP, WT>(nr_filters, nbr, values, filter, m, E, out, st); break;
            case 96: ln_conv_f16_launch_v<96, FLIP, WT>(nr_filters, nbr, values, filter, m, E, out, st); break;
            case 128: ln_conv_f16_launch_v<128, FLIP, WT>(nr_filters, nbr, values, filter, m, E, out, st); break;
            case 256: ln_conv_f16_launch_v<256, FLIP, WT>(nr_filters, nbr, values, filter, m, E, out, st); break;
            default: done = false; break;
        }
    }
    if (!done) {
        const long long work = (long long)m * nr_filters;
        LN_LAUNCH("k_conv_generic_f16", k_conv_generic_f16, dim3(ln_div_up(work, 256)), dim3(256), 0, st, nbr, values, filter, work, E, val_dim,
                  nr_filters, FLIP ? 1 : 0, WT ? 1 : 0, out);
    }
    return ln_check_launch("ln_conv_forward_f16");
}

extern "C" int ln_conv_forward_f16(const int* nbr, const void* values_neigh, const void* filter, int m, int filter_extent, int val_dim,
                                   int nr_filters, int flags, void* out, void* stream) {
    LN_REQUIRE(m >= 0 && filter_extent >= 3 && val_dim >= 1 && nr_filters >= 1, LN_ERR_ARG, "ln_conv_forward_f16: bad sizes");
    LN_REQUIRE(m == 0 || (nbr && values_neigh && filter && out), LN_ERR_ARG, "ln_conv_forward_f16: null buffer");
    LN_REQUIRE((flags & ~3) == 0, LN_ERR_ARG, "ln_conv_forward_f16: unknown flags %d", flags);
    if (m == 0) return LN_OK;
    hipStream_t st = (hipStream_t)stream;
    const _Float16* v = static_cast<const _Float16*>(values_neigh);
    const _Float16* w = static_cast<const _Float16*>(filter);
    _Float16* o = static_cast<_Float16*>(out);
    switch (flags) {
        case 0: return ln_conv_f16_dispatch<false, false>(nbr, v, w, m, filter_extent, val_dim, nr_filters, o, st);
        case LN_CONV_FLIP_NEIGHBOURS: return ln_conv_f16_dispatch<true, false>(nbr, v, w, m, filter_extent, val_dim, nr_filters, o, st);
        case LN_CONV_TRANSPOSED_FILTER: return ln_conv_f16_dispatch<false, true>(nbr, v, w, m, filter_extent, val_dim, nr_filters, o, st);
        default: return ln_conv_f16_dispatch<true, true>(nbr, v, w, m, filter_extent, val_dim, nr_filters, o, st);
    }
}

// ------------------------------------------------------------------------------------------
// filter gradient from fp16 activations / gradients, accumulated in fp32:
//   grad_filter[e*V+v, f] = sum_m values[nbr[m,e], v] * grad_out[m, f]
// grid = (row chunks, E).  The contraction runs over lattice vertices (rows), so both MFMA operands are read
// "down the rows": a sub-tile of 64 rows is staged row-major in LDS and read with the transposing LDS load of gfx950
// (ds_read_b64_tr_b16: lane (i, q) receives 4 consecutive rows of column i; two of them = the 8 rows 8q..8q+7 of a 32-row step of
// v_mfma_f32_16x16x32_f16).  Each wave owns whole 16x16 output
// tiles D[v, f]; partial [V, F] blocks per row chunk go to slabs, summed by k_reduce_slabs4 / k_reduce_slabs (ln_conv.hip, deterministic).
// ------------------------------------------------------------------------------------------
#define LN_GF16_ROWS 512
#define LN_GF16_SUB 64
template <int VT, int FT>
__global__ void __launch_bounds__(256)
    k_grad_filter_mfma_f16(const int* __restrict__ nbr, const _Float16* __restrict__ values, const _Float16* __restrict__ grad_out, int m,
                           int E, float* __restrict__ partial, int v_total, int v_off, int f_total, int f_off) {
    constexpr int V = VT * 16, F = FT * 16;
    constexpr int TILES = VT * FT;
    constexpr int TPW = (TILES + 3) / 4;
    // sub-tiles are staged ROW-major (one 8-byte LDS store per 4 channels; rows padded by 16 halfs, see SA below) and read "down
    // the rows" with ds_read_b64_tr_b16: lane i of a 16-lane
    // group passes the address of row i >> 2, columns 4 (i & 3).. of a 4 x 16 block and receives rows 0..3 of column i
    // (tools/probes/tr_read_probe.cpp) - the MFMA operand of a contraction over rows.  (The first version stored every half
    // on its own into transposed arrays: 256 two-byte LDS stores per thread and chunk.)
    constexpr int SA = V + 16, SG = F + 16;  // row stride = an odd multiple of 8 dwords: the 8 consecutive rows a 32-lane group of a transposing read touches fall on disjoint banks
    __shared__ __attribute__((aligned(16))) _Float16 s_a[LN_GF16_SUB * SA];
    __shared__ __attribute__((aligned(16))) _Float16 s_g[LN_GF16_SUB * SG];
    typedef short shortx4 __attribute__((ext_vector_type(4)));
    typedef short shortx8 __attribute__((ext_vector_type(8)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int e = blockIdx.y;
    const int chunk_begin = blockIdx.x * LN_GF16_ROWS;
    const int chunk_end = min(chunk_begin + LN_GF16_ROWS, m);
    floatx4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    // the rows of the NEXT sub-tile are fetched into registers (neighbour id -> row: two dependent round trips) while the matrix
    // instructions of the current one run: VT + FT 8-byte words per thread
    halfx4 ra[VT], rg[FT];
    auto fetch = [&](int sub) {
#pragma unroll
        for (int k = 0; k < VT; ++k) {
            const int x = tid + 256 * k;
            const int r = x / (V / 4), c4 = x - r * (V / 4);
            const int row = sub + r;
            const int nb = row < chunk_end ? nbr[(size_t)row * E + e] : -1;
            ra[k] = halfx4{0, 0, 0, 0};
            if (nb >= 0) ra[k] = *reinterpret_cast<const halfx4*>(values + (size_t)nb * v_total + v_off + c4 * 4);
        }
#pragma unroll
        for (int k = 0; k < FT; ++k) {
            const int x = tid + 256 * k;
            const int r = x / (F / 4), c4 = x - r * (F / 4);
            const int row = sub + r;
            rg[k] = halfx4{0, 0, 0, 0};
            if (row < chunk_end) rg[k] = *reinterpret_cast<const halfx4*>(grad_out + (size_t)row * f_total + f_off + c4 * 4);
        }
    };
    if (chunk_begin < chunk_end) fetch(chunk_begin);
    for (int sub = chunk_begin; sub < chunk_end; sub += LN_GF16_SUB) {
        __syncthreads();  // the previous sub-tile's reads are done
#pragma unroll
        for (int k = 0; k < VT; ++k) {
            const int x = tid + 256 * k;
            const int r = x / (V / 4), c4 = x - r * (V / 4);
            *reinterpret_cast<halfx4*>(s_a + r * SA + c4 * 4) = ra[k];
        }
#pragma unroll
        for (int k = 0; k < FT; ++k) {
            const int x = tid + 256 * k;
            const int r = x / (F / 4), c4 = x - r * (F / 4);
            *reinterpret_cast<halfx4*>(s_g + r * SG + c4 * 4) = rg[k];
        }
        __syncthreads();
        if (sub + LN_GF16_SUB < chunk_end) fetch(sub + LN_GF16_SUB);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = wave + 4 * t;
            if (tile < TILES) {
                const int vt = tile / FT, ft = tile - vt * FT;
                // rows 4q..4q+3 and 16+4q..16+4q+3 of the 32-row step (any order of the contraction that both operands share)
                const _Float16* pa = s_a + (4 * q + (i >> 2)) * SA + vt * 16 + 4 * (i & 3);
                const _Float16* pg = s_g + (4 * q + (i >> 2)) * SG + ft * 16 + 4 * (i & 3);
#pragma unroll
                for (int s = 0; s < LN_GF16_SUB / 32; ++s) {
                    const shortx4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(pa + 32 * s * SA));
                    const shortx4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(pa + (32 * s + 16) * SA));
                    const shortx4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(pg + 32 * s * SG));
                    const shortx4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(pg + (32 * s + 16) * SG));
                    const shortx8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const shortx8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b), acc[t], 0, 0, 0);
                }
            }
        }
    }
    float* dst = partial + ((size_t)blockIdx.x * E + e) * ((size_t)v_total * f_total) + (size_t)v_off * f_total + f_off;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + 4 * t;
        if (tile < TILES) {
            const int vt = tile / FT, ft = tile - vt * FT;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)(vt * 16 + q * 4 + r) * f_total + ft * 16 + i] = acc[t][r];
        }
    }
}

// any shape (fallback): lanes own (v, f) pairs, rows staged through LDS as floats
#define LN_GF16_GSUB 32
__global__ void __launch_bounds__(256)
    k_grad_filter_f16(const int* __restrict__ nbr, const _Float16* __restrict__ values, const _Float16* __restrict__ grad_out, int m, int E,
                      int V, int F, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float s_mem16[];
    float* s_a = s_mem16;                    // [SUB, V]
    float* s_g = s_a + LN_GF16_GSUB * V;     // [SUB, F]
    const int tid = threadIdx.x;
    const int e = blockIdx.y;
    const int chunk_begin = blockIdx.x * LN_GF16_ROWS;
    const int chunk_end = min(chunk_begin + LN_GF16_ROWS, m);
    const int VF = V * F;
    constexpr int MAXP = 16;                 // (v, f) pairs per thread; blockIdx.z walks V*F in chunks of 4096 pairs
    float acc[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) acc[k] = 0.f;
    for (int sub = chunk_begin; sub < chunk_end; sub += LN_GF16_GSUB) {
        __syncthreads();
        for (int x = tid; x < LN_GF16_GSUB * V; x += 256) {
            const int r = x / V, c = x - r * V;
            const int row = sub + r;
            const int nb = row < chunk_end ? nbr[(size_t)row * E + e] : -1;
            s_a[x] = nb >= 0 ? (float)values[(size_t)nb * V + c] : 0.f;
        }
        for (int x = tid; x < LN_GF16_GSUB * F; x += 256) {
            const int r = x / F, c = x - r * F;
            const int row = sub + r;
            s_g[x] = row < chunk_end ? (float)grad_out[(size_t)row * F + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int j = blockIdx.z * (MAXP * 256) + tid + k * 256;
            if (j < VF) {
                const int v = j / F, f = j - v * F;
                float a = acc[k];
#pragma unroll 8
                for (int r = 0; r < LN_GF16_GSUB; ++r) a = fmaf(s_a[r * V + v], s_g[r * F + f], a);
                acc[k] = a;
            }
        }
    }
    float* dst = partial + ((size_t)blockIdx.x * E + e) * VF;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int j = blockIdx.z * (MAXP * 256) + tid + k * 256;
        if (j < VF) dst[j] = acc[k];
    }
}

extern "C" size_t ln_conv_grad_filter_f16_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters) {
    if (m <= 0) return 256;
    return (size_t)ln_div_up(m, LN_GF16_ROWS) * filter_extent * val_dim * nr_filters * sizeof(float) + 256;
}

extern "C" int ln_conv_grad_filter_f16(const int* nbr, const void* values_neigh, const void* grad_out, int m, int filter_extent, int val_dim,
                                       int nr_filters, float* grad_filter, void* workspace, size_t workspace_bytes, void* stream) {
    LN_REQUIRE(m >= 0 && filter_extent >= 3 && val_dim >= 1 && nr_filters >= 1, LN_ERR_ARG, "ln_conv_grad_filter_f16: bad sizes");
    LN_REQUIRE(grad_filter && (m == 0 || (nbr && values_neigh && grad_out)), LN_ERR_ARG, "ln_conv_grad_filter_f16: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const int total = filter_extent * val_dim * nr_filters;
    if (m == 0) {
        (void)ln_zero_async(grad_filter, (size_t)total * sizeof(float), st);
        return ln_check_launch("ln_conv_grad_filter_f16");
    }
    LN_REQUIRE(workspace && workspace_bytes >= ln_conv_grad_filter_f16_workspace_bytes(m, filter_extent, val_dim, nr_filters), LN_ERR_WORKSPACE,
               "ln_conv_grad_filter_f16: workspace too small");
    const int chunks = ln_div_up(m, LN_GF16_ROWS);
    float* partial = static_cast<float*>(workspace);
    const bool aligned = ((reinterpret_cast<uintptr_t>(values_neigh) | reinterpret_cast<uintptr_t>(grad_out)) & 7) == 0;
    if (val_dim % 16 == 0 && nr_filters % 16 == 0 && aligned) {
        const dim3 grid(chunks, filter_extent), block(256);
        const _Float16* vv = static_cast<const _Float16*>(values_neigh);
        const _Float16* gg = static_cast<const _Float16*>(grad_out);
        for (int v_off = 0; v_off < val_dim;) {
            const int vleft = (val_dim - v_off) / 16;
            const int vt = vleft >= 4 ? 4 : (vleft >= 2 ? 2 : 1);
            for (int f_off = 0; f_off < nr_filters;) {
                const int fleft = (nr_filters - f_off) / 16;
                const int ft = fleft >= 4 ? 4 : (fleft >= 2 ? 2 : 1);
#define LN_GF16_CASE(A, B)                                                                                                          \
    if (vt == A && ft == B)                                                                                                         \
        LN_LAUNCH("k_grad_filter_mfma_f16", (k_grad_filter_mfma_f16<A, B>), grid, block, 0, st, nbr, vv, gg, m, filter_extent, partial, val_dim, \
                  v_off, nr_filters, f_off);
                LN_GF16_CASE(1, 1) LN_GF16_CASE(1, 2) LN_GF16_CASE(1, 4) LN_GF16_CASE(2, 1) LN_GF16_CASE(2, 2) LN_GF16_CASE(2, 4)
                LN_GF16_CASE(4, 1) LN_GF16_CASE(4, 2) LN_GF16_CASE(4, 4)
#undef LN_GF16_CASE
                f_off += ft * 16;
            }
            v_off += vt * 16;
        }
    } else {
        const size_t lds = sizeof(float) * LN_GF16_GSUB * ((size_t)val_dim + nr_filters);
        LN_REQUIRE(lds <= 64 * 1024, LN_ERR_UNSUPPORTED, "ln_conv_grad_filter_f16: val_dim + nr_filters = %d too large", val_dim + nr_filters);
        LN_LAUNCH("k_grad_filter_f16", k_grad_filter_f16, dim3(chunks, filter_extent, ln_div_up((long long)val_dim * nr_filters, 4096)), dim3(256),
                  lds, st, nbr, static_cast<const _Float16*>(values_neigh), static_cast<const _Float16*>(grad_out), m, filter_extent, val_dim,
                  nr_filters, partial);
    }
    (void)ln_reduce_slabs_async(partial, chunks, total, grad_filter, st);  // (147 slabs of 147 KB at C5: one serial pass per output took 35 us)
    return ln_check_launch("ln_conv_grad_filter_f16");
}
