// Lattice convolution as a gather-GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), and
// the filter gradient as the transposed gather-GEMM.  Replaces im2row + Tensor::mm
// (reference Lattice.cu:454-462, lattice_funcs.py:298-302) without ever materialising the
// [M, E*V] rowified tensor in HBM.
//
// Forward:  out[m, :] = sum_e values[nbr[m,e], :] @ W[e*V:(e+1)*V, :]
//   * a wave owns 16 lattice vertices (MFMA rows); 4 waves per workgroup.
//   * A operand straight from HBM/L2 into registers: lane (i = lane&15, q = lane>>4) reads the
//     q-th quarter of neighbour row nbr[m0+i, e] as contiguous floats (the K order inside a
//     neighbour is permuted to k' = q*KQ + kk, which only reorders an exact-arithmetic sum).
//   * B operand (the filter slice W_e) is staged once per workgroup into LDS in fragment order,
//     so every ds_read is lane-linear and conflict-free.
// Precision.  The fp32-input kernels (k_conv_mfma_full, k_conv_mfma, k_grad_filter_mfma, k_conv_backward_fused) accumulate fp32 products
// in fp32: fmaf chains in a fixed order.  The "bf16x3" kernels (k_conv_forward_b3, k_conv_mfma_b3, k_conv_backward_fused_b3: the default
// at V = F = 32 and from 4096 vertices on) split every operand EXACTLY into three bf16 parts and accumulate SIX of the nine cross
// products in fp32 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1); the three dropped ones are below 2^-23 |a||b| each, i.e. the result is
// fp32-accurate to ~2 ulp of the sum of magnitudes, not bit-identical to an fp32 fmaf chain (tests: 1e-5 of the per-element sum of
// magnitudes, incl. inputs spanning e^+-6).  LN_CONV_EXACT_F32=1 keeps every shape on the fp32-input kernels.
// The fused backward additionally REQUIRES a symmetric neighbour list (nbr(m, e) = n  <=>  nbr(n, e^1) = m): true for a lattice
// convolved with itself whenever both lookups succeed; a lookup that fails one way only (the reference's 300-probe retrieval cap on a
// table loaded beyond ~0.97) would make its filter gradient differ from the two-launch backward — ln_conv_backward takes the fused
// form only for nbr_q == nbr_n (same list object), and tables that full take the replayed atomic build, whose probe sequence is
// the same in both directions.
#include "ln_common.h"
#include <stdlib.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));

// fused backward of a same-lattice small-filter convolution (k_conv_backward_fused): vertices per workgroup, shapes it covers
#define LN_BWD_MAX_SUBTILES 4
#define LN_BWD_CUS 256
static bool ln_bwd_fused_shape(int filter_extent, int val_dim, int nr_filters) {
    return filter_extent == 9 && val_dim == 32 && nr_filters == 32;
}
// 64-vertex sub-tiles per workgroup (1..4).  A workgroup takes a whole CU, so the launch runs in rounds of 256 workgroups and a
// round costs about T + 1: the T with the cheapest rounds(T) * (T + 1) wins, larger T on ties (fewer slabs) — one round at C3 (T = 3; a
// 257th workgroup would run after all the others: twice the time), 3 rounds of T = 3 at 129 k vertices.
static bool ln_conv_b3_enabled();
#ifndef LN_BWD_B3_MAX_T
#define LN_BWD_B3_MAX_T 3  // sub-tiles of the bf16x3 form: four fit the LDS since round 6 (unpadded staging) but need 148 registers of the 128 a
                           // 1024-thread workgroup may have (20 spilled); the fp32 form at T = 4 was the slowest choice at 129 k vertices (65 us
                           // against 55 for the bf16x3 form at T = 3: profiles/r6_kernel_stats_C4_one_in_flight.csv)
#endif
static int ln_bwd_subtiles(int m) {
    const int max_t = (ln_conv_b3_enabled() && !(ln_debug_mask() & 65536)) ? LN_BWD_B3_MAX_T : LN_BWD_MAX_SUBTILES;
    {
        static int forced = -1;  // experiment knob: LN_BWD_T=1..4 forces the sub-tile count
        if (forced < 0) {
            const char* e = getenv("LN_BWD_T");
            forced = e ? atoi(e) : 0;
        }
        if (forced >= 1 && forced <= LN_BWD_MAX_SUBTILES) return forced;
    }
    const int s = (m + 63) / 64;
    int best = 1, best_cost = 1 << 30;
    for (int t = 1; t <= max_t; ++t) {
        const int wgs = (s + t - 1) / t;
        const int cost = ((wgs + LN_BWD_CUS - 1) / LN_BWD_CUS) * (t + 1);  // (+ 1: the bank staging and the slab epilogue of a workgroup — at 129 k
                                                                           // vertices T = 2 / 3 run 4 / 3 rounds and measure 884 / 940 Mpoints/s)
        if (cost <= best_cost) {
            best = t;
            best_cost = cost;
        }
    }
    return best;
}
static int ln_bwd_workgroups(int m) { return (m + 64 * ln_bwd_subtiles(m) - 1) / (64 * ln_bwd_subtiles(m)); }

// Phase stamps of the small-filter convolution (tools/kernel_timeline.py --conv; -DLN_STAMPS builds only)
#ifdef LN_STAMPS
__device__ unsigned long long* g_ln_stamps_conv = nullptr;
extern "C" int ln_debug_set_stamps_conv(void* buffer) {
    unsigned long long* p = static_cast<unsigned long long*>(buffer);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ln_stamps_conv), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define LN_CSTAMP(slot)                                                                                                  \
    do {                                                                                                                 \
        if (g_ln_stamps_conv && threadIdx.x == 0) g_ln_stamps_conv[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define LN_CSTAMP(slot) do { } while (0)
#endif

template <int KQ>
__device__ __forceinline__ void ln_load_quarter(const float* __restrict__ src, float* a) {
    if constexpr (KQ % 4 == 0) {
#pragma unroll
        for (int k = 0; k < KQ; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + k);
            a[k] = v.x;
            a[k + 1] = v.y;
            a[k + 2] = v.z;
            a[k + 3] = v.w;
        }
    } else if constexpr (KQ % 2 == 0) {
#pragma unroll
        for (int k = 0; k < KQ; k += 2) {
            const float2 v = *reinterpret_cast<const float2*>(src + k);
            a[k] = v.x;
            a[k + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KQ; ++k) a[k] = src[k];
    }
}

// FLIP: read slot e^1 of an un-flipped neighbour list (the flipped traversal only swaps the np/nm slots
// of every axis, LatticeGPU.cuh:1622-1626).  WT: `filter` is the bank of the op being differentiated,
// [E*F, V]; the contraction uses its per-slot transpose (lattice_funcs.py:307-311) without materialising it.
// Compiled for 2 waves per SIMD: left to itself the compiler aims at 3-4 (it has LDS for that) and spills the prefetched
// neighbour row — 36 dwords of scratch per lane at V = 128.
#ifndef LN_CONV_WAVES_ATTR
#define LN_CONV_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
template <int V, int NT, bool FLIP, bool WT>
__global__ void __launch_bounds__(256) LN_CONV_WAVES_ATTR
    k_conv_mfma(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ filter, int m, int E,
                float* __restrict__ out, int f_total, int f_off, int e_per) {
    // gridDim.z > 1 (few vertices: the coarse levels of a U-Net leave most CUs without a tile): split over the filter slots —
    // workgroup z contracts slots [z * e_per, (z + 1) * e_per) only and writes its partial sums to slab z of `out`
    // ([gridDim.z, m, f_total]); ln_k_sum_partials adds the slabs.
    const int e_begin = blockIdx.z * e_per;
    const int e_end = min(E, e_begin + e_per);
    out += (size_t)blockIdx.z * m * f_total;
    // computes output columns [f_off, f_off + 16*NT) of an [.., f_total]-wide convolution.
    // Software pipeline over the filter slots: while the matrix cores work on slot e (operand A in registers, W_e in LDS),
    // the gather of slot e+1's neighbour rows and the float4 loads of W_{e+1} are already in flight; W_{e+1} goes from
    // registers to LDS between two barriers once every wave is done with W_e.
    constexpr int F = 16 * NT;
    constexpr int KQ = V / 4;
    constexpr int W4 = (V * F / 4 + 255) / 256;  // float4 of a filter slice per thread
    static_assert(V % 4 == 0, "V must be a multiple of 4");
    __shared__ __attribute__((aligned(16))) float s_b[V * F];  // W_e in fragment order [(kk*NT+nt)*64 + lane]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int m0 = blockIdx.x * 64 + wave * 16;
    const int my_row = m0 + i;
    f_off += blockIdx.y * F;  // gridDim.y consecutive column chunks of this width in one launch

    floatx4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    float a_cur[KQ], a_nxt[KQ];
    float4 w_nxt[W4];
    auto issue = [&](int e, float (&a)[KQ]) {  // loads of slot e: this lane's quarter of the neighbour row + its share of W_e
        const int es = (FLIP && e < E - 1) ? (e ^ 1) : e;
        const int nb = (my_row < m) ? nbr[(size_t)my_row * E + es] : -1;
        ln_load_quarter<KQ>(values + (size_t)(nb >= 0 ? nb : 0) * V + q * KQ, a);
        if (nb < 0) {
#pragma unroll
            for (int k = 0; k < KQ; ++k) a[k] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < W4; ++s) {
            const int x = (tid + s * 256) * 4;
            if (x < V * F) {
                // !WT: x = k*F + f (4 consecutive f);  WT: x = f*V + k (4 consecutive k)
                const int k = WT ? (x % V) : (x / F);
                const int f = WT ? (x / V) : (x - k * F);
                const size_t src = WT ? ((size_t)e * f_total + f_off + f) * V + k : ((size_t)e * V + k) * f_total + f_off + f;
                w_nxt[s] = *reinterpret_cast<const float4*>(filter + src);
            }
        }
    };
    auto stage = [&]() {  // W registers -> LDS in fragment order
#pragma unroll
        for (int s = 0; s < W4; ++s) {
            const int x = (tid + s * 256) * 4;
            if (x < V * F) {
                if constexpr (!WT) {
                    const int k = x / F, f = x - k * F;
                    const int qq = k / KQ, kk = k - qq * KQ;
                    *reinterpret_cast<float4*>(s_b + ((kk * NT) + (f >> 4)) * 64 + qq * 16 + (f & 15)) = w_nxt[s];
                } else {
                    const int f = x / V, k0 = x - f * V;
                    const float v4[4] = {w_nxt[s].x, w_nxt[s].y, w_nxt[s].z, w_nxt[s].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k0 + j;
                        const int qq = k / KQ, kk = k - qq * KQ;
                        s_b[((kk * NT) + (f >> 4)) * 64 + qq * 16 + (f & 15)] = v4[j];
                    }
                }
            }
        }
    };
    issue(e_begin, a_cur);
    stage();
    __syncthreads();
    for (int e = e_begin; e < e_end; ++e) {
        if (e + 1 < e_end) issue(e + 1, a_nxt);
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float b = s_b[(kk * NT + nt) * 64 + lane];
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[kk], b, acc[nt], 0, 0, 0);
            }
        }
        if (e + 1 < e_end) {
            __syncthreads();  // every wave is done with W_e
            stage();
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KQ; ++k) a_cur[k] = a_nxt[k];
        }
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) out[(size_t)row * f_total + f_off + nt * 16 + i] = acc[nt][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// The per-slot gather-GEMM on the bf16 matrix cores with exactly-split operands ("bf16x3"), for lattices of >= LN_CONV_B3_MIN_ROWS vertices.
// fp32-input MFMA runs at the fp32 VECTOR rate on gfx950 (32 cycles per 16x16x4): at ScanNet / SemanticKITTI shapes the kernel
// above sits at 50-80 % of that peak.  Here a = a1 + a2 + a3 EXACTLY, each part the top 16 bits of what is left (bf16 has fp32's
// exponent range, so no scaling; 3 x 8 significant bits = fp32's 24), the same for the filter, and a*b is accumulated in fp32 as
// the six products of total order <= 4: a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (what is dropped is < 2^-23 |ab|).  One
// v_mfma_f32_16x16x32_bf16 contracts 32 channels in ~17 cycles, so a slot costs 6 V/32 NT of those instead of V/4 NT of 32
// cycles: 2.5x less matrix time, for ~8 bit operations per gathered value.  The filter bank is split once per call
// (k_conv_split_bank) into fragment order, so staging a slot is a straight 16-byte copy.
// Lane (i, q) holds its quarter of the gathered row, V/4 contiguous channels: MFMA step s takes channels [8s, 8s + 8) of the
// quarter as the A fragment of k-group q, i.e. the contraction order inside a slot is permuted; the bank uses the same order.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = hi + mid + lo exactly, each the top 16 bits of the remainder; returned as raw bf16 bit patterns in the HIGH half-words
// (`lo` comes back UNMASKED — its low half-word is whatever is left below the third part: every user takes the high half-word only, by a
// 16-bit shift or through ln_pack_hi)
__device__ __forceinline__ void ln_split3_bits(float x, unsigned int& hi, unsigned int& mid, unsigned int& lo) {
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mid);
    lo = __float_as_uint(r2);
}
// two bf16 per dword: the HIGH half-word of `a` in the low half, the high half-word of `b` in the high half — one v_perm_b32 (round 6;
// before: shift + or on masked words, 3 more vector-ALU instructions per pair of split values)
__device__ __forceinline__ unsigned int ln_pack_hi(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// One block of the slab sum (k_reduce_slabs4 below): 64 outputs as 16 float4 columns x 16 slab groups (every thread has its
// <= ceil(nslabs / 16) float4 loads in flight at once), combined through LDS in a fixed order.  total % 64 == 0.
__device__ __forceinline__ void ln_reduce_slabs4_block(int vb, const float* __restrict__ partial, int nslabs, int total, float* __restrict__ out) {
    __shared__ float4 s_part[16][16];
    const int c = threadIdx.x & 15;
    const int g = threadIdx.x >> 4;
    const size_t col = (size_t)vb * 16 + c;  // float4 column
    const size_t stride4 = (size_t)total / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = g; s0 < nslabs; s0 += 16 * 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int sl = s0 + 16 * k;
            v[k] = sl < nslabs ? reinterpret_cast<const float4*>(partial)[(size_t)sl * stride4 + col] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w;
        }
    }
    s_part[g][c] = acc;
    __syncthreads();
    if (g == 0) {
        float4 r = s_part[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float4 t = s_part[k][c];
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        reinterpret_cast<float4*>(out)[col] = r;
    }
}

// Horizontal fusion in the backward pass of a convolution: the slab sum of the FILTER gradient (slabs written by the previous
// launch) rides in the launch that splits the bank of the VALUE-gradient convolution — two short launches that do not depend on each
// other become one (18 slab sums of ~5 us per step of the SemanticKITTI network).  The split kernels take the extra workgroups
// [split_x, gridDim.x) of every (y, z) plane: virtual block ((z * gridDim.y + y) * (gridDim.x - split_x) + x - split_x).
struct LnSlabSum {
    const float* partial;
    int nslabs, total;
    float* out;
};
#define LN_SLAB_SUM_IN_SPLIT(job, split_x)                                                                                       \
    if ((int)blockIdx.x >= (split_x)) {                                                                                          \
        const int vb_ = ((int)(blockIdx.z * gridDim.y + blockIdx.y)) * ((int)gridDim.x - (split_x)) + ((int)blockIdx.x - (split_x)); \
        if (vb_ < (job).total / 64) ln_reduce_slabs4_block(vb_, (job).partial, (job).nslabs, (job).total, (job).out);           \
        return;                                                                                                                  \
    }

// Filter slice of (slot e, column chunk y), split and in fragment order: [e][y][((s * NT + nt) * 3 + part) * 64 + lane][8 bf16]
template <int V, int NT, bool WT>
__global__ void __launch_bounds__(256)
    k_conv_split_bank(const float* __restrict__ filter, int f_total, int f_off, unsigned short* __restrict__ bank, int split_x, LnSlabSum job) {
    LN_SLAB_SUM_IN_SPLIT(job, split_x)
    constexpr int F = 16 * NT;
    constexpr int KQ = V / 4;
    constexpr int S = KQ / 8;
    constexpr int SLICE = S * NT * 3 * 64 * 8;  // bf16 elements per (slot, chunk)
    const int e = blockIdx.y;
    const int y = blockIdx.z;
    unsigned short* dst = bank + ((size_t)e * gridDim.z + y) * SLICE;
    const int fo = f_off + y * F;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= V * F) return;
    // !WT: x = k*F + f;  WT: x = f*V + k  (coalesced reads either way)
    const int k = WT ? (x % V) : (x / F);
    const int f = WT ? (x / V) : (x - k * F);
    const size_t src = WT ? ((size_t)e * f_total + fo + f) * V + k : ((size_t)e * V + k) * f_total + fo + f;
    unsigned int h, md, l;
    ln_split3_bits(filter[src], h, md, l);
    const int qq = k / KQ, kk = k - qq * KQ;
    const int st = kk >> 3, j = kk & 7;
    const int base = ((st * NT + (f >> 4)) * 3 * 64 + qq * 16 + (f & 15)) * 8 + j;
    dst[base] = (unsigned short)(h >> 16);
    dst[base + 64 * 8] = (unsigned short)(md >> 16);
    dst[base + 2 * 64 * 8] = (unsigned short)(l >> 16);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef LN_CONV_B3_PIPE
#define LN_CONV_B3_PIPE(V) ((V) <= 128)  // fragments read one chain ahead: +12 registers (the register estimate at the use decides: at 128 channels only with T = 3)
#endif
#ifndef LN_CONV_B3_DEEP
#define LN_CONV_B3_DEEP(V) ((V) <= 96)
#endif
#ifndef LN_CONV_LDS_E
#define LN_CONV_LDS_E 16  // filter extents up to 2 (d + 1) + 1 with d <= 6 keep their neighbour ids in LDS
#endif
#ifndef LN_CONV_PROBE
#define LN_CONV_PROBE 0  // timing ablations (wrong results): 1 no matrix products, 2 no operand split, 4 no bank staging, 8 no gather
#endif
// Waves per SIMD by gathered width (registers: 2 x V/4 row quarters + the staged bank slice + 4 NT accumulators; LDS: 36-48 KB bank
// slice + 4 KB ids per workgroup): 3 up to 128 channels (<= 168 registers, 3 x 52 KB of LDS), 2 above (192 / 256 channels need
// 174 / 220 registers: at 3 they spill, 1.77 ms instead of 0.63 at 256 x 256).  Measured at 46 k rows, 2 -> 3 waves: 64 x 64 37.7 -> 32.1 us,
// 96 x 96 2 x 48.4 -> 2 x 40.8, 128 x 128 129 -> 122, 32 -> 64 20.7 -> 16.1.
#define LN_CONV_B3_WAVES(V) ((V) <= 128 ? 3 : 2)
// T sub-tiles of 64 rows per workgroup (256 T threads) share ONE staging of the slot's bank slice: T = 3 (one workgroup per CU, the
// shape of k_conv_forward_b3) re-reads the bank from L2 and writes it to LDS a third as often as T = 1 (three workgroups per CU) —
// at 46 k rows x 64 channels the bank traffic of T = 1 (727 workgroups x 9 slices of 24 KB = 157 MB) exceeds the gathered rows (107 MB).
#ifndef LN_MFMA_B3_LINE
#define LN_MFMA_B3_LINE 1  // 0: fragment-shaped gathers at 64 channels too (A/B)
#endif
template <int V, int NT, bool FLIP, int T>
__global__ void __launch_bounds__(256 * T) __attribute__((amdgpu_waves_per_eu(LN_CONV_B3_WAVES(V), LN_CONV_B3_WAVES(V))))
    k_conv_mfma_b3(const int* __restrict__ nbr, const float* __restrict__ values, const u32x4* __restrict__ bank, int m, int E,
                   float* __restrict__ out, int f_total, int f_off, int e_per) {
    const int e_begin = blockIdx.z * e_per;  // slot split, as in k_conv_mfma
    const int e_end = min(E, e_begin + e_per);
    out += (size_t)blockIdx.z * m * f_total;
    constexpr int F = 16 * NT;
    constexpr int KQ = V / 4;
    constexpr int S = KQ / 8;                 // MFMA steps per slot
    constexpr int BANK16 = S * NT * 3 * 64;   // 16-byte fragments of one (slot, chunk) slice
    constexpr int THREADS = 256 * T;
    constexpr int W16 = (BANK16 + THREADS - 1) / THREADS;
    static_assert(V % 32 == 0, "bf16x3 path: V must be a multiple of 32");
    static_assert(T == 1 || LN_CONV_B3_WAVES(V) == T, "T > 1: one workgroup per CU, its waves are the SIMDs' whole occupancy");
    __shared__ u32x4 s_b[BANK16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int m0 = blockIdx.x * (64 * T) + wave * 16;
    const int my_row = m0 + i;
    f_off += blockIdx.y * F;

    floatx4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Line-shaped gathers at 64 channels (round 6; k_conv_forward_b3 and the fused backward have them at 32): a load shaped like the MFMA
    // fragment — lane (i, q) reads the 64-byte quarter of row i — touches 16 rows, 32 half-used lines, per wave-instruction; here lane l
    // loads piece l & 15 (16 bytes) of rows (l >> 4) + 4 j: four whole 256-byte rows per instruction.  The wave's 16 x 64 floats change
    // shape through a private 4 KB LDS region when they become the current slot's operand: piece p of row r sits at column
    // p ^ ((r & 3) | ((r >> 2) & 1) << 3) of the row's 256 bytes (= the 64 banks) — conflict-free for the four lane groups of
    // ds_read_b128 under the (i, q) mapping, whichever of its four pieces a lane reads (exhaustive search over the linear maps), and for
    // the stores (8 adjacent lanes = half a row).  Taken where it measures faster (A/B of two builds on one box, tools/probes/r6_ab_conv64.sh):
    // 64 -> 32 at 46.5 k rows 20.8 -> 18.2 us, 64 -> 64 at 11.4 k rows 13.8 -> 12.9; with three sub-tiles and 64 or 128 columns per
    // chunk the kernel is bound by its bank slices and LDS reads, not by the gather: 23.5 -> 24.0 and 39.7 -> 40.3 us — left as they were.
    constexpr bool LINE = LN_MFMA_B3_LINE && V == 64 && (NT <= 2 || (T == 1 && NT <= 4));
    __shared__ floatx4 s_x[LINE ? 4 * T : 1][LINE ? 16 * 16 : 1];
    const int lr = lane >> 4, pc = lane & 15;
    auto fsw = [](int r) -> int { return (r & 3) | (((r >> 2) & 1) << 3); };
    floatx4 l_nxt[4];
    int nbl_nxt[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int j = 0; j < 4; ++j) l_nxt[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a_cur[KQ], a_nxt[KQ];
    u32x4 w_nxt[W16];
#pragma unroll
    for (int k = 0; k < KQ; ++k) a_nxt[k] = 0.f;
    // Loads ahead of the products of slot e: the gathered quarter row and the bank slice of slot e + 1 (global), the
    // neighbour id of slot e + 2 (LDS).  The ids of the workgroup's 64 rows are copied into LDS once, coalesced (the launch
    // requires E <= LN_CONV_LDS_E): a per-lane global id load inside the loop is either waited for on the spot or, hoisted a
    // slot ahead by hand, sunk back to its use by the compiler.  Only the first slot's id comes straight from global memory,
    // beside that copy.
    __shared__ int s_nbr[64 * T * LN_CONV_LDS_E];
    {
        const size_t g0 = (size_t)blockIdx.x * (64 * T) * E, g_end = (size_t)m * E;
        for (int x = tid; x < 64 * T * E; x += THREADS) s_nbr[x] = (g0 + x < g_end) ? nbr[g0 + x] : -1;
    }
    auto slot_of = [&](int e) -> int { return (FLIP && e < E - 1) ? (e ^ 1) : e; };
    const int* my_ids = s_nbr + ((tid >> 6) * 16 + i) * E;
    auto load_nb = [&](int e) -> int { return e < e_end ? my_ids[slot_of(e)] : -1; };
    // DEEP (V <= 96): an absent neighbour's row is zeroed when it becomes a_cur, so the gather is not waited for before the
    // products of the current slot.  Wider rows: zeroed at issue — the wave waits for its gather first.  A/B with everything else
    // in place, 46 k rows, deferred vs at issue: 32 -> 64 16.1 vs 17.8 us, 64 x 64 30.6 vs 32.1, 96 x 96 2 x 38.7 vs 2 x 41.1,
    // 128 x 128 116-117 vs 115-118 (equal), 128 -> 64 62-63 vs 61 (the 32 extra selects per slot show).
    constexpr bool DEEP = LN_CONV_B3_DEEP(V);
    auto load_nbl = [&](int e, int (&dst)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = e < e_end ? s_nbr[((tid >> 6) * 16 + 4 * j + lr) * E + slot_of(e)] : -1;
    };
    auto reshape = [&]() {  // the rows loaded for the next slot become a_cur (absent neighbours: zero rows, written by the lane that loaded them)
        floatx4* xs = s_x[LINE ? (tid >> 6) : 0];
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[(4 * j + lr) * 16 + (pc ^ fsw(4 * j + lr))] = nbl_nxt[j] >= 0 ? l_nxt[j] : floatx4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const floatx4 x = xs[i * 16 + ((4 * q + jj) ^ fsw(i))];
            a_cur[4 * jj] = x[0], a_cur[4 * jj + 1] = x[1], a_cur[4 * jj + 2] = x[2], a_cur[4 * jj + 3] = x[3];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto issue = [&](int e, int nb, float (&a)[KQ]) {
        if constexpr (LINE) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                l_nxt[j] = *reinterpret_cast<const floatx4*>(values + (size_t)(nbl_nxt[j] >= 0 ? nbl_nxt[j] : 0) * V + pc * 4);
        } else {
            ln_load_quarter<KQ>(values + (size_t)(nb >= 0 ? nb : 0) * V + q * KQ, a);
            if (!DEEP && nb < 0) {
#pragma unroll
                for (int k = 0; k < KQ; ++k) a[k] = 0.f;
            }
        }
        const u32x4* src = bank + ((size_t)e * gridDim.y + blockIdx.y) * BANK16;
#pragma unroll
        for (int s = 0; s < W16; ++s) {
            const int x = tid + s * THREADS;
            if (BANK16 % THREADS == 0 || x < BANK16) w_nxt[s] = src[x];
        }
    };
    auto stage = [&]() {  // straight copy: the bank is already in fragment order
#pragma unroll
        for (int s = 0; s < W16; ++s) {
            const int x = tid + s * THREADS;
            if (BANK16 % THREADS == 0 || x < BANK16) s_b[x] = w_nxt[s];
        }
    };
    int nb_nxt = -1, nb_nn = -1;
    if constexpr (LINE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) nbl_nxt[j] = (m0 + 4 * j + lr < m) ? nbr[(size_t)(m0 + 4 * j + lr) * E + slot_of(e_begin)] : -1;
#pragma unroll
        for (int k = 0; k < KQ; ++k) a_cur[k] = 0.f;
    } else {
        nb_nxt = my_row < m ? nbr[(size_t)my_row * E + slot_of(e_begin)] : -1;
    }
    issue(e_begin, nb_nxt, a_cur);
    if (!LINE && nb_nxt < 0) {
#pragma unroll
        for (int k = 0; k < KQ; ++k) a_cur[k] = 0.f;
    }
    stage();
    __syncthreads();
    if constexpr (LINE) {
        reshape();
        load_nbl(e_begin + 1, nbl_nxt);
    } else {
        nb_nxt = load_nb(e_begin + 1);
    }
    for (int e = e_begin; e < e_end; ++e) {
        if (e + 1 < e_end) {
            if constexpr (!LINE) nb_nn = load_nb(e + 2);
#if LN_CONV_PROBE & 8
#pragma unroll
            for (int k = 0; k < KQ; ++k) a_nxt[k] = a_cur[k] * 1.5f;
#elif LN_CONV_PROBE & 4
            ln_load_quarter<KQ>(values + (size_t)(nb_nxt >= 0 ? nb_nxt : 0) * V + q * KQ, a_nxt);
#else
            issue(e + 1, nb_nxt, a_nxt);
#endif
        }
        // (at 128 channels with one sub-tile per workgroup the staged bank slice takes 24-48 registers per thread: the double buffer spills there)
        constexpr bool PIPE = LN_CONV_B3_PIPE(V) && (V <= 96 || T == 3 || NT == 1);
        u32x4 fb[2][3];
        if constexpr (PIPE) {
#pragma unroll
            for (int x = 0; x < 3; ++x) fb[0][x] = s_b[x * 64 + lane];
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 48, 0);  // the first step's split
        }
#pragma unroll
        for (int st = 0; st < S; ++st) {
            // three bf16x8 fragments of this lane's 8 channels: two bf16 per dword, the LOWER channel in the low half-word
            u32x4 p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned int h0, m0_, l0, h1, m1_, l1;
                ln_split3_bits(a_cur[st * 8 + 2 * j], h0, m0_, l0);
                ln_split3_bits(a_cur[st * 8 + 2 * j + 1], h1, m1_, l1);
                p1[j] = ln_pack_hi(h0, h1);
                p2[j] = ln_pack_hi(m0_, m1_);
                p3[j] = ln_pack_hi(l0, l1);
            }
#if LN_CONV_PROBE & 2
            p1 = u32x4{__float_as_uint(a_cur[st * 8]), __float_as_uint(a_cur[st * 8 + 1]), __float_as_uint(a_cur[st * 8 + 2]), __float_as_uint(a_cur[st * 8 + 3])};
            p2 = u32x4{__float_as_uint(a_cur[st * 8 + 4]), __float_as_uint(a_cur[st * 8 + 5]), __float_as_uint(a_cur[st * 8 + 6]), __float_as_uint(a_cur[st * 8 + 7])};
            p3 = p1;
#endif
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, p1), a2 = __builtin_bit_cast(bf16x8, p2), a3 = __builtin_bit_cast(bf16x8, p3);
#if LN_CONV_PROBE & 1
            acc[0][0] += __uint_as_float((p1[0] ^ p2[1] ^ p3[2]) + (p1[1] ^ p2[2] ^ p3[3]) + (p1[2] ^ p2[3] ^ p3[0]) + (p1[3] ^ p2[0] ^ p3[1]));
            continue;
#endif
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                // PIPE: the three bank fragments of the NEXT chain are read from LDS ahead of this chain's six products, which
                // then issue back to back (an instruction between two products on one accumulator costs ~43 cycles,
                // MI355X_MICROARCH.md); the vector-ALU work of the next step's split goes between chains.  The order is pinned
                // with scheduling groups (DS read 0x100, MFMA 0x008, VALU 0x002).
                const int idx = st * NT + nt;
                bf16x8 b1, b2, b3;
                if constexpr (PIPE) {
                    if (idx + 1 < S * NT) {
#pragma unroll
                        for (int x = 0; x < 3; ++x) fb[(idx + 1) & 1][x] = s_b[((idx + 1) * 3 + x) * 64 + lane];
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (48 + NT - 1) / NT + 2, 0);
                    b1 = __builtin_bit_cast(bf16x8, fb[idx & 1][0]), b2 = __builtin_bit_cast(bf16x8, fb[idx & 1][1]), b3 = __builtin_bit_cast(bf16x8, fb[idx & 1][2]);
                } else {
                    const u32x4* pb = s_b + (idx * 3) * 64 + lane;
                    b1 = __builtin_bit_cast(bf16x8, pb[0]), b2 = __builtin_bit_cast(bf16x8, pb[64]), b3 = __builtin_bit_cast(bf16x8, pb[128]);
                }
                // small terms first, the dominant product last
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[nt], 0, 0, 0);
            }
        }
        if (e + 1 < e_end) {
#if !(LN_CONV_PROBE & 4)
            __syncthreads();  // every wave is done with W_e
            stage();
            __syncthreads();
#endif
        }
        // unconditional (a_nxt is initialised): no register of a_cur is ever undefined on a path through the loop — with
        // undefined lanes in the loop-carried registers hipcc 7.2 has mis-assigned the operands of the split's pack instructions
        if constexpr (LINE) {
            reshape();
            load_nbl(e + 2, nbl_nxt);  // (ids from LDS: read here, used at the top of the next trip — four more registers a slot ahead spill at T = 3)
        } else {
#pragma unroll
            for (int k = 0; k < KQ; ++k) a_cur[k] = (!DEEP || nb_nxt >= 0) ? a_nxt[k] : 0.f;
            nb_nxt = nb_nn;
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) out[(size_t)row * f_total + f_off + nt * 16 + i] = acc[nt][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Per-slot convolution, wide form (k_conv_rows32_b3): the same bf16x3 contraction on v_mfma_f32_32x32x16_bf16 with BOTH operands
// arriving in LDS by LDS-DMA (global_load_lds_dwordx4), no staging registers:
//   * A (gathered rows).  Measured with tools/probes/gather_layout_probe.cpp at 46 538 rows x 9 slots x 512 B: loads shaped like the
//     MFMA fragment (lane (i, q) reads its quarter of row i: 64 lines touched per wave-instruction) gather at 3.5 TB/s = 61 us per pass
//     whatever the cache hit rate; 8 adjacent lanes per 128-byte line gather at 7.0 TB/s = 31 us.  k_conv_mfma_b3 gathers fragment-shaped
//     AND once per 64-column chunk (two passes at 128 filters): its 100 us at 128 x 128 is that gather.  Here a wave fetches the
//     32-channel chunk of its 32 rows as 4 wave-instructions of 8 rows x 128 B into a private 4 KB LDS region (no workgroup barrier
//     for A: the wave's own vmcnt orders it), ONCE for all output columns of the launch, and reads its fragments with ds_read_b128.
//     The region is XOR-swizzled through the SOURCE addresses (the DMA writes lane-linear): 16-byte column c of row r sits at
//     position c ^ ((r >> 1) & 7), conflict-free for the four 16-lane groups of ds_read_b128.
//   * B (the pre-split bank, k_conv_split_bank32) comes in chunks of 32 channels x (32 NT) columns x 3 parts = 6 NT KB, double
//     buffered: ONE barrier per chunk, and every 1 KB fragment read from LDS feeds 32 rows (the 16-row form: 16).
// Workgroup = 4 waves x 32 rows; 64 KB of LDS at 128 columns -> two workgroups per CU, whose barriers overlap.
// Absent neighbours and rows past m read a zero row, so the DMA needs no predication.
// ------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));
// One LDS-DMA wave-instruction (64 lanes x 16 bytes -> LDS bytes [lds_dst, lds_dst + 1024)), issued from inline asm so that hipcc
// does not know an LDS write is pending: with the builtin it puts `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY LDS
// object, i.e. waits for the chunk just requested.  The kernel orders the data itself (counted vmcnt + barrier).  hipcc's own vmcnt
// accounting stays safe: it only ever assumes FEWER loads outstanding than there are, so its waits are stricter, never laxer.
__device__ __forceinline__ void ln_glds16(const void* gsrc, unsigned int lds_dst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned int ln_lds_addr(const void* p) {
    return (unsigned int)(size_t)(__attribute__((address_space(3))) const char*)(const char*)p;
}
__device__ __attribute__((aligned(256))) float g_ln_zero_row[1024];  // zero-initialised: the source of absent neighbours' rows

// Bank for k_conv_rows32_b3: [e][y][kc][(s * NT + nt) * 3 + part][lane = h * 32 + n][8 bf16] with
// value = part of W[e][kc * 32 + s * 16 + h * 8 + j][f_off + (y * NT + nt) * 32 + n]
template <int V, int NT, bool WT>
__global__ void __launch_bounds__(256)
    k_conv_split_bank32(const float* __restrict__ filter, int f_total, int f_off, unsigned short* __restrict__ bank, int split_x, LnSlabSum job) {
    LN_SLAB_SUM_IN_SPLIT(job, split_x)
    constexpr int F = 32 * NT;
    constexpr int CHUNK = 2 * NT * 3 * 64 * 8;  // bf16 elements per (slot, column chunk, channel chunk)
    const int e = blockIdx.y;
    const int y = blockIdx.z;
    const int fo = f_off + y * F;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= V * F) return;
    const int k = WT ? (x % V) : (x / F);
    const int f = WT ? (x / V) : (x - k * F);
    const size_t src = WT ? ((size_t)e * f_total + fo + f) * V + k : ((size_t)e * V + k) * f_total + fo + f;
    unsigned int h, md, l;
    ln_split3_bits(filter[src], h, md, l);
    const int kc = k >> 5, s = (k >> 4) & 1, hh = (k >> 3) & 1, j = k & 7;
    unsigned short* dst = bank + (((size_t)e * gridDim.z + y) * (V / 32) + kc) * CHUNK;
    const int base = (((s * NT + (f >> 5)) * 3) * 64 + hh * 32 + (f & 31)) * 8 + j;
    dst[base] = (unsigned short)(h >> 16);
    dst[base + 64 * 8] = (unsigned short)(md >> 16);
    dst[base + 2 * 64 * 8] = (unsigned short)(l >> 16);
}

// -DLN_STAMPS builds: per-wave phase accounting of k_conv_rows32_b3 (tools/probes/r32_phase_probe.py) with s_memtime (shader
// cycles; its lgkmcnt wait falls on phase boundaries where no LDS read is outstanding or all are about to be consumed)
#ifdef LN_STAMPS
#define LN_R32_CYC() ((unsigned)__builtin_amdgcn_s_memtime())
#define LN_R32_PHASE(k)                                           \
    do {                                                          \
        __builtin_amdgcn_sched_barrier(0);                        \
        const unsigned now_ = LN_R32_CYC();                       \
        ph[k] += (now_ - t_prev) & 0xFFFFFu;                      \
        t_prev = now_;                                            \
        __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)
#else
// (the scheduling barrier stays in the product build: left free, hipcc moves the split of the next rows and the fragment reads across
// the phase boundaries and the kernel is 5-8 % slower than the instrumented build)
#ifndef LN_CONV_R32_PIN
#define LN_CONV_R32_PIN 1
#endif
#if LN_CONV_R32_PIN
#define LN_R32_PHASE(k) __builtin_amdgcn_sched_barrier(0)
#else
#define LN_R32_PHASE(k) do { } while (0)
#endif
#endif
#ifndef LN_CONV_R32_SPREAD
#define LN_CONV_R32_SPREAD 0  // 1: the requests of an iteration go out in two halves
#endif
#ifndef LN_CONV_R32_PRIO
#define LN_CONV_R32_PRIO 0    // 1..3: s_setprio around the products of a K-step
#endif
#ifndef LN_CONV_R32_SKEW
#define LN_CONV_R32_SKEW 0    // 1: the second K-step's products are held back behind the next barrier
#endif
#ifndef LN_CONV_R32_PROBE
#define LN_CONV_R32_PROBE 0  // timing ablations (wrong results): 1 no matrix products, 2 no operand split, 4 no A gather (one row), 8 no B staging
#endif
// Workgroup = RT row tiles of 32 rows x CH column groups of 32 NTW columns: RT * CH waves, wave (rt, ch) owns the [32 x 32 NTW]
// block of the output.  The CH waves of a row tile share its gathered A chunk (double buffered: the partner may still be reading
// while the next one is requested) and request a CH-th of its four pieces each.  At 128 columns: RT = 6, CH = 2, NTW = 2 — twelve
// waves (three per SIMD, covering each other's LDS reads and operand splits), 192 rows, 108 KB of LDS, one workgroup per CU,
// 243 workgroups for the 46.5 k rows of the SemanticKITTI lattice (four waves x 32 rows x 128 columns with two workgroups per CU
// measured 89 us there: 108 CUs hold two of the 364 workgroups and a lone wave per SIMD leaves the matrix pipe idle during its
// 1100-cycle prelude of LDS reads and splits — tools/probes/r32_phase_probe.py).
template <int V, int NTW, int CH, int RT, bool FLIP>
__global__ void __launch_bounds__(64 * RT * CH) __attribute__((amdgpu_waves_per_eu((RT * CH + 3) / 4, (RT * CH) % 6 == 0 ? 3 : 2)))
    k_conv_rows32_b3(const int* __restrict__ nbr, const float* __restrict__ values, const u32x4* __restrict__ bank, int m, int E,
                     float* __restrict__ out, int f_total, int f_off, int e_per) {
    const int e_begin = blockIdx.z * e_per;  // slot split, as in k_conv_mfma
    const int e_end = min(E, e_begin + e_per);
    out += (size_t)blockIdx.z * m * f_total;
    constexpr int NT = NTW * CH;             // 32-column tiles of the workgroup
    constexpr int W = RT * CH;               // waves
    constexpr int R = 32 * RT;               // rows
    constexpr int NKC = V / 32;              // channel chunks per slot
    constexpr int BCH16 = 2 * NT * 3 * 64;   // 16-byte fragments of one bank chunk (1 KB = 64 of them per wave-instruction)
    constexpr int BPIECES = BCH16 / 64;      // 6 NT wave-instructions per chunk
    constexpr int BPW = (BPIECES + W - 1) / W;  // ... per wave
    constexpr int APW = 4 / CH;              // A pieces per wave (of the four of its row tile)
    constexpr int G = 2 * NTW;               // MFMA groups (K-step, column tile) per chunk and wave: 6 products each
    static_assert(V % 32 == 0 && (CH == 1 || CH == 2 || CH == 4) && W <= 16, "rows32 shape");
    __shared__ u32x4 s_b[2][BCH16];
    __shared__ u32x4 s_a[CH > 1 ? 2 : 1][RT][256];  // per row tile: 32 rows x 8 positions of 16 bytes
    __shared__ int s_nbr[R * LN_CONV_LDS_E];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave / CH, ch = wave % CH;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int m0 = blockIdx.x * R + rt * 32;
    const int f_wg = f_off + blockIdx.y * (32 * NT);  // first column of the workgroup
    f_off = f_wg + ch * (32 * NTW);                    // ... of this wave
    {
        const size_t g0 = (size_t)blockIdx.x * R * E, g_end = (size_t)m * E;
        for (int x = tid; x < R * E; x += 64 * W) s_nbr[x] = (g0 + x < g_end) ? nbr[g0 + x] : -1;
    }
    auto slot_of = [&](int e) -> int { return (FLIP && e < E - 1) ? (e ^ 1) : e; };
    // A piece j of a row tile covers its rows 8j .. 8j+7: lane -> (row 8j + (lane >> 3), position lane & 7); this wave requests
    // pieces ch * APW .. ch * APW + APW - 1
    const int a_row = lane >> 3;
    const int* my_ids = s_nbr + (rt * 32 + ch * APW * 8 + a_row) * E;
    const unsigned int lds_b = __builtin_amdgcn_readfirstlane(ln_lds_addr(&s_b[0][0]));
    const unsigned int lds_a = __builtin_amdgcn_readfirstlane(ln_lds_addr(&s_a[0][0][0]) + (unsigned)(rt * 4096 + ch * APW * 1024));
    const u32x4* bank_y = bank + (size_t)blockIdx.y * NKC * BCH16 + lane;   // + e * gridDim.y * NKC * BCH16 + kc * BCH16
    const size_t bank_e = (size_t)gridDim.y * NKC * BCH16;
    // float offset of this lane's 16 bytes inside a gathered row's 32-channel chunk (the XOR swizzle, on the source side)
    int a_off[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) a_off[j] = (((lane & 7) ^ (((8 * (ch * APW + j) + a_row) >> 1) & 7)) * 4);
    // piece k of the NEXT chunk (e_n, kc_n): k < BPW this wave's share of the bank chunk, then its A pieces
    int ids[APW];
    auto dma_piece = [&](int k, int e_n, int kc_n, int buf_n) {
        if (k < BPW) {
            const int p = wave + W * k;
            if (BPIECES % W == 0 || p < BPIECES)
                ln_glds16(bank_y + (size_t)e_n * bank_e + (size_t)kc_n * BCH16 + p * 64, lds_b + (unsigned)(buf_n * BCH16 + p * 64) * 16u);
        } else {
            const int j = k - BPW;
            const int nb = ids[j];
            const float* src = (nb >= 0 ? values + (size_t)nb * V : g_ln_zero_row) + kc_n * 32 + a_off[j];
#if LN_CONV_R32_PROBE & 4
            src = values + (size_t)(m0 + 8 * (ch * APW + j) + a_row < m ? m0 + 8 * (ch * APW + j) + a_row : 0) * V + a_off[j];
#endif
            ln_glds16(src, lds_a + (unsigned)((CH > 1 ? buf_n : 0) * (RT * 4096) + j * 1024));
        }
    };
    constexpr int NPIECES = BPW + APW;
    floatx16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int total = (e_end - e_begin) * NKC;
    // chunk t of the walk = (slot e_begin + t / NKC, channel chunk t % NKC); past the end the last chunk is requested again
    // (nobody reads it) instead of branching around every request
    auto chunk_e = [&](int t) -> int { return e_begin + min(t, total - 1) / NKC; };
    auto chunk_kc = [&](int t) -> int { return min(t, total - 1) % NKC; };
    auto load_ids_of_slot = [&](int e_) {
        const int sl = slot_of(e_);
#pragma unroll
        for (int j = 0; j < APW; ++j) ids[j] = my_ids[j * 8 * E + sl];
    };
    auto load_ids = [&](int t) { load_ids_of_slot(chunk_e(t)); };
    const int a_sw = (i >> 1) & 7;
    u32x4 araw[2][2];
    auto read_a = [&](int buf) {
        const u32x4* my_a = &s_a[CH > 1 ? buf : 0][rt][0] + i * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) araw[s][j2] = my_a[(s * 4 + h * 2 + j2) ^ a_sw];
    };
    bf16x8 ap[2][3];
    auto split_step = [&](int s, bf16x8 (&dst)[2][3]) {
        u32x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // x = hi + mid + lo with each part the top 16 bits of the remainder (ln_split3_bits), written out so that only the two
            // masks the subtractions need are computed: the byte permutes below take the top half-words of x, r1 and r2 directly.
            // Two bf16 per dword, the lower channel in the low half-word.
            const float x0 = __uint_as_float(araw[s][j >> 1][(2 * j) & 3]), x1 = __uint_as_float(araw[s][j >> 1][(2 * j + 1) & 3]);
            const float r10 = x0 - __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u), r11 = x1 - __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
            const float r20 = r10 - __uint_as_float(__float_as_uint(r10) & 0xFFFF0000u), r21 = r11 - __uint_as_float(__float_as_uint(r11) & 0xFFFF0000u);
            p1[j] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
            p2[j] = __builtin_amdgcn_perm(__float_as_uint(r11), __float_as_uint(r10), 0x07060302u);
            p3[j] = __builtin_amdgcn_perm(__float_as_uint(r21), __float_as_uint(r20), 0x07060302u);
        }
#if LN_CONV_R32_PROBE & 2
        p1 = araw[s][0];
        p2 = araw[s][1];
        p3 = p1;
#endif
        dst[s][0] = __builtin_bit_cast(bf16x8, p1), dst[s][1] = __builtin_bit_cast(bf16x8, p2), dst[s][2] = __builtin_bit_cast(bf16x8, p3);
    };

    // Software pipeline, per wave.  While the products of chunk t issue, the rows of chunk t + 1 (landed before the barrier of
    // iteration t) are read from LDS and split; and the products of a chunk's SECOND K-step are held back until the barrier of the
    // next iteration has been passed, so that the matrix pipe has work while the first fragments of the new chunk are on their way
    // from LDS (every wave of the workgroup leaves the barrier in the same state).  In flight across an iteration: the bank chunk
    // t + 1 and the rows of chunk t + 2.  Two register sets of split operands alternate (iterations come in pairs: no copies).
    // A tile none of whose rows has any neighbour (static-rows mode launches over a row BOUND: the tiles past the lattice's real rows
    // hold -1 everywhere) writes zeros and leaves: 46.5 k rows under a 6 % bound are 257 workgroups of 192 rows — one more than the
    // chip has CUs, i.e. a second round for nothing (the whole-network graph ran 0.17 ms slower with the 192-row split-K tiles).
    {
        bool any = false;
        for (int x = tid; x < R * E; x += 64 * W) any |= s_nbr[x] >= 0;
        if (!__syncthreads_or(any)) {  // (also the barrier behind the id copy)
            for (int x = tid; x < R * (32 * NT / 4); x += 64 * W) {
                const int row = blockIdx.x * R + x / (32 * NT / 4), c4 = x % (32 * NT / 4);
                if (row < m) *reinterpret_cast<float4*>(out + (size_t)row * f_total + f_wg + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
    }
    load_ids(0);
#pragma unroll
    for (int k = 0; k < NPIECES; ++k) dma_piece(k, chunk_e(0), chunk_kc(0), 0);
    if constexpr (CH > 1) {
        load_ids(1);
#pragma unroll
        for (int k = BPW; k < NPIECES; ++k) dma_piece(k, chunk_e(1), chunk_kc(1), 1);
        load_ids(2);
    } else {
        load_ids(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (CH == 1) {
#pragma unroll
        for (int k = BPW; k < NPIECES; ++k) dma_piece(k, chunk_e(1), chunk_kc(1), 0);
        load_ids(2);
    }
    bf16x8 ap_b[2][3];
    split_step(0, ap);
    split_step(1, ap);
#ifdef LN_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long wall0 = wall_clock64();
    unsigned t_prev = LN_R32_CYC();
#endif
    u32x4 fb0[NTW][3], fb1[NTW][3];  // fragments of the first / second K-step
    // products of K-step s of one chunk: the NTW accumulation chains advance together (product p of every column tile, then
    // product p + 1); small terms first, the dominant product last
    auto products = [&](bf16x8 (&a)[3], u32x4 (&f)[NTW][3]) {
#if LN_CONV_R32_PRIO
        __builtin_amdgcn_s_setprio(LN_CONV_R32_PRIO);
#endif
#if LN_CONV_R32_PROBE & 1
        acc[0][0] += __uint_as_float(f[0][0][0] ^ f[NTW - 1][1][1] ^ f[0][2][2]) + (float)a[0][0] + (float)a[1][1] + (float)a[2][2];
#else
#define LN_R32_PRODUCT(PA, PB)                                                                                                        \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                                                                \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], __builtin_bit_cast(bf16x8, f[nt][PB]), acc[nt], 0, 0, 0);
        LN_R32_PRODUCT(2, 0) LN_R32_PRODUCT(0, 2) LN_R32_PRODUCT(1, 1) LN_R32_PRODUCT(1, 0) LN_R32_PRODUCT(0, 1) LN_R32_PRODUCT(0, 0)
#undef LN_R32_PRODUCT
#endif
#if LN_CONV_R32_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // the walk's next three chunks, advanced once per iteration (no division in the loop); past the end they stay on the last chunk
    int w_e[3], w_kc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) w_e[k] = chunk_e(k + 1), w_kc[k] = chunk_kc(k + 1);
    auto advance = [&](int it) {  // after iteration `it`: w[k] becomes chunk it + 2 + k
        w_e[0] = w_e[1], w_kc[0] = w_kc[1];
        w_e[1] = w_e[2], w_kc[1] = w_kc[2];
        if (it + 4 < total) {
            if (++w_kc[2] == NKC) {
                w_kc[2] = 0;
                ++w_e[2];
            }
        }
    };
    auto requests = [&](int it, int first, int last) {  // pieces [first, last) of: bank chunk it + 1, rows of chunk it + 2
        const int e1 = w_e[0], kc1 = w_kc[0];
        const int e2 = w_e[1], kc2 = w_kc[1];
#pragma unroll
        for (int k = first; k < last; ++k) {
            if (k < BPW) {
#if !(LN_CONV_R32_PROBE & 8)
                dma_piece(k, e1, kc1, (it + 1) & 1);
#endif
            } else {
                // (CH == 1: into the private region the rows of chunk it + 1 have just been read from, once those reads have returned)
                if (CH == 1 && k == BPW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_piece(k, e2, kc2, it & 1);
            }
        }
    };
    auto iteration = [&](int it, bf16x8 (&cur)[2][3], bf16x8 (&nxt)[2][3]) {
        LN_R32_PHASE(0);  // tail of the previous chunk (loop control, address arithmetic)
        // everything requested so far has landed behind this wait + barrier: the bank chunk `it` and the rows of chunk it + 1;
        // and everybody is done reading the buffers the next requests overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LN_R32_PHASE(1);  // waiting for the chunks
        __builtin_amdgcn_s_barrier();
        LN_R32_PHASE(2);  // waiting for the other waves
        const u32x4* sb = &s_b[it & 1][0] + ch * (NTW * 3 * 64) + lane;  // fragment ((s * NT + ch * NTW + nt) * 3 + part) * 64 + lane
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int x = 0; x < 3; ++x) fb0[nt][x] = sb[(nt * 3 + x) * 64];
        read_a((it + 1) & 1);  // rows of chunk it + 1
        LN_R32_PHASE(3);  // LDS reads issued
#if LN_CONV_R32_SKEW
        if (it > 0) products(nxt[1], fb1);  // second K-step of chunk it - 1 (its operands are in the other register set)
        LN_R32_PHASE(4);  // held-back products issued
        requests(it, 0, LN_CONV_R32_SPREAD ? NPIECES / 2 : NPIECES);
        load_ids_of_slot(w_e[2]);
        LN_R32_PHASE(5);  // requests issued
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int x = 0; x < 3; ++x) fb1[nt][x] = sb[((NT + nt) * 3 + x) * 64];
        products(cur[0], fb0);
        LN_R32_PHASE(6);  // first K-step's products issued
        split_step(0, nxt);
        if (LN_CONV_R32_SPREAD) requests(it, NPIECES / 2, NPIECES);
        split_step(1, nxt);
        LN_R32_PHASE(7);  // next rows split
#else
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int x = 0; x < 3; ++x) fb1[nt][x] = sb[((NT + nt) * 3 + x) * 64];
        products(cur[0], fb0);
        LN_R32_PHASE(4);  // first K-step's products issued
        requests(it, 0, LN_CONV_R32_SPREAD ? NPIECES / 2 : NPIECES);
        load_ids_of_slot(w_e[2]);
        LN_R32_PHASE(5);  // requests issued
        split_step(0, nxt);
        products(cur[1], fb1);
        LN_R32_PHASE(6);  // second K-step's products issued
        if (LN_CONV_R32_SPREAD) requests(it, NPIECES / 2, NPIECES);
        split_step(1, nxt);
        LN_R32_PHASE(7);  // next rows split
#endif
        advance(it);
    };
    for (int it = 0; it < total; it += 2) {
        iteration(it, ap, ap_b);
        if (it + 1 < total) iteration(it + 1, ap_b, ap);
    }
#if LN_CONV_R32_SKEW
    if (total & 1) products(ap[1], fb1);
    else products(ap_b[1], fb1);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant last requests must not outlive the workgroup's LDS
#ifdef LN_STAMPS
    if (g_ln_stamps_conv && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long* o = g_ln_stamps_conv + ((size_t)blockIdx.x * W + wave) * 12;
        o[0] = wall0;
        o[1] = wall_clock64();
        o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
        for (int k = 0; k < 8; ++k) o[3 + k] = ph[k];
        o[11] = total;
    }
#endif
    // accumulator register r of lane (i, h): row 8 (r >> 2) + 4 h + (r & 3), column i
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (row < m) out[(size_t)row * f_total + f_off + nt * 32 + i] = acc[nt][r];
        }
    }
}

// Split-K form of the wide convolution (k_conv_rows32sk_b3): the two waves of a row tile share the COLUMNS and split the two K-steps of
// every 32-channel chunk between them (wave kh takes channels 16 kh .. 16 kh + 15), each with its own accumulators for all NT column
// tiles; the pair's partial sums meet once, through LDS, behind the slot loop.  Against the column-split form (k_conv_rows32_b3 with
// CH = 2): a wave splits only the 8 floats per lane of ITS K-step (the operand split is no longer done twice), reads half the bank
// fragments, and a workgroup of 6 x 2 waves covers 192 rows for ANY number of column tiles — 96 filters (three tiles) had to run as
// four waves x 128 rows with two workgroups per CU (0.31 of the matrix pipe against 0.43 at 128 filters).
template <int V, int NT, int RT, bool FLIP>
__global__ void __launch_bounds__(128 * RT) __attribute__((amdgpu_waves_per_eu((2 * RT + 3) / 4, (2 * RT) % 6 == 0 ? 3 : 2)))
    k_conv_rows32sk_b3(const int* __restrict__ nbr, const float* __restrict__ values, const u32x4* __restrict__ bank, int m, int E,
                       float* __restrict__ out, int f_total, int f_off, int e_per) {
    const int e_begin = blockIdx.z * e_per;
    const int e_end = min(E, e_begin + e_per);
    out += (size_t)blockIdx.z * m * f_total;
    constexpr int W = 2 * RT;                // waves
    constexpr int R = 32 * RT;               // rows
    constexpr int NKC = V / 32;              // channel chunks per slot
    constexpr int BCH16 = 2 * NT * 3 * 64;   // 16-byte fragments of one bank chunk
    constexpr int BPIECES = BCH16 / 64;
    constexpr int BPW = (BPIECES + W - 1) / W;
    constexpr int APW = 2;                   // A pieces per wave (of the four of its row tile)
    constexpr int NPIECES = BPW + APW;
    static_assert(V % 32 == 0 && NT >= 1 && NT <= 4 && W <= 16, "rows32sk shape");
    // the partial sums of the kh = 1 waves (RT tiles x NT x 16 registers x 64 lanes x 4 bytes) are parked over s_b and s_a at the end
    constexpr int PARK16 = RT * NT * 16 * 64 / 4;
    constexpr int POOL16 = (2 * BCH16 + 2 * RT * 256) > PARK16 ? (2 * BCH16 + 2 * RT * 256) : PARK16;
    __shared__ u32x4 s_pool[POOL16];
    u32x4* s_b = s_pool;                     // [2][BCH16]
    u32x4* s_a = s_pool + 2 * BCH16;         // [2][RT][256]: per row tile 32 rows x 8 positions of 16 bytes
    __shared__ int s_nbr[R * LN_CONV_LDS_E];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave >> 1, kh = wave & 1;
    const int i = lane & 31;
    const int h = lane >> 5;
    const int m0 = blockIdx.x * R + rt * 32;
    f_off += blockIdx.y * (32 * NT);
    {
        const size_t g0 = (size_t)blockIdx.x * R * E, g_end = (size_t)m * E;
        for (int x = tid; x < R * E; x += 64 * W) s_nbr[x] = (g0 + x < g_end) ? nbr[g0 + x] : -1;
    }
    auto slot_of = [&](int e) -> int { return (FLIP && e < E - 1) ? (e ^ 1) : e; };
    const int a_row = lane >> 3;  // A piece j of a row tile = rows 8j .. 8j+7; this wave requests pieces 2 kh and 2 kh + 1
    const int* my_ids = s_nbr + (rt * 32 + kh * 16 + a_row) * E;
    const unsigned int lds_b = __builtin_amdgcn_readfirstlane(ln_lds_addr(s_b));
    const unsigned int lds_a = __builtin_amdgcn_readfirstlane(ln_lds_addr(s_a) + (unsigned)(rt * 4096 + kh * 2048));
    const u32x4* bank_y = bank + (size_t)blockIdx.y * NKC * BCH16 + lane;
    const size_t bank_e = (size_t)gridDim.y * NKC * BCH16;
    int a_off[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) a_off[j] = (((lane & 7) ^ (((8 * (kh * APW + j) + a_row) >> 1) & 7)) * 4);
    int ids[APW];
    auto dma_piece = [&](int k, int e_n, int kc_n, int buf_n) {
        if (k < BPW) {
            const int p = wave + W * k;
            if (BPIECES % W == 0 || p < BPIECES)
                ln_glds16(bank_y + (size_t)e_n * bank_e + (size_t)kc_n * BCH16 + p * 64, lds_b + (unsigned)(buf_n * BCH16 + p * 64) * 16u);
        } else {
            const int j = k - BPW;
            const int nb = ids[j];
            const float* src = (nb >= 0 ? values + (size_t)nb * V : g_ln_zero_row) + kc_n * 32 + a_off[j];
            ln_glds16(src, lds_a + (unsigned)(buf_n * (RT * 4096) + j * 1024));
        }
    };
    floatx16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int total = (e_end - e_begin) * NKC;
    auto chunk_e = [&](int t) -> int { return e_begin + min(t, total - 1) / NKC; };
    auto chunk_kc = [&](int t) -> int { return min(t, total - 1) % NKC; };
    auto load_ids_of_slot = [&](int e_) {
        const int sl = slot_of(e_);
#pragma unroll
        for (int j = 0; j < APW; ++j) ids[j] = my_ids[j * 8 * E + sl];
    };
    const int a_sw = (i >> 1) & 7;
    u32x4 araw[2];  // this wave's K-step of the NEXT chunk's rows: channels 16 kh + 8 h .. + 7 of row i
    auto read_a = [&](int buf) {
        const u32x4* my_a = s_a + (buf * RT + rt) * 256 + i * 8;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) araw[j2] = my_a[(kh * 4 + h * 2 + j2) ^ a_sw];
    };
    auto split = [&](bf16x8 (&dst)[3]) {
        u32x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = __uint_as_float(araw[j >> 1][(2 * j) & 3]), x1 = __uint_as_float(araw[j >> 1][(2 * j + 1) & 3]);
            const float r10 = x0 - __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u), r11 = x1 - __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
            const float r20 = r10 - __uint_as_float(__float_as_uint(r10) & 0xFFFF0000u), r21 = r11 - __uint_as_float(__float_as_uint(r11) & 0xFFFF0000u);
            p1[j] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
            p2[j] = __builtin_amdgcn_perm(__float_as_uint(r11), __float_as_uint(r10), 0x07060302u);
            p3[j] = __builtin_amdgcn_perm(__float_as_uint(r21), __float_as_uint(r20), 0x07060302u);
        }
        dst[0] = __builtin_bit_cast(bf16x8, p1), dst[1] = __builtin_bit_cast(bf16x8, p2), dst[2] = __builtin_bit_cast(bf16x8, p3);
    };
    // the walk's next three chunks (as in k_conv_rows32_b3)
    int w_e[3], w_kc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) w_e[k] = chunk_e(k + 1), w_kc[k] = chunk_kc(k + 1);
    auto advance = [&](int it) {
        w_e[0] = w_e[1], w_kc[0] = w_kc[1];
        w_e[1] = w_e[2], w_kc[1] = w_kc[2];
        if (it + 4 < total) {
            if (++w_kc[2] == NKC) {
                w_kc[2] = 0;
                ++w_e[2];
            }
        }
    };
    {   // a tile without any neighbour (past the real rows of a static-rows launch) writes zeros and leaves (see k_conv_rows32_b3)
        bool any = false;
        for (int x = tid; x < R * E; x += 64 * W) any |= s_nbr[x] >= 0;
        if (!__syncthreads_or(any)) {
            for (int x = tid; x < R * (32 * NT / 4); x += 64 * W) {
                const int row = blockIdx.x * R + x / (32 * NT / 4), c4 = x % (32 * NT / 4);
                if (row < m) *reinterpret_cast<float4*>(out + (size_t)row * f_total + f_off + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
    }
    load_ids_of_slot(chunk_e(0));
#pragma unroll
    for (int k = 0; k < NPIECES; ++k) dma_piece(k, chunk_e(0), chunk_kc(0), 0);
    load_ids_of_slot(chunk_e(1));
#pragma unroll
    for (int k = BPW; k < NPIECES; ++k) dma_piece(k, chunk_e(1), chunk_kc(1), 1);
    load_ids_of_slot(chunk_e(2));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16x8 ap[3], ap_b[3];
    split(ap);
    auto iteration = [&](int it, bf16x8 (&cur)[3], bf16x8 (&nxt)[3]) {
        LN_R32_PHASE(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bank chunk `it` and the rows of chunk it + 1 have landed ...
        __builtin_amdgcn_s_barrier();                      // ... everybody's, and everybody is done with the buffers overwritten next
        LN_R32_PHASE(2);
        const u32x4* sb = s_b + (it & 1) * BCH16 + kh * (NT * 3 * 64) + lane;  // fragment ((kh * NT + nt) * 3 + part) * 64 + lane
        u32x4 fb[NT][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int x = 0; x < 3; ++x) fb[nt][x] = sb[(nt * 3 + x) * 64];
        read_a((it + 1) & 1);
        LN_R32_PHASE(3);
        // the NT accumulation chains advance together; small terms first, the dominant product last
#define LN_R32SK_PRODUCT(PA, PB)                                                                                                       \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                                                  \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[PA], __builtin_bit_cast(bf16x8, fb[nt][PB]), acc[nt], 0, 0, 0);
        LN_R32SK_PRODUCT(2, 0) LN_R32SK_PRODUCT(0, 2) LN_R32SK_PRODUCT(1, 1) LN_R32SK_PRODUCT(1, 0) LN_R32SK_PRODUCT(0, 1) LN_R32SK_PRODUCT(0, 0)
#undef LN_R32SK_PRODUCT
        LN_R32_PHASE(4);
        // requests: bank chunk it + 1, rows of chunk it + 2
#pragma unroll
        for (int k = 0; k < BPW; ++k) dma_piece(k, w_e[0], w_kc[0], (it + 1) & 1);
#pragma unroll
        for (int k = BPW; k < NPIECES; ++k) dma_piece(k, w_e[1], w_kc[1], it & 1);
        load_ids_of_slot(w_e[2]);
        LN_R32_PHASE(5);
        split(nxt);
        LN_R32_PHASE(7);
        advance(it);
    };
    for (int it = 0; it < total; it += 2) {
        iteration(it, ap, ap_b);
        if (it + 1 < total) iteration(it + 1, ap_b, ap);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant last requests must not outlive the workgroup's LDS
    __builtin_amdgcn_s_barrier();                      // nobody reads s_b / s_a any more: the pool becomes the parking area
    floatx4* park = reinterpret_cast<floatx4*>(s_pool) + (size_t)rt * NT * 4 * 64 + lane;  // [rt][nt][quad][lane] x 16 bytes
    if (kh == 1) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                park[(nt * 4 + qd) * 64] = floatx4{acc[nt][4 * qd], acc[nt][4 * qd + 1], acc[nt][4 * qd + 2], acc[nt][4 * qd + 3]};
    }
    __syncthreads();
    if (kh == 0) {
        // accumulator register r of lane (i, h): row 8 (r >> 2) + 4 h + (r & 3), column i
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const floatx4 o = park[(nt * 4 + qd) * 64];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int row = m0 + 8 * qd + 4 * h + r4;
                    if (row < m) out[(size_t)row * f_total + f_off + nt * 32 + i] = acc[nt][4 * qd + r4] + o[r4];
                }
            }
        }
    }
}

// (Round-2 first attempt, kept for the record: the same kernel on v_mfma_f32_16x16x32_bf16 with 3-way split operands — a = a1 + a2 + a3 in
// bf16, six products of total order <= 4, fp32 accumulation; accurate to the same 1e-5 bar — with the filter bank pre-split
// once per call.  2.5x less matrix time on paper, but the per-slot kernel is not bound by the matrix pipe even at 46 k
// vertices x 128 channels (50 % of the fp32 MFMA peak): 91 vs 87 us there, whole LNN step 6.0 vs 5.5 ms.  What bounds it is the
// slot loop itself: gather + 32-48 KB of bank staging + two workgroup barriers per slot.)
// Small-filter fast path (E*V*F*4 <= 64 KiB, e.g. V = F = 32 with E = 9): the WHOLE filter bank is
// staged into LDS once per workgroup (one barrier), and every lane issues the gathers of all E
// neighbour rows up front, so the kernel pays one memory latency instead of E.
template <int V, int NT, int E, bool FLIP, bool WT>
__global__ void __launch_bounds__(256)
    k_conv_mfma_full(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ filter, int m,
                     float* __restrict__ out, int conv_blocks, const float* __restrict__ slab_partial, int nslabs, int slab_total,
                     float* __restrict__ slab_out) {
    // Horizontal fusion for the backward pass: workgroups past conv_blocks sum the filter-gradient slabs of the launch
    // before this one (independent of the convolution; slab_out == nullptr: plain convolution).
    if (slab_out != nullptr && (int)blockIdx.x >= conv_blocks) {
        ln_sum_slabs_body<false>(blockIdx.x - conv_blocks, slab_partial, nslabs, slab_total, slab_total, slab_out);
        return;
    }
    constexpr int F = 16 * NT;
    constexpr int KQ = V / 4;
    static_assert(E * V * F * 4 <= 64 * 1024, "filter bank must fit 64 KiB of LDS");
    __shared__ __attribute__((aligned(16))) float s_b[E * V * F];  // [((e*KQ + kk)*NT + nt)*64 + lane]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int m0 = blockIdx.x * 64 + wave * 16;
    const int my_row = m0 + i;

    LN_CSTAMP(0);
    // Load order = dependency order: the filter bank depends on nothing, so its loads go out first, then the neighbour
    // indices, then the gathers that need them.  Loads return in order: staging the bank only waits for the bank (the gathers
    // stay in flight behind it), and the MFMAs of slot e only wait for the gather of slot e.
    constexpr int N4 = E * V * F / 4;
    constexpr int NST = (N4 + 255) / 256;
    float4 wv[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int x4 = tid + s * 256;
        wv[s] = (x4 < N4) ? reinterpret_cast<const float4*>(filter)[x4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int nb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) nb[e] = (my_row < m) ? nbr[(size_t)my_row * E + ((FLIP && e < E - 1) ? (e ^ 1) : e)] : -1;
    // unconditional gathers (absent neighbours read row 0 and are zeroed afterwards): no branch, so all
    // E gathers are in flight together
    float a[E][KQ];
#pragma unroll
    for (int e = 0; e < E; ++e) ln_load_quarter<KQ>(values + (size_t)(nb[e] >= 0 ? nb[e] : 0) * V + q * KQ, a[e]);
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int x4 = tid + s * 256;
        if (x4 < N4) {
            const int x = x4 * 4;
            if constexpr (!WT) {
                const int ek = x / F;       // e*V + k
                const int f = x - ek * F;   // multiple of 4
                const int e = ek / V;
                const int k = ek - e * V;
                const int qq = k / KQ;
                const int kk = k - qq * KQ;
                float* dst = s_b + (((e * KQ + kk) * NT + (f >> 4)) * 64 + qq * 16 + (f & 15));
                *reinterpret_cast<float4*>(dst) = wv[s];
            } else {
                const int ef = x / V;       // e*F + f  (rows of the differentiated op's bank)
                const int k0 = x - ef * V;  // multiple of 4
                const int e = ef / F;
                const int f = ef - e * F;
                const float vals4[4] = {wv[s].x, wv[s].y, wv[s].z, wv[s].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    const int qq = k / KQ;
                    const int kk = k - qq * KQ;
                    s_b[((e * KQ + kk) * NT + (f >> 4)) * 64 + qq * 16 + (f & 15)] = vals4[j];
                }
            }
        }
    }
    __syncthreads();
    LN_CSTAMP(1);
    floatx4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const float av = nb[e] >= 0 ? a[e][kk] : 0.f;  // absent neighbour: a zero row
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float b = s_b[((e * KQ + kk) * NT + nt) * 64 + lane];
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[nt], 0, 0, 0);
            }
        }
    }
    LN_CSTAMP(2);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) out[(size_t)row * F + nt * 16 + i] = acc[nt][r];
        }
    }
    LN_CSTAMP(3);
}

// Any (V, F): one thread per output element.
__global__ void __launch_bounds__(256)
    k_conv_generic(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ filter, long long work,
                   int E, int V, int F, int flip, int wt, float* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long mrow = g / F;
    const int f = int(g - mrow * F);
    float acc = 0.0f;
    for (int e = 0; e < E; ++e) {
        const int nb = nbr[mrow * E + ((flip && e < E - 1) ? (e ^ 1) : e)];
        if (nb < 0) continue;
        const float* vr = values + (size_t)nb * V;
        if (wt) {
            const float* wr = filter + ((size_t)e * F + f) * V;
            for (int v = 0; v < V; ++v) acc = fmaf(vr[v], wr[v], acc);
        } else {
            const float* wr = filter + (size_t)e * V * F + f;
            for (int v = 0; v < V; ++v) acc = fmaf(vr[v], wr[(size_t)v * F], acc);
        }
    }
    out[g] = acc;
}

// Output columns are produced in chunks of 16*NT (NT in {8, 4, 2, 1}; the per-slot filter slice V x 16NT must fit LDS),
// so any nr_filters that is a multiple of 16 runs on the matrix cores.
// out[i] = sum over s of partial[s * total + i], float4 per thread (total % 4 == 0)
__global__ void __launch_bounds__(256) ln_k_sum_partials(const float* __restrict__ partial, int nslabs, long long total4, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 acc = reinterpret_cast<const float4*>(partial)[i];
    for (int sl = 1; sl < nslabs; ++sl) {
        const float4 v = reinterpret_cast<const float4*>(partial)[(long long)sl * total4 + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = acc;
}

// Split of the per-slot convolution over the filter slots: 1 (no split) while the vertex tiles alone fill the chip.
#ifndef LN_CONV_SPLIT_TILES
#define LN_CONV_SPLIT_TILES 512  // workgroups aimed at (two per CU)
#endif
template <int V>
static int ln_conv_slots_per_split(int m, int E, int nr_filters) {
    constexpr int NT_MAX = (V * 16 * 8 * 4 <= 32 * 1024) ? 8 : ((V * 16 * 4 * 4 <= 48 * 1024) ? 4 : 2);
    const long long tiles = (long long)ln_div_up(m, 64) * ln_div_up(nr_filters, 16 * NT_MAX);
    if (tiles * 2 > LN_CONV_SPLIT_TILES || E < 2) return E;
    int nsplit = int((LN_CONV_SPLIT_TILES + tiles - 1) / tiles);
    if (nsplit > E) nsplit = E;
    return (E + nsplit - 1) / nsplit;  // slots per workgroup
}
#ifndef LN_CONV_WIDE_SPLIT_MIN_V
#define LN_CONV_WIDE_SPLIT_MIN_V 128
#endif
#ifndef LN_CONV_B3_MIN_ROWS
#define LN_CONV_B3_MIN_ROWS 4096
#endif
static size_t ln_conv_bank_bytes(int m, int E, int val_dim, int nr_filters);
// Slot split of the WIDE form on mid-size lattices (0 / 1: not taken), from 128 gathered channels on where the 192-row workgroups of
// the unsplit wide form would fill less than half the chip.  LN_CONV_WIDE_SPLIT=0 switches it off, =N forces N (A/B; read once).
// Measured at 11.4 k rows (level 2 of the SemanticKITTI network; tools/conv_time.py --coarse 1, us per call incl. bank split and
// partial sum): 128 -> 128 46.9 -> 35.7, 256 -> 256 296 -> 109, 192 -> 192 191 -> 76, 256 -> 128 157 -> 55, 128 -> 64 32.2 -> 26.3.
template <int V>
static int ln_conv_wide_split(int m, int E, int nr_filters, bool have_bank) {
    static int knob = -2;
    if (knob == -2) {
        const char* e = getenv("LN_CONV_WIDE_SPLIT");
        knob = e ? atoi(e) : -1;
    }
    if (knob == 0 || !have_bank || V % 32 != 0 || V < LN_CONV_WIDE_SPLIT_MIN_V || nr_filters % 32 != 0 || E < 3 || m < LN_CONV_B3_MIN_ROWS) return 0;
    if ((long long)ln_div_up(m, 192) * ln_div_up(nr_filters, 128) >= LN_BWD_CUS / 2) return 0;  // the unsplit wide form already runs
    if (knob > 1) return knob <= E ? knob : E;
    // workgroups of the widest launch: the 128-column chunks go out together, a narrower rest as a launch of its own
    const long long wgs = (long long)ln_div_up(m, 192) * (nr_filters >= 128 ? nr_filters / 128 : 1);
    // rounds of one workgroup per CU x slots walked per workgroup; the smallest split among the cheapest (fewer partial slabs)
    int best = 1;
    long long best_cost = 1ll << 60;
    for (int n = 2; n <= E; ++n) {
        const long long cost = ((wgs * n + LN_BWD_CUS - 1) / LN_BWD_CUS) * ((E + n - 1) / n);
        if (cost < best_cost) {
            best = n;
            best_cost = cost;
        }
    }
    return best;
}
static int ln_conv_wide_split_rt(int m, int E, int val_dim, int nr_filters) {
    const bool bank = ln_conv_bank_bytes(m, E, val_dim, nr_filters) > 0;
    switch (val_dim) {
        case 96: return ln_conv_wide_split<96>(m, E, nr_filters, bank);
        case 128: return ln_conv_wide_split<128>(m, E, nr_filters, bank);
        case 192: return ln_conv_wide_split<192>(m, E, nr_filters, bank);
        case 256: return ln_conv_wide_split<256>(m, E, nr_filters, bank);
        default: return 0;
    }
}
static int ln_conv_slots_per_split_rt(int m, int E, int val_dim, int nr_filters) {
    switch (val_dim) {
        case 8: return ln_conv_slots_per_split<8>(m, E, nr_filters);
        case 16: return ln_conv_slots_per_split<16>(m, E, nr_filters);
        case 32: return ln_conv_slots_per_split<32>(m, E, nr_filters);
        case 48: return ln_conv_slots_per_split<48>(m, E, nr_filters);
        case 64: return ln_conv_slots_per_split<64>(m, E, nr_filters);
        case 96: return ln_conv_slots_per_split<96>(m, E, nr_filters);
        case 128: return ln_conv_slots_per_split<128>(m, E, nr_filters);
        case 192: return ln_conv_slots_per_split<192>(m, E, nr_filters);
        case 256: return ln_conv_slots_per_split<256>(m, E, nr_filters);
        default: return E;
    }
}

// bf16x3 path: channel counts that are multiples of 32, lattices large enough to be matrix-bound, LN_CONV_EXACT_F32=1 switches
// it off (A/B; read once)
#ifndef LN_CONV_B3_MIN_ROWS
#define LN_CONV_B3_MIN_ROWS 4096
#endif
static bool ln_conv_b3_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LN_CONV_EXACT_F32");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}
static bool ln_conv_rows32_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LN_CONV_ROWS32");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static size_t ln_conv_bank_bytes(int m, int E, int val_dim, int nr_filters) {
    if (val_dim % 32 != 0 || nr_filters % 16 != 0 || m < LN_CONV_B3_MIN_ROWS || E > LN_CONV_LDS_E || !ln_conv_b3_enabled()) return 0;
    return (((size_t)E * val_dim * nr_filters * 3 * sizeof(unsigned short)) + 255) & ~size_t(255);
}

extern "C" size_t ln_conv_bank_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters) {
    if (m <= 0 || nr_filters % 16 != 0) return 0;
    const bool small_filter = filter_extent == 9 && (size_t)filter_extent * val_dim * nr_filters * 4 <= 64 * 1024 && val_dim <= 32;
    return small_filter ? 0 : ln_conv_bank_bytes(m, filter_extent, val_dim, nr_filters);
}

extern "C" size_t ln_conv_forward_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters) {
    if (m <= 0 || nr_filters % 16 != 0) return 256;
    int e_per = ln_conv_slots_per_split_rt(m, filter_extent, val_dim, nr_filters);
    const int ws_ = ln_conv_wide_split_rt(m, filter_extent, val_dim, nr_filters);
    if (ws_ > 1) e_per = (filter_extent + ws_ - 1) / ws_;
    const int nsplit = (filter_extent + e_per - 1) / e_per;
    // + the filter bank split into three bf16 parts (bf16x3 path of the per-slot kernel; not the small-filter fast path)
    const bool small_filter = filter_extent == 9 && (size_t)filter_extent * val_dim * nr_filters * 4 <= 64 * 1024 && val_dim <= 32;
    const size_t bank = small_filter ? 0 : ln_conv_bank_bytes(m, filter_extent, val_dim, nr_filters);
    return bank + (nsplit > 1 ? (size_t)nsplit * m * nr_filters * sizeof(float) : 0) + 256;
}

// Sub-tiles per workgroup of the bf16x3 per-slot kernel: 3 (one 768-thread workgroup per CU) where the kernel runs at three waves per
// SIMD and there are at least as many such workgroups as CUs; LN_CONV_B3_T=1|3 forces it (A/B, read once).
template <int V>
static int ln_conv_b3_subtiles(int m, int chunks) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("LN_CONV_B3_T");
        forced = e ? atoi(e) : 0;
    }
    if (LN_CONV_B3_WAVES(V) != 3) return 1;
    if (forced == 1 || forced == 3) return forced;
    return (long long)ln_div_up(m, 192) * chunks >= LN_BWD_CUS * 3 / 4 ? 3 : 1;
}

// LN_CONV_BANK_READY of the call in progress: the workspace already holds the split bank of this filter (written by an earlier call
// with the same sizes and flags), so the k_conv_split_bank launches are skipped
static thread_local bool g_ln_bank_ready = false;
// ln_conv_row_partition: LnTable.row_regions of the space-ordered table the next convolutions of this thread run over (device memory,
// read by the kernels; nullptr = none).  A placement hint only — ln_partition_tile is a bijection whatever the array holds.
static thread_local const int* g_ln_row_partition = nullptr;
extern "C" int ln_conv_row_partition(const int* row_starts) {
    g_ln_row_partition = row_starts;
    return LN_OK;
}
// slab sum waiting for a split launch to ride in (set by ln_conv_backward around its value-gradient convolution)
static thread_local LnSlabSum g_ln_slab_job = {nullptr, 0, 0, nullptr};
static LnSlabSum ln_take_slab_job() {
    const LnSlabSum j = g_ln_slab_job;
    g_ln_slab_job = LnSlabSum{nullptr, 0, 0, nullptr};
    return j;
}
static int ln_slab_extra_blocks(const LnSlabSum& j, int planes) { return j.partial ? ln_div_up(j.total / 64, planes) : 0; }

template <int V, bool FLIP, bool WT>
static bool ln_conv_launch_v(int nr_filters, const int* nbr, const float* values, const float* filter, int m, int E, float* out,
                             void* workspace, size_t workspace_bytes, hipStream_t st) {
    const dim3 block(256);
    constexpr int NT_MAX = (V * 16 * 8 * 4 <= 32 * 1024) ? 8 : ((V * 16 * 4 * 4 <= 48 * 1024) ? 4 : 2);
    // workspace: [filter bank split into three bf16 parts (bf16x3 path)] [partial slabs of the slot split]
    const bool ws_ok = workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0;
    const size_t bank_bytes = ln_conv_bank_bytes(m, E, V, nr_filters);
    const bool b3 = bank_bytes > 0 && ws_ok && workspace_bytes >= bank_bytes;
    unsigned short* bank = b3 ? static_cast<unsigned short*>(workspace) : nullptr;
    char* slab_ws = ws_ok ? static_cast<char*>(workspace) + (b3 ? bank_bytes : 0) : nullptr;
    const size_t slab_bytes = ws_ok ? workspace_bytes - (b3 ? bank_bytes : 0) : 0;
    int e_per = ln_conv_slots_per_split<V>(m, E, nr_filters);
    int nsplit = (E + e_per - 1) / e_per;
    // Mid-size lattices with wide rows (coarse levels of a U-net: 5-30 k rows x 128+ channels): too few 192-row workgroups for the wide
    // form, and the 16-row kernels re-gather every row once per column chunk.  There the wide form runs with the slots split over
    // gridDim.z (its workgroups then fill the chip) and the partial sums are added by the launch behind it.
    const int wide_split = ln_conv_wide_split<V>(m, E, nr_filters, bank_bytes > 0 && ws_ok && workspace_bytes >= bank_bytes);
    if (wide_split > 1) {
        e_per = (E + wide_split - 1) / wide_split;
        nsplit = (E + e_per - 1) / e_per;
    }
    if (nsplit > 1 && (!slab_ws || slab_bytes < (size_t)nsplit * m * nr_filters * sizeof(float))) {
        e_per = E;  // no room for the partial slabs: one workgroup walks all slots
        nsplit = 1;
    }
    const bool wide_mid = wide_split > 1 && nsplit > 1;
    float* dst = nsplit > 1 ? reinterpret_cast<float*>(slab_ws) : out;
    int f_off = 0;
    size_t bank_off = 0;  // bf16 elements
    // wide form (both operands by LDS-DMA, 32-row MFMA tiles, every output column in one pass over the gathered rows): column
    // chunks of 128, then one narrower chunk.  LN_CONV_ROWS32=0 keeps the 16-row kernels (A/B; read once)
    if constexpr (V % 32 == 0) {
        // taken from 96 gathered channels on (below, the 16-row kernels' gathers are as fast: 64 x 64 27 vs 29 us at 46 k rows) and
        // while the 192-row workgroups alone fill half the chip (no slot split in this form)
        if (b3 && nr_filters % 32 == 0 && V >= 96 && ln_conv_rows32_enabled() &&
            (wide_mid || (nsplit == 1 && (long long)ln_div_up(m, 192) * ln_div_up(nr_filters, 128) >= LN_BWD_CUS / 2))) {
#define LN_CONV_R32SK(NTC, RTT)                                                                                                     \
    {                                                                                                                               \
        const int cnt = (nr_filters - f_off) / (32 * NTC);                                                                          \
        if (cnt > 0) {                                                                                                              \
            if (!g_ln_bank_ready) {                                                                                                 \
                const LnSlabSum job_ = ln_take_slab_job();                                                                          \
                const int sx_ = ln_div_up(V * 32 * NTC, 256);                                                                       \
                LN_LAUNCH("k_conv_split_bank", (k_conv_split_bank32<V, NTC, WT>), dim3(sx_ + ln_slab_extra_blocks(job_, E * cnt), E, cnt), block, 0, st, \
                          filter, nr_filters, f_off, bank + bank_off, sx_, job_);                                                    \
            }                                                                                                                       \
            LN_LAUNCH("k_conv_mfma", (k_conv_rows32sk_b3<V, NTC, RTT, FLIP>), dim3(ln_div_up(m, 32 * RTT), cnt, nsplit),            \
                      dim3(128 * RTT), 0, st, nbr, values, reinterpret_cast<const u32x4*>(bank + bank_off), m, E, dst, nr_filters, f_off, e_per); \
            bank_off += (size_t)E * cnt * V * 32 * NTC * 3;                                                                         \
            f_off += cnt * 32 * NTC;                                                                                                \
        }                                                                                                                           \
    }
#define LN_CONV_R32(NTC, NTWW, CHH, RTT)                                                                                            \
    {                                                                                                                               \
        const int cnt = (nr_filters - f_off) / (32 * NTC);                                                                          \
        if (cnt > 0) {                                                                                                              \
            if (!g_ln_bank_ready) {                                                                                                 \
                const LnSlabSum job_ = ln_take_slab_job();                                                                          \
                const int sx_ = ln_div_up(V * 32 * NTC, 256);                                                                       \
                LN_LAUNCH("k_conv_split_bank", (k_conv_split_bank32<V, NTC, WT>), dim3(sx_ + ln_slab_extra_blocks(job_, E * cnt), E, cnt), block, 0, st, \
                          filter, nr_filters, f_off, bank + bank_off, sx_, job_);                                                    \
            }                                                                                                                       \
            LN_LAUNCH("k_conv_mfma", (k_conv_rows32_b3<V, NTWW, CHH, RTT, FLIP>), dim3(ln_div_up(m, 32 * RTT), cnt, nsplit),        \
                      dim3(64 * RTT * CHH), 0, st, nbr, values, reinterpret_cast<const u32x4*>(bank + bank_off), m, E, dst, nr_filters, f_off, e_per); \
            bank_off += (size_t)E * cnt * V * 32 * NTC * 3;                                                                         \
            f_off += cnt * 32 * NTC;                                                                                                \
        }                                                                                                                           \
    }
            // 96 columns (three tiles: no even split of the columns over a pair of waves) take the split-K pairs: 96 -> 96 62.5 -> 54.1 us,
            // 128 -> 96 80.7 -> 69.1 us at 46.5 k rows; at 128 / 64 columns the column-split pairs are faster (80 vs 85, 50.6 vs 52 us:
            // the pair's partial sums cost a pass through LDS at the end).  LN_CONV_R32_SK=0: column-split everywhere (A/B; read once)
            static int sk = -1;
            if (sk < 0) {
                const char* ev = getenv("LN_CONV_R32_SK");
                sk = (ev && ev[0] == '0') ? 0 : 1;
            }
            if (sk == 1 && (nr_filters - f_off) % 128 == 96) {
                LN_CONV_R32(4, 2, 2, 6) LN_CONV_R32SK(3, 6)
            }
            LN_CONV_R32(4, 2, 2, 6) LN_CONV_R32(3, 3, 1, 4) LN_CONV_R32(2, 1, 2, 6) LN_CONV_R32(1, 1, 1, 4)
#undef LN_CONV_R32
#undef LN_CONV_R32SK
        }
    }
    // all chunks of the widest size go out as ONE launch (gridDim.y = their count), then at most one launch per narrower size
#define LN_CONV_CHUNKS(NTC)                                                                                                         \
    if constexpr (NT_MAX >= NTC) {                                                                                                  \
        /* (256 gathered channels x 32 columns spills in the bf16x3 form: those lattices take 16 columns per workgroup) */          \
        const int cnt = (b3 && V >= 256 && NTC > 1) ? 0 : (nr_filters - f_off) / (16 * NTC);                                        \
        if (cnt > 0) {                                                                                                              \
            bool done_b3 = false;                                                                                                   \
            if constexpr (V % 32 == 0 && V * 16 * NTC * 6 <= 64 * 1024 && (V < 256 || NTC == 1)) { /* 256 x 32 columns spills */      \
                if (b3) {                                                                                                           \
                    if (!g_ln_bank_ready) {                                                                                         \
                        const LnSlabSum job_ = ln_take_slab_job();                                                                  \
                        const int sx_ = ln_div_up(V * 16 * NTC, 256);                                                               \
                        LN_LAUNCH("k_conv_split_bank", (k_conv_split_bank<V, NTC, WT>), dim3(sx_ + ln_slab_extra_blocks(job_, E * cnt), E, cnt), block, 0, \
                                  st, filter, nr_filters, f_off, bank + bank_off, sx_, job_);                                        \
                    }                                                                                                               \
                    if (ln_conv_b3_subtiles<V>(m, cnt * nsplit) == 3) {                                                             \
                        if constexpr (LN_CONV_B3_WAVES(V) == 3)                                                                     \
                            LN_LAUNCH("k_conv_mfma", (k_conv_mfma_b3<V, NTC, FLIP, 3>), dim3(ln_div_up(m, 192), cnt, nsplit), dim3(768), 0, st, nbr, \
                                      values, reinterpret_cast<const u32x4*>(bank + bank_off), m, E, dst, nr_filters, f_off, e_per); \
                    } else {                                                                                                        \
                        LN_LAUNCH("k_conv_mfma", (k_conv_mfma_b3<V, NTC, FLIP, 1>), dim3(ln_div_up(m, 64), cnt, nsplit), block, 0, st, nbr, values, \
                                  reinterpret_cast<const u32x4*>(bank + bank_off), m, E, dst, nr_filters, f_off, e_per);            \
                    }                                                                                                               \
                    bank_off += (size_t)E * cnt * V * 16 * NTC * 3;                                                                 \
                    done_b3 = true;                                                                                                 \
                }                                                                                                                   \
            }                                                                                                                       \
            if (!done_b3)                                                                                                           \
                LN_LAUNCH("k_conv_mfma", (k_conv_mfma<V, NTC, FLIP, WT>), dim3(ln_div_up(m, 64), cnt, nsplit), block, 0, st, nbr, values, filter, \
                          m, E, dst, nr_filters, f_off, e_per);                                                                     \
            f_off += cnt * 16 * NTC;                                                                                                \
        }                                                                                                                           \
    }
    LN_CONV_CHUNKS(8) LN_CONV_CHUNKS(4) LN_CONV_CHUNKS(2) LN_CONV_CHUNKS(1)
#undef LN_CONV_CHUNKS
    if (nsplit > 1) {
        const long long total4 = (long long)m * nr_filters / 4;
        LN_LAUNCH("k_conv_sum_partials", ln_k_sum_partials, dim3(ln_div_up(total4, 256)), block, 0, st, (const float*)dst, nsplit, total4, out);
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// Forward of the V = F = 32, E = 9 convolution on the bf16 matrix cores with exactly 3-way split operands, in the workgroup shape of
// the fused backward (T sub-tiles of 64 vertices per workgroup, one workgroup per CU and round: the bank is split and staged once per
// 64 * T vertices instead of once per 64).  out[row][f] = sum_e sum_v values[nbr[row][e]][v] * W[e][v][f]:
//   A = this lane's gathered quarter row (channels 8q..8q+7 = its k-group), split in registers;
//   B = W_e split while the bank is staged: one 16-byte fragment per (slot, column tile, part, lane (f, q)) = W[e][8q..8q+7][f].
// After the one barrier behind the staging the slot loop is gathers, register splits and 12 matrix instructions per slot and wave.
// ------------------------------------------------------------------------------------------
#ifndef LN_FWD_LINE
#define LN_FWD_LINE 1  // 0: the fragment-shaped gathers of rounds 2-4 (A/B)
#endif
template <int T>
__global__ void __launch_bounds__(256 * T) __attribute__((amdgpu_waves_per_eu(T, T)))
    k_conv_forward_b3(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ filter, int m,
                      float* __restrict__ out, const int* __restrict__ row_part) {
    constexpr int V = 32, F = 32, E = 9, KQ = 8, NT = 2;
    constexpr int THREADS = 256 * T;
    constexpr int FRAG16 = E * NT * 3 * 64;  // 16-byte fragments of the split bank (54 KB)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[FRAG16 * 16];
    const u32x4* s_frag = reinterpret_cast<const u32x4*>(s_raw);
#if LN_FWD_LINE
    // Line-shaped gathers (round 5).  Loads shaped like the MFMA fragment — lane (i, q) reads its 32-byte quarter of row i — touch 64
    // different 128-byte lines per wave-instruction and run at half the rate of loads in which 8 adjacent lanes read one whole row
    // (tools/probes/gather_layout_probe.cpp: 3.5 vs 7.0 TB/s whatever the hit rate).  So lane l loads piece l & 7 (16 bytes) of rows
    // l >> 3 and 8 + (l >> 3) of the wave's 16, and the wave re-shapes the 2 KB through a private LDS region: two 16-byte stores, two
    // 16-byte reads per slot.  Piece p of row r sits at position p ^ f(r), f(r) = ((r >> 1) & 7) ^ (((r >> 2) & 1) << 1):
    // conflict-free for the 16-lane groups of ds_read_b128 under the (i, q) mapping (checked exhaustively) and for the stores
    // (8 adjacent lanes = one row).  LDS operations of one wave execute in program order: no barrier, only a compiler fence.
    __shared__ __attribute__((aligned(16))) floatx4 s_x[4 * T][16 * 8];
    __shared__ int s_nbr[64 * T * E];
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int bx = ln_partition_tile(blockIdx.x, gridDim.x, row_part, 64 * T);  // space-ordered table: XCD x takes the tiles of kd region x
    const int m0 = bx * (64 * T) + (tid >> 6) * 16;
    const int my_row = m0 + i;
    // bank staging: one item = (slot e, 8 consecutive rows v of W_e, column f) = ONE 16-byte fragment per split part.  The eight
    // loads of an item are 4 bytes each but consecutive lanes take consecutive columns, so every load instruction reads whole
    // lines; the fragment goes to LDS with three 16-byte stores.  (Staging a float4 of four columns per thread instead scattered
    // its twelve halfs over twelve fragments: 36 two-byte LDS stores per thread; the kernel time at T = 3 is the same, 25.3 -> 23.1 us at T = 1.)
    constexpr int ITEMS = E * (V / 8) * F;
    constexpr int NIT = (ITEMS + THREADS - 1) / THREADS;
    float wv[NIT][8];
#pragma unroll
    for (int s = 0; s < NIT; ++s) {
        const int x = min(tid + s * THREADS, ITEMS - 1);  // (clamped: the loads of the last, partly used round stay unconditional — no branch
        const int f = x % F;                              //  around eight loads; the staging below skips what lies beyond ITEMS)
        const int ev8 = x / F;  // e * (V / 8) + v / 8
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[s][j] = filter[(ev8 * 8 + j) * F + f];
    }
#if LN_FWD_LINE
    {
        const size_t g0 = (size_t)bx * (64 * T) * E, g_end = (size_t)m * E;
        for (int x = tid; x < 64 * T * E; x += THREADS) s_nbr[x] = (g0 + x < g_end) ? nbr[g0 + x] : -1;
    }
    {   // a tile without any neighbour (past the real rows of a static-rows launch: 257 tiles of 192 rows for 256 CUs) writes zeros and leaves
        bool any = false;
        for (int x = tid; x < 64 * T * E; x += THREADS) any |= s_nbr[x] >= 0;
        if (!__syncthreads_or(any)) {
            for (int x = tid; x < 64 * T * (F / 4); x += THREADS) {
                const int row = bx * (64 * T) + x / (F / 4);
                if (row < m) *reinterpret_cast<float4*>(out + (size_t)row * F + (x % (F / 4)) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
    }
    const int wv_ = tid >> 6;
    const int lr = lane >> 3;  // the two rows this lane loads: lr and 8 + lr
    int nb[2][E];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) nb[j][e] = s_nbr[(wv_ * 16 + 8 * j + lr) * E + e];
    auto fsw = [](int r) -> int { return ((r >> 1) & 7) ^ (((r >> 2) & 1) << 1); };
    floatx4* xw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) xw[j] = &s_x[wv_][(8 * j + lr) * 8 + ((lane & 7) ^ fsw(8 * j + lr))];
    const floatx4* xr0 = &s_x[wv_][i * 8 + ((2 * q) ^ fsw(i))];
    const floatx4* xr1 = &s_x[wv_][i * 8 + ((2 * q + 1) ^ fsw(i))];
#else
    int nb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) nb[e] = (my_row < m) ? nbr[(size_t)my_row * E + e] : -1;
#endif
#ifndef LN_FWD_DEPTH
#define LN_FWD_DEPTH 4
#endif
    constexpr int DEPTH = LN_FWD_DEPTH;  // ring of gathered quarter rows: DEPTH - 1 gathers in flight (4: 14.9 us, 7: 15.7, 10 = all nine up front: 17.1)
#if LN_FWD_LINE
    floatx4 a[DEPTH][2];
    auto gather = [&](int e, floatx4 (&dst)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            dst[j] = *reinterpret_cast<const floatx4*>(values + (size_t)(nb[j][e] >= 0 ? nb[j][e] : 0) * V + (lane & 7) * 4);
    };
#pragma unroll
    for (int k = 0; k < DEPTH - 1 && k < E; ++k) gather(k, a[k]);
#else
    float a[DEPTH][KQ];
#pragma unroll
    for (int k = 0; k < DEPTH - 1 && k < E; ++k) ln_load_quarter<KQ>(values + (size_t)(nb[k] >= 0 ? nb[k] : 0) * V + q * KQ, a[k]);
#endif
    // bank -> split -> LDS fragments (16-byte unit ((e * NT + f / 16) * 3 + part) * 64 + (v / 8) * 16 + f % 16 holds rows v..v+7)
    u32x4* s_frag_w = reinterpret_cast<u32x4*>(s_raw);
#pragma unroll
    for (int s = 0; s < NIT; ++s) {
        const int x = tid + s * THREADS;
        if (x < ITEMS) {
            const int f = x % F;
            const int ev8 = x / F;
            const int e = ev8 / (V / 8);
            const int vo = ev8 - e * (V / 8);
            unsigned int h[8], md[8], lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ln_split3_bits(wv[s][j], h[j], md[j], lo[j]);
            const int unit = ((e * NT + (f >> 4)) * 3) * 64 + vo * 16 + (f & 15);
            s_frag_w[unit] = u32x4{ln_pack_hi(h[0], h[1]), ln_pack_hi(h[2], h[3]), ln_pack_hi(h[4], h[5]), ln_pack_hi(h[6], h[7])};
            s_frag_w[unit + 64] = u32x4{ln_pack_hi(md[0], md[1]), ln_pack_hi(md[2], md[3]), ln_pack_hi(md[4], md[5]), ln_pack_hi(md[6], md[7])};
            s_frag_w[unit + 128] = u32x4{ln_pack_hi(lo[0], lo[1]), ln_pack_hi(lo[2], lo[3]), ln_pack_hi(lo[4], lo[5]), ln_pack_hi(lo[6], lo[7])};
        }
    }
    floatx4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
#if LN_FWD_LINE
        if (e + DEPTH - 1 < E) gather(e + DEPTH - 1, a[(e + DEPTH - 1) % DEPTH]);
        // rows of absent neighbours are zeroed by the lane that loaded them; then the wave's 16 x 32 floats change shape through LDS
#pragma unroll
        for (int j = 0; j < 2; ++j) *xw[j] = (nb[j][e] >= 0) ? a[e % DEPTH][j] : floatx4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const floatx4 lo4 = *xr0, hi4 = *xr1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float ae[KQ] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        #else
        float (&ae)[KQ] = a[e % DEPTH];
        if (e + DEPTH - 1 < E)
            ln_load_quarter<KQ>(values + (size_t)(nb[e + DEPTH - 1] >= 0 ? nb[e + DEPTH - 1] : 0) * V + q * KQ, a[(e + DEPTH - 1) % DEPTH]);
#endif
        u32x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int h0, m0_, l0, h1, m1_, l1;
#if LN_FWD_LINE
            ln_split3_bits(ae[2 * j], h0, m0_, l0);
            ln_split3_bits(ae[2 * j + 1], h1, m1_, l1);
#else
            ln_split3_bits(nb[e] >= 0 ? ae[2 * j] : 0.f, h0, m0_, l0);
            ln_split3_bits(nb[e] >= 0 ? ae[2 * j + 1] : 0.f, h1, m1_, l1);
#endif
            p1[j] = ln_pack_hi(h0, h1);
            p2[j] = ln_pack_hi(m0_, m1_);
            p3[j] = ln_pack_hi(l0, l1);
        }
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, p1), a2 = __builtin_bit_cast(bf16x8, p2), a3 = __builtin_bit_cast(bf16x8, p3);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const u32x4* pb = s_frag + ((e * NT + nt) * 3) * 64 + lane;
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, pb[0]), b2 = __builtin_bit_cast(bf16x8, pb[64]), b3 = __builtin_bit_cast(bf16x8, pb[128]);
#ifdef LN_CONV_PROBE_NO_MFMA  // attribution build (wrong results): the gathers and operand loads stay, ONE matrix instruction per slot and tile
            asm volatile("" ::"v"(a2), "v"(a3), "v"(b2), "v"(b3));  // (operands stay computed and loaded)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[nt], 0, 0, 0);
#else
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[nt], 0, 0, 0);
#endif
        }
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) out[(size_t)row * F + nt * 16 + i] = acc[nt][r];
        }
}

template <bool FLIP, bool WT>
static int ln_conv_dispatch(const int* nbr, const float* values_neigh, const float* filter, int m, int filter_extent, int val_dim,
                            int nr_filters, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
    bool done = false;
    if (filter_extent == 9 && (reinterpret_cast<uintptr_t>(filter) & 15) == 0) {  // d = 3 small-filter fast path
        if constexpr (!FLIP && !WT) {  // V = F = 32 forward on the bf16 matrix cores (LN_DEBUG_MASK & 524288: fp32 form, A/B)
            if (val_dim == 32 && nr_filters == 32 && m >= LN_CONV_B3_MIN_ROWS && ln_conv_b3_enabled() && !(ln_debug_mask() & 524288) &&
                (reinterpret_cast<uintptr_t>(values_neigh) & 15) == 0) {
                int t = min(ln_bwd_subtiles(m), 3);
                {
                    static int forced = -1;  // experiment knob: LN_FWD_T=1..3 forces the forward's sub-tile count alone (read once)
                    if (forced < 0) {
                        const char* e = getenv("LN_FWD_T");
                        forced = e ? atoi(e) : 0;
                    }
                    if (forced >= 1 && forced <= 3) t = forced;
                }
                const dim3 grid_t(ln_div_up(m, 64 * t)), block_t(256 * t);
                if (t == 1) LN_LAUNCH("k_conv_mfma", (k_conv_forward_b3<1>), grid_t, block_t, 0, st, nbr, values_neigh, filter, m, out, g_ln_row_partition);
                else if (t == 2) LN_LAUNCH("k_conv_mfma", (k_conv_forward_b3<2>), grid_t, block_t, 0, st, nbr, values_neigh, filter, m, out, g_ln_row_partition);
                else LN_LAUNCH("k_conv_mfma", (k_conv_forward_b3<3>), grid_t, block_t, 0, st, nbr, values_neigh, filter, m, out, g_ln_row_partition);
                return ln_check_launch("ln_conv_forward");
            }
        }
        const dim3 grid(ln_div_up(m, 64)), block(256);
#define LN_CONV_FULL(VV, NN)                                                                                                        \
    if (!done && val_dim == VV && nr_filters == 16 * NN) {                                                                          \
        LN_LAUNCH("k_conv_mfma", (k_conv_mfma_full<VV, NN, 9, FLIP, WT>), grid, block, 0, st, nbr, values_neigh, filter, m, out,  \
                  (int)grid.x, (const float*)nullptr, 0, 0, (float*)nullptr);                                                       \
        done = true;                                                                                                                \
    }
        LN_CONV_FULL(32, 2) LN_CONV_FULL(32, 1) LN_CONV_FULL(16, 1) LN_CONV_FULL(16, 2) LN_CONV_FULL(16, 4) LN_CONV_FULL(8, 1)
        LN_CONV_FULL(8, 2) LN_CONV_FULL(8, 4) LN_CONV_FULL(8, 8)
#undef LN_CONV_FULL
    }
    if (!done && nr_filters % 16 == 0 && ((reinterpret_cast<uintptr_t>(values_neigh) | reinterpret_cast<uintptr_t>(filter)) & 15) == 0) {
        const int nf = nr_filters;
        switch (val_dim) {
            case 8: done = ln_conv_launch_v<8, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 16: done = ln_conv_launch_v<16, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 32: done = ln_conv_launch_v<32, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 48: done = ln_conv_launch_v<48, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 64: done = ln_conv_launch_v<64, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 96: done = ln_conv_launch_v<96, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 128: done = ln_conv_launch_v<128, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 192: done = ln_conv_launch_v<192, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            case 256: done = ln_conv_launch_v<256, FLIP, WT>(nf, nbr, values_neigh, filter, m, filter_extent, out, ws, ws_bytes, st); break;
            default: break;
        }
    }
    if (!done) {
        const long long work = (long long)m * nr_filters;
        LN_LAUNCH("k_conv_generic", k_conv_generic, dim3(ln_div_up(work, 256)), dim3(256), 0, st, nbr, values_neigh, filter, work,
                  filter_extent, val_dim, nr_filters, FLIP ? 1 : 0, WT ? 1 : 0, out);
    }
    return ln_check_launch("ln_conv_forward");
}

extern "C" int ln_conv_forward_ws(const int* nbr, const float* values_neigh, const float* filter, int m, int filter_extent, int val_dim,
                                  int nr_filters, int flags, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    LN_REQUIRE(m >= 0 && filter_extent >= 1 && val_dim >= 1 && nr_filters >= 1, LN_ERR_ARG, "ln_conv_forward: bad sizes");
    LN_REQUIRE(m == 0 || (nbr && values_neigh && filter && out), LN_ERR_ARG, "ln_conv_forward: null buffer");
    LN_REQUIRE((flags & ~7) == 0, LN_ERR_ARG, "ln_conv_forward: unknown flags %d", flags);
    if (m == 0) return LN_OK;
    hipStream_t st = (hipStream_t)stream;
    void* ws = workspace;
    const size_t wb = workspace_bytes;
    struct BankReady {  // (scoped: the flag never outlives the call)
        explicit BankReady(bool on) { g_ln_bank_ready = on; }
        ~BankReady() { g_ln_bank_ready = false; }
    } bank_ready((flags & LN_CONV_BANK_READY) != 0 && workspace != nullptr);
    flags &= 3;
    switch (flags) {
        case 0: return ln_conv_dispatch<false, false>(nbr, values_neigh, filter, m, filter_extent, val_dim, nr_filters, out, ws, wb, st);
        case LN_CONV_FLIP_NEIGHBOURS: return ln_conv_dispatch<true, false>(nbr, values_neigh, filter, m, filter_extent, val_dim, nr_filters, out, ws, wb, st);
        case LN_CONV_TRANSPOSED_FILTER: return ln_conv_dispatch<false, true>(nbr, values_neigh, filter, m, filter_extent, val_dim, nr_filters, out, ws, wb, st);
        default: return ln_conv_dispatch<true, true>(nbr, values_neigh, filter, m, filter_extent, val_dim, nr_filters, out, ws, wb, st);
    }
}

extern "C" int ln_conv_forward(const int* nbr, const float* values_neigh, const float* filter, int m, int filter_extent,
                               int val_dim, int nr_filters, int flags, float* out, void* stream) {
    return ln_conv_forward_ws(nbr, values_neigh, filter, m, filter_extent, val_dim, nr_filters, flags, out, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------
// filter gradient: grad_filter[e*V+v, f] = sum_m values[nbr[m,e], v] * grad_out[m, f]
// Stage 1: grid (row chunks, E); each wave reduces its rows with MFMA (D[v][f] += A[v][m] B[m][f]),
//          the 4 waves are combined through LDS and the workgroup writes one partial [V,F] slab.
// Stage 2: deterministic sum of the slabs.
// ------------------------------------------------------------------------------------------
#ifndef LN_GF_ROWS
#define LN_GF_ROWS 320   // lattice vertices per workgroup (one slab each)
#endif
#define LN_GF_SUB 64     // vertices staged in LDS at a time (25 KiB of LDS -> 6 workgroups per CU)

// Stage 1.  grid = (row chunks, E).  A workgroup walks its chunk in sub-tiles of LN_GF_SUB vertices: the
// gathered neighbour rows A[sub, V] and the gradient rows G[sub, F] are staged in LDS with float4 loads
// (row stride +16 floats so that the k-strided MFMA operand reads are bank-conflict free), the loads of
// the next sub-tile are issued before the MFMAs of the current one, and each wave owns whole 16x16
// output tiles D[v, f] += sum_rows A[row, v] * G[row, f] with two accumulators per tile.
template <int VT, int FT>
__global__ void __launch_bounds__(256)
    k_grad_filter_mfma(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ grad_out, int m,
                       int E, float* __restrict__ partial, int v_total, int v_off, int f_total, int f_off) {
    // rows [v_off, v_off + 16*VT) x columns [f_off, f_off + 16*FT) of every slot's [v_total, f_total] block; with
    // gridDim.z > 1 the launch covers a uniform tiling of the block: sub-block z = (row tile, column tile)
    if (gridDim.z > 1) {
        const int col_tiles = f_total / (16 * FT);
        v_off = (blockIdx.z / col_tiles) * 16 * VT;
        f_off = (blockIdx.z % col_tiles) * 16 * FT;
    }
    constexpr int V = VT * 16;
    constexpr int F = FT * 16;
    constexpr int SA = V + 16;  // LDS row strides (floats), = 16 mod 32
    constexpr int SG = F + 16;
    constexpr int TILES = VT * FT;
    constexpr int TPW = (TILES + 3) / 4;     // output tiles per wave
    constexpr int A4 = LN_GF_SUB * V / 4 / 256;  // float4 staged per thread
    constexpr int G4 = LN_GF_SUB * F / 4 / 256;
    static_assert(A4 >= 1 && G4 >= 1, "tile too small for 256 staging threads");
    __shared__ __attribute__((aligned(16))) float s_a[LN_GF_SUB * SA];
    __shared__ __attribute__((aligned(16))) float s_g[LN_GF_SUB * SG];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int e = blockIdx.y;
    const int chunk_begin = blockIdx.x * LN_GF_ROWS;
    const int chunk_end = min(chunk_begin + LN_GF_ROWS, m);

    floatx4 acc[TPW][2];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        acc[t][0] = floatx4{0.f, 0.f, 0.f, 0.f};
        acc[t][1] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    float4 ra[A4], rg[G4];
    auto issue_loads = [&](int sub_begin) {
#pragma unroll
        for (int s = 0; s < A4; ++s) {
            const int x4 = tid + s * 256;
            const int r = x4 / (V / 4);
            const int c = x4 - r * (V / 4);
            const int row = sub_begin + r;
            const int nb = (row < chunk_end) ? nbr[(size_t)row * E + e] : -1;
            ra[s] = *reinterpret_cast<const float4*>(values + (size_t)(nb >= 0 ? nb : 0) * v_total + v_off + c * 4);
            if (nb < 0) ra[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s = 0; s < G4; ++s) {
            const int x4 = tid + s * 256;
            const int r = x4 / (F / 4);
            const int c = x4 - r * (F / 4);
            const int row = sub_begin + r;
            rg[s] = (row < chunk_end) ? *reinterpret_cast<const float4*>(grad_out + (size_t)row * f_total + f_off + c * 4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (chunk_begin < chunk_end) issue_loads(chunk_begin);
    for (int sub = chunk_begin; sub < chunk_end; sub += LN_GF_SUB) {
        __syncthreads();  // previous sub-tile fully consumed
#pragma unroll
        for (int s = 0; s < A4; ++s) {
            const int x4 = tid + s * 256;
            const int r = x4 / (V / 4);
            const int c = x4 - r * (V / 4);
            *reinterpret_cast<float4*>(s_a + r * SA + c * 4) = ra[s];
        }
#pragma unroll
        for (int s = 0; s < G4; ++s) {
            const int x4 = tid + s * 256;
            const int r = x4 / (F / 4);
            const int c = x4 - r * (F / 4);
            *reinterpret_cast<float4*>(s_g + r * SG + c * 4) = rg[s];
        }
        __syncthreads();
        if (sub + LN_GF_SUB < chunk_end) issue_loads(sub + LN_GF_SUB);  // in flight during the MFMAs below
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = wave + 4 * t;
            if (tile < TILES) {
                const int vt = tile / FT;
                const int ft = tile - vt * FT;
                const float* pa = s_a + q * SA + vt * 16 + i;
                const float* pg = s_g + q * SG + ft * 16 + i;
#pragma unroll 8
                for (int k = 0; k < LN_GF_SUB / 4; k += 2) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[(k * 4) * SA], pg[(k * 4) * SG], acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[(k * 4 + 4) * SA], pg[(k * 4 + 4) * SG], acc[t][1], 0, 0, 0);
                }
            }
        }
    }
    float* dst = partial + ((size_t)blockIdx.x * E + e) * ((size_t)v_total * f_total) + (size_t)v_off * f_total + f_off;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + 4 * t;
        if (tile < TILES) {
            const int vt = tile / FT;
            const int ft = tile - vt * FT;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)(vt * 16 + q * 4 + r) * f_total + ft * 16 + i] = acc[t][0][r] + acc[t][1][r];
        }
    }
}

// generic fallback: thread per (e*V+v, f), loops all rows (slow; small or odd shapes only)
__global__ void __launch_bounds__(256)
    k_grad_filter_generic(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ grad_out, int m,
                          int E, int V, int F, float* __restrict__ grad_filter) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E * V * F) return;
    const int f = g % F;
    const int ev = g / F;
    const int v = ev % V;
    const int e = ev / V;
    float acc = 0.0f;
    for (int row = 0; row < m; ++row) {
        const int nb = nbr[(size_t)row * E + e];
        if (nb >= 0) acc = fmaf(values[(size_t)nb * V + v], grad_out[(size_t)row * F + f], acc);
    }
    grad_filter[g] = acc;
}

// 16 outputs per workgroup; the slabs are split over 16 thread rows and combined through LDS
__global__ void __launch_bounds__(256) k_reduce_slabs(const float* __restrict__ partial, int nslabs, int total, float* __restrict__ out) {
    __shared__ float s_part[16][17];
    const int o = threadIdx.x & 15;
    const int part = threadIdx.x >> 4;
    const int g = blockIdx.x * 16 + o;
    float acc = 0.0f;
    if (g < total)
        for (int s = part; s < nslabs; s += 16) acc += partial[(size_t)s * total + g];
    s_part[part][o] = acc;
    __syncthreads();
    if (part == 0 && g < total) {
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += s_part[k][o];
        out[g] = r;
    }
}

// total % 64 == 0: 64 outputs per workgroup as 16 float4 columns x 16 slab groups (every thread has its <= ceil(nslabs / 16)
// float4 loads in flight at once), combined through LDS in a fixed order
__global__ void __launch_bounds__(256) k_reduce_slabs4(const float* __restrict__ partial, int nslabs, int total, float* __restrict__ out) {
    ln_reduce_slabs4_block(blockIdx.x, partial, nslabs, total, out);
}

// out[i] = sum over s of partial[s * total + i], in slab order (deterministic); shared with the fp16 filter gradient
int ln_reduce_slabs_async(const float* partial, int nslabs, int total, float* out, hipStream_t st) {
    if (total % 64 == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0)
        LN_LAUNCH("k_reduce_slabs", k_reduce_slabs4, dim3(total / 64), dim3(256), 0, st, partial, nslabs, total, out);
    else
        LN_LAUNCH("k_reduce_slabs", k_reduce_slabs, dim3(ln_div_up(total, 16)), dim3(256), 0, st, partial, nslabs, total, out);
    return LN_OK;
}

// ------------------------------------------------------------------------------------------
// Filter gradient on the bf16 matrix cores with exactly 3-way split operands (the arithmetic of k_conv_mfma_b3: x = hi + mid + lo,
// six of the nine cross products, fp32 accumulation; 1e-5 per element against fp64).
//   grad_filter[e][v, f] = sum_rows values[nbr[row, e], v] * grad_out[row, f]
// The contraction runs over lattice vertices, and the gradient rows G[row, :] are THE SAME for all nine slots: a workgroup owns a
// chunk of rows and a [16 VT, 16 FT] block of every slot's [V, F] matrix, walks its chunk in sub-tiles of 64 rows, splits and
// stages G once per sub-tile and the gathered neighbour rows A_e once per slot (three bf16 planes each, row-major; the matrix
// operands are read "down the rows" with ds_read_b64_tr_b16), and keeps the E x (tiles per wave) accumulators in registers.
// The waves form a 2 x 2 grid over the block: a wave owns VT/2 x FT/2 tiles and reads (VT/2 + FT/2) x 3 fragments for
// VT/2 x FT/2 x 6 matrix instructions per 32-row step.  The next slot's rows (ids -> rows: two dependent round trips) are fetched
// while the matrix instructions of the current one run.  One [E, V, F] slab per row chunk, summed by k_reduce_slabs4.
// (k_grad_filter_mfma, one slot per workgroup on v_mfma_f32_16x16x4_f32, ran at 43-46 % of the fp32 matrix peak: 191 us for
// M = 46.5 k, V = F = 128.)
// ------------------------------------------------------------------------------------------
typedef short gf_short4 __attribute__((ext_vector_type(4)));
#ifndef LN_GFB_SUB
#define LN_GFB_SUB 64
#endif
#ifndef LN_GFB_WAVES
#define LN_GFB_WAVES 2
#endif
#ifndef LN_GFB_PROBE
#define LN_GFB_PROBE 0  // timing ablations (wrong results): 1 no matrix products / fragment reads, 2 no gather, 4 no operand split
#endif
#ifndef LN_GFB_EG
#define LN_GFB_EG 3   // slots per workgroup: E = 9 as three groups (gridDim.y); the gradient rows are split three times instead of nine
#endif
// WV x WF waves, each owning (VT / WV) x (FT / WF) tiles of 16 x 16 per slot.  2 x 2 waves on a 64 x 64 block (two workgroups per CU)
// is the round-4 shape; at 128 x 128 the block is the whole [V, F] face on 4 x 4 waves (one 1024-thread workgroup per CU): every
// gathered row is fetched ONCE for all its channels and used against all filters, the gradient rows once per slot group — with
// 64 x 64 sub-blocks the rows were gathered twice and the gradient rows read six times (571 MB through the CUs' load path for
// 285 MB here; the kernel ran at that rate).
template <int VT, int FT, int E, int WV, int WF>
__global__ void __launch_bounds__(64 * WV * WF, (WV * WF) > 8 ? 1 : LN_GFB_WAVES)
    k_grad_filter_b3(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ grad_out, int m, int rows_per_wg,
                     float* __restrict__ partial, int v_total, int f_total) {
    constexpr int V = VT * 16, F = FT * 16;
    constexpr int EG = LN_GFB_EG;
    static_assert(E % EG == 0, "slot groups");
    constexpr int TPV = VT / WV, TPF = FT / WF;      // tiles of a wave
    constexpr int THREADS = 64 * WV * WF;
    // bf16 elements per staged row: + 32 bytes, i.e. a row stride that is an odd multiple of 8 dwords.  A transposing read is served in
    // two groups of 32 lanes; with the row mapping of `frag` below a group touches 8 CONSECUTIVE rows x 32 bytes, which then fall on 8
    // disjoint sets of 8 banks.  (Round 4's + 16 bytes with rows {8q .. 8q+3} per 16 lanes put rows r and r + 2 on overlapping banks and
    // the two halves of a group on the same ones: SQ_LDS_BANK_CONFLICT was 41 % of the LDS-array cycles, profiles/r5_pmc_lds.json.)
    constexpr int RSA = V + 16, RSG = F + 16;
    constexpr int PA = LN_GFB_SUB * RSA, PG = LN_GFB_SUB * RSG;  // one plane
    constexpr int A4 = LN_GFB_SUB * V / 4 / THREADS, G4 = LN_GFB_SUB * F / 4 / THREADS;  // float4 fetched per thread and sub-tile
    static_assert(A4 >= 1 && G4 >= 1 && VT % WV == 0 && FT % WF == 0 && (LN_GFB_SUB * V / 4) % THREADS == 0 && (LN_GFB_SUB * F / 4) % THREADS == 0,
                  "block shape");
    extern __shared__ __attribute__((aligned(16))) unsigned short s_gfb[];
    unsigned short* s_a = s_gfb;                     // [3][64][RSA]
    unsigned short* s_g = s_gfb + 3 * PA;            // [3][64][RSG]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int col_blocks = f_total / F;
    const int v_off = (blockIdx.z / col_blocks) * V, f_off = (blockIdx.z % col_blocks) * F;
    const int e0 = blockIdx.y * EG;
    const int chunk_begin = blockIdx.x * rows_per_wg;
    const int chunk_end = min(chunk_begin + rows_per_wg, m);
    const int wv = wave / WF, wf = wave % WF;        // the wave's place in the WV x WF grid

    floatx4 acc[EG][TPV][TPF];
#pragma unroll
    for (int e = 0; e < EG; ++e)
#pragma unroll
        for (int a = 0; a < TPV; ++a)
#pragma unroll
            for (int b = 0; b < TPF; ++b) acc[e][a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    float4 ra[A4], rg[G4];
    int ids[EG][A4];  // neighbour ids of the sub-tile's rows for the group's slots: fetched one sub-tile ahead, so that a slot's
                      // row reads depend on registers only (ids -> rows were two dependent round trips per slot: the kernel ran at their latency)
    auto fetch_ids = [&](int sub) {
#pragma unroll
        for (int e = 0; e < EG; ++e)
#pragma unroll
            for (int k = 0; k < A4; ++k) {
                const int row = sub + (tid + THREADS * k) / (V / 4);
                ids[e][k] = row < chunk_end ? nbr[(size_t)row * E + e0 + e] : -1;
            }
    };
    auto fetch_a = [&](const int (&nb)[A4]) {
#pragma unroll
        for (int k = 0; k < A4; ++k) {
            const int x4 = tid + THREADS * k;
            const int c4 = x4 % (V / 4);
            ra[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#if LN_GFB_PROBE & 2
            ra[k] = make_float4(1.f + nb[k], 2.f, 3.f + c4, 4.f);
#else
            if (nb[k] >= 0) ra[k] = *reinterpret_cast<const float4*>(values + (size_t)nb[k] * v_total + v_off + c4 * 4);
#endif
        }
    };
    auto fetch_g = [&](int sub) {
#pragma unroll
        for (int k = 0; k < G4; ++k) {
            const int x4 = tid + THREADS * k;
            const int r = x4 / (F / 4), c4 = x4 - r * (F / 4);
            const int row = sub + r;
            rg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < chunk_end) rg[k] = *reinterpret_cast<const float4*>(grad_out + (size_t)row * f_total + f_off + c4 * 4);
        }
    };
    auto stage = [&](const float4& x, unsigned short* dst, int plane) {  // four consecutive channels of one row -> three planes
        unsigned int h[4], md[4], lo[4];
#if LN_GFB_PROBE & 4
        h[0] = md[0] = lo[0] = __float_as_uint(x.x); h[1] = md[1] = lo[1] = __float_as_uint(x.y);
        h[2] = md[2] = lo[2] = __float_as_uint(x.z); h[3] = md[3] = lo[3] = __float_as_uint(x.w);
#else
        ln_split3_bits(x.x, h[0], md[0], lo[0]);
        ln_split3_bits(x.y, h[1], md[1], lo[1]);
        ln_split3_bits(x.z, h[2], md[2], lo[2]);
        ln_split3_bits(x.w, h[3], md[3], lo[3]);
#endif
        *reinterpret_cast<uint2*>(dst) = make_uint2(ln_pack_hi(h[0], h[1]), ln_pack_hi(h[2], h[3]));
        *reinterpret_cast<uint2*>(dst + plane) = make_uint2(ln_pack_hi(md[0], md[1]), ln_pack_hi(md[2], md[3]));
        *reinterpret_cast<uint2*>(dst + 2 * plane) = make_uint2(ln_pack_hi(lo[0], lo[1]), ln_pack_hi(lo[2], lo[3]));
    };
    auto frag = [&](const unsigned short* base, int rs) {  // 8 bf16 down the rows: rows 4q..4q+3 and 16+4q..16+4q+3 of the 32-row step
        // (the order of the contraction inside a step is free as long as both operands use the same one), one column
        const gf_short4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_short4 __attribute__((address_space(3)))*)(base));
        const gf_short4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_short4 __attribute__((address_space(3)))*)(base + 16 * rs));
        u32x4 p;
        p[0] = (unsigned int)(unsigned short)lo4[0] | ((unsigned int)(unsigned short)lo4[1] << 16);
        p[1] = (unsigned int)(unsigned short)lo4[2] | ((unsigned int)(unsigned short)lo4[3] << 16);
        p[2] = (unsigned int)(unsigned short)hi4[0] | ((unsigned int)(unsigned short)hi4[1] << 16);
        p[3] = (unsigned int)(unsigned short)hi4[2] | ((unsigned int)(unsigned short)hi4[3] << 16);
        return __builtin_bit_cast(bf16x8, p);
    };
    if (chunk_begin < chunk_end) {
        fetch_ids(chunk_begin);
        fetch_g(chunk_begin);
        fetch_a(ids[0]);
    }
    for (int sub = chunk_begin; sub < chunk_end; sub += LN_GFB_SUB) {
        int nxt[A4];  // slot e0's ids of the NEXT sub-tile (the other slots' are fetched when this sub-tile's are used up)
#pragma unroll
        for (int k = 0; k < A4; ++k) {
            const int row = sub + LN_GFB_SUB + (tid + THREADS * k) / (V / 4);
            nxt[k] = row < chunk_end ? nbr[(size_t)row * E + e0] : -1;
        }
#pragma unroll
        for (int e = 0; e < EG; ++e) {
            __syncthreads();  // the matrix instructions of the previous slot are done with s_a (and, at e = 0, with s_g)
            if (e == 0) {
#pragma unroll
                for (int k = 0; k < G4; ++k) {
                    const int x4 = tid + THREADS * k;
                    const int r = x4 / (F / 4), c4 = x4 - r * (F / 4);
                    stage(rg[k], s_g + r * RSG + c4 * 4, PG);
                }
            }
#pragma unroll
            for (int k = 0; k < A4; ++k) {
                const int x4 = tid + THREADS * k;
                const int r = x4 / (V / 4), c4 = x4 - r * (V / 4);
                stage(ra[k], s_a + r * RSA + c4 * 4, PA);
            }
            __syncthreads();
            if (e + 1 < EG) {
                fetch_a(ids[e + 1]);  // in flight during the matrix instructions below
            } else if (sub + LN_GFB_SUB < chunk_end) {
                fetch_g(sub + LN_GFB_SUB);
                fetch_a(nxt);
                fetch_ids(sub + LN_GFB_SUB);
            }
#pragma unroll
            for (int st = 0; st < ((LN_GFB_PROBE & 1) ? 0 : LN_GFB_SUB / 32); ++st) {
                const int row0 = 32 * st + 4 * q + (i >> 2);
                bf16x8 fa[TPV][3], fb[TPF][3];
#pragma unroll
                for (int a = 0; a < TPV; ++a)
#pragma unroll
                    for (int part = 0; part < 3; ++part) fa[a][part] = frag(s_a + part * PA + row0 * RSA + (wv * TPV + a) * 16 + (i & 3) * 4, RSA);
#pragma unroll
                for (int b = 0; b < TPF; ++b)
#pragma unroll
                    for (int part = 0; part < 3; ++part) fb[b][part] = frag(s_g + part * PG + row0 * RSG + (wf * TPF + b) * 16 + (i & 3) * 4, RSG);
#pragma unroll
                for (int a = 0; a < TPV; ++a)
#pragma unroll
                    for (int b = 0; b < TPF; ++b) {
                        floatx4 c = acc[e][a][b];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][2], fb[b][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[b][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][1], fb[b][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][1], fb[b][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[b][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[b][0], c, 0, 0, 0);
                        acc[e][a][b] = c;
                    }
            }
        }
    }
    // D: column = lane & 15 (f), row = q * 4 + reg (v)
#pragma unroll
    for (int e = 0; e < EG; ++e) {
        float* dst = partial + ((size_t)blockIdx.x * E + e0 + e) * ((size_t)v_total * f_total) + (size_t)v_off * f_total + f_off;
#pragma unroll
        for (int a = 0; a < TPV; ++a)
#pragma unroll
            for (int b = 0; b < TPF; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[(size_t)((wv * TPV + a) * 16 + q * 4 + r) * f_total + (wf * TPF + b) * 16 + i] = acc[e][a][b][r];
    }
}

// rows per workgroup of the bf16x3 filter gradient: as few as fill the chip (>= 512 workgroups over chunks x slot groups x sub-blocks) while the
// slabs the chunks write (and k_reduce_slabs4 reads back) stay under LN_GFB_SLAB_BYTES; a multiple of the 64-row sub-tile
#define LN_GFB_SLAB_BYTES (24ll << 20)
// LN_GFB_WIDE=0 keeps the 64 x 64 sub-blocks of round 4 (A/B; read once)
static bool ln_gfb_wide() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LN_GFB_WIDE");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static bool ln_gfb_block(int val_dim, int nr_filters, int* vs, int* fs) {
    // (whole faces of 128 x 64, 64 x 128 and 96 x 96 on 4 x 2 / 2 x 4 / 3 x 2 waves measured the same as their 64 x 64 / 32 x 96
    // sub-blocks — 65 vs 66 us, 71 vs 71 us at 46 k rows —, and 96 x 96 on 2 x 2 waves spills: only 128 x 128 takes the whole face)
    static const int cand[7][2] = {{128, 128}, {64, 64}, {32, 96}, {96, 32}, {64, 32}, {32, 64}, {32, 32}};
    for (auto& c : cand)
        if (val_dim % c[0] == 0 && nr_filters % c[1] == 0 && (ln_gfb_wide() || (c[0] + c[1] <= 128))) {
            *vs = c[0];
            *fs = c[1];
            return true;
        }
    return false;
}
static int ln_gfb_rows(int m, int filter_extent, int val_dim, int nr_filters) {
    int vs = 0, fs = 0;
    if (filter_extent != 9 || !ln_gfb_block(val_dim, nr_filters, &vs, &fs)) return 0;  // (the kernel is instantiated for E = 9: d = 3)
    const long long z = (long long)(val_dim / vs) * (nr_filters / fs) * (filter_extent / LN_GFB_EG);  // workgroups per row chunk
    const long long slab = (long long)filter_extent * val_dim * nr_filters * 4;
    const bool one_per_cu = vs * fs > 64 * 96;                              // blocks on more than eight waves take a whole CU
    long long chunks = ((one_per_cu ? 256 : 512) + z - 1) / z;              // (else two workgroups per CU) ...
    const long long budget = one_per_cu ? 2 * LN_GFB_SLAB_BYTES : LN_GFB_SLAB_BYTES;
    const long long cap = budget / slab > 0 ? budget / slab : 1;
    if (chunks > cap) chunks = cap;                                         // ... unless the slabs would cost more than the products
    long long rows = ((m + chunks - 1) / chunks + LN_GFB_SUB - 1) / LN_GFB_SUB * LN_GFB_SUB;
    if (rows < LN_GFB_SUB) rows = LN_GFB_SUB;
    return int(rows);
}
static bool ln_gfb_enabled(int m, int filter_extent, int val_dim, int nr_filters) {
    int vs, fs;
    return filter_extent == 9 && m >= LN_CONV_B3_MIN_ROWS && ln_conv_b3_enabled() && !(ln_debug_mask() & 8388608) &&
           ln_gfb_block(val_dim, nr_filters, &vs, &fs);
}

// Any multiple of 16 in both dimensions: the [V, F] block of a slot is covered by sub-blocks of {64, 32, 16} x {64, 32, 16}.
static bool ln_gf_mfma_supported(int val_dim, int nr_filters) { return val_dim % 16 == 0 && nr_filters % 16 == 0; }

extern "C" size_t ln_conv_grad_filter_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters) {
    if (!ln_gf_mfma_supported(val_dim, nr_filters) || m <= 0) return 256;
    // one [E, V, F] slab per row chunk; the fused backward of a same-lattice convolution (ln_conv_backward) has its own chunking
    int chunks = ln_bwd_fused_shape(filter_extent, val_dim, nr_filters) ? max(ln_div_up(m, LN_GF_ROWS), ln_bwd_workgroups(m))
                                                                        : ln_div_up(m, LN_GF_ROWS);
    const int rows_b3 = ln_gfb_rows(m, filter_extent, val_dim, nr_filters);  // (whichever of the two forms runs: LN_DEBUG_MASK can switch)
    if (rows_b3 > 0) chunks = max(chunks, ln_div_up(m, rows_b3));
    return (size_t)chunks * filter_extent * val_dim * nr_filters * sizeof(float) + 256;
}

// stage 1 of the MFMA filter gradient: per-row-chunk partial blocks -> slabs [chunk][E*V*F]; returns the chunk count
static int ln_gf_launch_partials(const int* nbr, const float* values_neigh, const float* grad_out, int m, int filter_extent, int val_dim,
                                 int nr_filters, float* partial, hipStream_t st) {
    const dim3 block(256);
    if (ln_gfb_enabled(m, filter_extent, val_dim, nr_filters)) {  // bf16 matrix cores, gradient rows split once for all nine slots
        int vs = 0, fs = 0;
        ln_gfb_block(val_dim, nr_filters, &vs, &fs);
        const int rows = ln_gfb_rows(m, filter_extent, val_dim, nr_filters);
        const int chunks_b3 = ln_div_up(m, rows);
        const dim3 grid(chunks_b3, filter_extent / LN_GFB_EG, (val_dim / vs) * (nr_filters / fs));
        const size_t lds = (size_t)3 * LN_GFB_SUB * ((vs + 16) + (fs + 16)) * sizeof(unsigned short);
#define LN_GFB_CASE(A, B, WVV, WFF)                                                                                                    \
    if (vs == 16 * A && fs == 16 * B) {                                                                                                \
        static bool attr_set = false;                                                                                                  \
        if (!attr_set && lds > 64 * 1024) {                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grad_filter_b3<A, B, 9, WVV, WFF>),                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                           \
            attr_set = true;                                                                                                           \
        }                                                                                                                              \
        LN_LAUNCH("k_grad_filter_mfma", (k_grad_filter_b3<A, B, 9, WVV, WFF>), grid, dim3(64 * WVV * WFF), lds, st, nbr, values_neigh, grad_out, \
                  m, rows, partial, val_dim, nr_filters);                                                                              \
    }
#ifndef LN_GFB_W128F
#define LN_GFB_W128F 2  // waves across the filters of a 128 x 128 block: 4 x 2 waves (512 threads, 248 registers; measured 101 us at 46 k rows
                        // against 118 for 4 x 4 waves, whose 128-register budget spills 13 dwords, and 132 for 64 x 64 sub-blocks)
#endif
        LN_GFB_CASE(8, 8, 4, LN_GFB_W128F) 
        LN_GFB_CASE(4, 4, 2, 2) LN_GFB_CASE(2, 6, 2, 2) LN_GFB_CASE(6, 2, 2, 2) LN_GFB_CASE(4, 2, 2, 2) LN_GFB_CASE(2, 4, 2, 2) LN_GFB_CASE(2, 2, 2, 2)
#undef LN_GFB_CASE
        return chunks_b3;
    }
    const int chunks = ln_div_up(m, LN_GF_ROWS);
    // uniform tiling (both dimensions multiples of the widest tile that divides them): ONE launch, gridDim.z = sub-blocks
    for (int t = 4; t >= 1; t >>= 1) {
        if (val_dim % (16 * t) == 0 && nr_filters % (16 * t) == 0) {
            const dim3 grid(chunks, filter_extent, (val_dim / (16 * t)) * (nr_filters / (16 * t)));
            if (t == 4)
                LN_LAUNCH("k_grad_filter_mfma", (k_grad_filter_mfma<4, 4>), grid, block, 0, st, nbr, values_neigh, grad_out, m, filter_extent, partial,
                          val_dim, 0, nr_filters, 0);
            else if (t == 2)
                LN_LAUNCH("k_grad_filter_mfma", (k_grad_filter_mfma<2, 2>), grid, block, 0, st, nbr, values_neigh, grad_out, m, filter_extent, partial,
                          val_dim, 0, nr_filters, 0);
            else
                LN_LAUNCH("k_grad_filter_mfma", (k_grad_filter_mfma<1, 1>), grid, block, 0, st, nbr, values_neigh, grad_out, m, filter_extent, partial,
                          val_dim, 0, nr_filters, 0);
            return chunks;
        }
    }
    return chunks;  // unreachable: both dimensions are multiples of 16 (ln_gf_mfma_supported)
}

static int ln_conv_grad_filter_impl(const int* nbr, const float* values_neigh, const float* grad_out, int m, int filter_extent, int val_dim,
                                    int nr_filters, float* grad_filter, void* workspace, size_t workspace_bytes, void* stream, bool defer_sum);

extern "C" int ln_conv_grad_filter(const int* nbr, const float* values_neigh, const float* grad_out, int m, int filter_extent,
                                   int val_dim, int nr_filters, float* grad_filter, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    return ln_conv_grad_filter_impl(nbr, values_neigh, grad_out, m, filter_extent, val_dim, nr_filters, grad_filter, workspace, workspace_bytes,
                                    stream, false);
}

// defer_sum: leave the slab sum as g_ln_slab_job for the next bank split of this thread to carry (ln_conv_backward; the caller
// launches it itself if nothing took it)
static int ln_conv_grad_filter_impl(const int* nbr, const float* values_neigh, const float* grad_out, int m, int filter_extent, int val_dim,
                                    int nr_filters, float* grad_filter, void* workspace, size_t workspace_bytes, void* stream, bool defer_sum) {
    LN_REQUIRE(m >= 0 && filter_extent >= 1 && val_dim >= 1 && nr_filters >= 1, LN_ERR_ARG, "ln_conv_grad_filter: bad sizes");
    LN_REQUIRE(grad_filter && (m == 0 || (nbr && values_neigh && grad_out)), LN_ERR_ARG, "ln_conv_grad_filter: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const int total = filter_extent * val_dim * nr_filters;
    if (m == 0) {
        (void)ln_zero_async(grad_filter, (size_t)total * sizeof(float), st);
        return ln_check_launch("ln_conv_grad_filter");
    }
    if (ln_gf_mfma_supported(val_dim, nr_filters)) {
        LN_REQUIRE(workspace && workspace_bytes >= ln_conv_grad_filter_workspace_bytes(m, filter_extent, val_dim, nr_filters),
                   LN_ERR_WORKSPACE, "ln_conv_grad_filter: workspace too small");
        LN_REQUIRE((reinterpret_cast<uintptr_t>(values_neigh) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 15) == 0, LN_ERR_ARG,
                   "ln_conv_grad_filter: values / grad_out must be 16-byte aligned");
        float* partial = static_cast<float*>(workspace);
        const int chunks = ln_gf_launch_partials(nbr, values_neigh, grad_out, m, filter_extent, val_dim, nr_filters, partial, st);
        // slabs are laid out [chunk][e][V*F]; summing over chunks with stride E*V*F
        if (defer_sum && total % 64 == 0 && ((reinterpret_cast<uintptr_t>(grad_filter) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0)
            g_ln_slab_job = LnSlabSum{partial, chunks, total, grad_filter};
        else
            (void)ln_reduce_slabs_async(partial, chunks, total, grad_filter, st);
    } else {
        LN_LAUNCH("k_grad_filter_generic", k_grad_filter_generic, dim3(ln_div_up(total, 256)), dim3(256), 0, st, nbr, values_neigh, grad_out, m,
                           filter_extent, val_dim, nr_filters, grad_filter);
    }
    return ln_check_launch("ln_conv_grad_filter");
}

// ------------------------------------------------------------------------------------------
// Both gradients of a same-lattice small-filter convolution from ONE gather per (vertex, slot).
//   grad_values[n, :] = sum_e G_e[n, :] W_e^T            with G_e[n, :] = grad_out[nbr(n, e^1), :]   (centre: nbr(n, 8) = n)
//   grad_filter[e]    = sum_m values[nbr(m, e), :]^T grad_out[m, :]
//                     = sum_n values[n, :]^T G_e[n, :]   (substitute n = nbr(m, e); neighbour lists of one lattice are symmetric:
//                                                          nbr(m, e) = n  <=>  nbr(n, e^1) = m, both or neither present)
// so the rows the value gradient gathers are exactly the rows the filter gradient needs, against the vertex's OWN value row.
// The separate launches gathered 9 rows per vertex twice (grad_out rows for one, value rows for the other: 37 + 73 MB of
// L2 misses at C3) and staged the bank / the gradient rows twice.
// Workgroup = T 64-vertex sub-tiles of 4 waves each (T = 3 at C3: 192 vertices, one workgroup per CU, 3 waves per SIMD); a
// sub-tile's wave w owns the vertices 16w..16w+15 for the value gradient (A = its gathered quarter rows in registers, B =
// W_e^T fragments in LDS, as k_conv_mfma_full) and the 16x16 tile (w / FT, w % FT) of every slot's filter gradient over the
// sub-tile's 64 vertices (A = its own value rows, transposed, in registers for all slots; B = G_e staged row-major in LDS,
// double-buffered: one barrier per slot).  The sub-tiles' filter-gradient accumulators are added through LDS and
// the workgroup writes one [E, V, F] slab; k_reduce_slabs4 adds the slabs (deterministic order).
// ------------------------------------------------------------------------------------------
template <int V, int F, int E, int T>
__global__ void __launch_bounds__(256 * T) __attribute__((amdgpu_waves_per_eu(T, T)))
    k_conv_backward_fused(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ grad_out,
                          const float* __restrict__ filter, int m, float* __restrict__ grad_values, float* __restrict__ slabs) {
    constexpr int KQ = F / 4;    // contraction channels (forward outputs) per lane quarter
    constexpr int NT = V / 16;   // value-gradient column tiles
    constexpr int VT = V / 16, FT = F / 16;
    static_assert(VT * FT == 4, "one filter-gradient tile per wave of a sub-tile");
    static_assert(E * V * F * 4 <= 64 * 1024, "filter bank must fit 64 KiB of LDS");
    constexpr int SG = F + 16;   // LDS row stride of the staged gradient rows (floats, = 16 mod 32: conflict-free k-strided reads)
    constexpr int THREADS = 256 * T;
    constexpr int BANK = E * V * F;
    constexpr int STAGE = 2 * 64 * SG;  // floats per sub-tile: G_e double-buffered
    // one array: [W_e^T fragments ((e*KQ + kk)*NT + nt)*64 + lane][G_e of each sub-tile]; reused at the end to add up the
    // sub-tiles' filter gradients (T - 1 parked copies)
    constexpr int LDS_FLOATS = (BANK + T * STAGE) > (T - 1) * BANK ? (BANK + T * STAGE) : (T - 1) * BANK;
    __shared__ __attribute__((aligned(16))) float s_all[LDS_FLOATS];
    float* s_b = s_all;
    const int tid = threadIdx.x;
    const int sub = tid >> 8;
    const int t256 = tid & 255;
    const int lane = tid & 63;
    const int wave = t256 >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int sub0 = blockIdx.x * (64 * T) + sub * 64;  // first vertex of this sub-tile
    const int m0 = sub0 + wave * 16;
    const int my_row = m0 + i;

    LN_CSTAMP(0);
    // loads in dependency order: bank (depends on nothing), neighbour ids, own value rows, the E gathers
    constexpr int N4 = E * V * F / 4;
    constexpr int NST = (N4 + THREADS - 1) / THREADS;
    float4 wv[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int x4 = tid + s * THREADS;
        wv[s] = (x4 < N4) ? reinterpret_cast<const float4*>(filter)[x4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int nb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) nb[e] = (my_row < m) ? nbr[(size_t)my_row * E + (e < E - 1 ? (e ^ 1) : e)] : -1;
    // A operand of the filter gradient: values[sub0 + 4*step + q][vt*16 + i] for the 16 steps over the sub-tile's rows
    const int vt = wave / FT, ft = wave % FT;
    float vT[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        const int row = sub0 + 4 * st + q;
        vT[st] = (row < m) ? values[(size_t)row * V + vt * 16 + i] : 0.f;
    }
    // The gathers run DEPTH slots ahead of the matrix cores (loads return in order: issuing all E up front, as k_conv_mfma_full
    // does, makes slot 0 wait behind the whole 53 MB burst of every wave on the chip — with one workgroup per CU nothing else
    // would cover that wait)
    // The centre slot goes FIRST: its row is the vertex's own (no neighbour id to wait for), so its gather leaves with the
    // id loads and the first barrier is one round trip away instead of two; the id only masks it (rows beyond the build).
    constexpr int DEPTH = 3;
    auto slot_of = [](int k) { return k == 0 ? E - 1 : k - 1; };  // processing order -> filter slot
    float a[DEPTH + 1][KQ];
    ln_load_quarter<KQ>(grad_out + (size_t)(my_row < m ? my_row : 0) * F + q * KQ, a[0]);
#pragma unroll
    for (int k = 1; k < DEPTH && k < E; ++k)
        ln_load_quarter<KQ>(grad_out + (size_t)(nb[slot_of(k)] >= 0 ? nb[slot_of(k)] : 0) * F + q * KQ, a[k % (DEPTH + 1)]);
    // bank -> LDS as W_e^T fragments (the bank is [e][v][f]: contraction index f, output index v)
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int x4 = tid + s * THREADS;
        if (x4 < N4) {
            const int x = x4 * 4;
            const int ev = x / F;       // e*V + v
            const int k0 = x - ev * F;  // f, multiple of 4
            const int e = ev / V;
            const int v = ev - e * V;
            const float vals4[4] = {wv[s].x, wv[s].y, wv[s].z, wv[s].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j;
                const int qq = k / KQ;
                const int kk = k - qq * KQ;
                s_b[((e * KQ + kk) * NT + (v >> 4)) * 64 + qq * 16 + (v & 15)] = vals4[j];
            }
        }
    }
    floatx4 acc_v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc_v[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 acc_w[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc_w[e] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int e = slot_of(k);
        float* sg = s_all + BANK + sub * STAGE + (k & 1) * (64 * SG);
        float (&ae)[KQ] = a[k % (DEPTH + 1)];
        if (k + DEPTH < E)
            ln_load_quarter<KQ>(grad_out + (size_t)(nb[slot_of(k + DEPTH)] >= 0 ? nb[slot_of(k + DEPTH)] : 0) * F + q * KQ, a[(k + DEPTH) % (DEPTH + 1)]);
        // this lane's quarter of G_e[row i of its wave] -> LDS (zeros for absent neighbours)
        {
            float* dst = sg + (wave * 16 + i) * SG + q * KQ;
#pragma unroll
            for (int c = 0; c < KQ; c += 4)
                *reinterpret_cast<float4*>(dst + c) = nb[e] >= 0 ? make_float4(ae[c], ae[c + 1], ae[c + 2], ae[c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();  // G_e staged (and, the first time, the bank); buffer (k+1)&1 was last read before the previous barrier
        if (k == 0) LN_CSTAMP(1);
        if (k == 1) LN_CSTAMP(4);
        if (k == 8) LN_CSTAMP(5);
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const float av = nb[e] >= 0 ? ae[kk] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s_b[((e * KQ + kk) * NT + nt) * 64 + lane], acc_v[nt], 0, 0, 0);
        }
        const float* pg = sg + q * SG + ft * 16 + i;
        floatx4 w0 = floatx4{0.f, 0.f, 0.f, 0.f}, w1 = floatx4{0.f, 0.f, 0.f, 0.f};  // two chains: no back-to-back dependent MFMAs
#pragma unroll
        for (int st = 0; st < 16; st += 2) {
            w0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vT[st], pg[(4 * st) * SG], w0, 0, 0, 0);
            w1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vT[st + 1], pg[(4 * st + 4) * SG], w1, 0, 0, 0);
        }
        acc_w[e] = w0 + w1;
    }
    LN_CSTAMP(2);
    // value gradient: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) grad_values[(size_t)row * V + nt * 16 + i] = acc_v[nt][r];
        }
    }
    // filter gradient: sub-tiles 1.. park their tiles in LDS, sub-tile 0 adds them (fixed order) and writes the workgroup's slab
    __syncthreads();
    if (sub > 0) {
        float* park = s_all + (size_t)(sub - 1) * BANK;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) park[(e * V + vt * 16 + q * 4 + r) * F + ft * 16 + i] = acc_w[e][r];
    }
    if constexpr (T > 1) __syncthreads();
    if (sub == 0) {
        float* dst = slabs + (size_t)blockIdx.x * BANK;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (e * V + vt * 16 + q * 4 + r) * F + ft * 16 + i;
                float sum = acc_w[e][r];
#pragma unroll
                for (int k = 1; k < T; ++k) sum += s_all[(size_t)(k - 1) * BANK + o];
                dst[o] = sum;
            }
    }
    LN_CSTAMP(3);
}

// ------------------------------------------------------------------------------------------
// The fused backward on the bf16 matrix cores with exactly 3-way split operands (see k_conv_mfma_b3 for the arithmetic).
// V = F = 32 only: one v_mfma_f32_16x16x32_bf16 step contracts all 32 channels (value gradient) or 32 vertices (filter gradient).
//   value gradient : A = this lane's gathered quarter row (channels 8q..8q+7 = its k-group), split in registers; B = W_e^T split
//                    while the bank is staged — LDS holds one 16-byte fragment per (slot, column tile, part, lane), 54 KB
//   filter gradient: D[v][f] += sum_rows values[row][v] G_e[row][f].  A = own value rows, transposed (lane (v, q): rows 8q..8q+7 of
//                    each 32-row step), loaded and split once per sub-tile; B = G_e, whose three parts every lane stages row-major
//                    as bf16 — the SAME parts it feeds the value gradient with — and the column access the MFMA needs (lane (f, q):
//                    rows 8q..8q+7 of column f) comes from ds_read_b64_tr_b16, the LDS transpose read (tools/probes/tr_read_probe.cpp)
// Per slot and wave 12 + 12 MFMAs of 16 cycles instead of 16 + 16 of 32: backward 31.0 -> 24.5 us at C3.
// T = 1..3 (four sub-tiles of staging do not fit LDS; LN_DEBUG_MASK & 65536 selects the fp32 kernel for A/B).
// History worth keeping: for a while only T = 3 was launched, because with T = 1 / 2 — workgroups that leave room on their CU for
// waves of other kernels — two concurrently replayed scans returned wrong filter gradients in 1-3 % of the replays.  The kernel was
// never wrong (a -DLN_TR_CHECK build re-reads every transposed fragment element by element): its INPUT was, reduced by the other
// scan's k_csr_reduce_segments while this kernel was on the chip.  Cause, isolated in tools/probes/pk_fma_vs_mfma_probe.cpp: on this
// part a packed fp32 instruction whose LOW result takes a source from the HIGH half of a register pair (v_pk_fma_f32 op_sel:[0,1,0],
// v_pk_mul_f32 op_sel:[0,1] — the compiler's weight broadcasts in the reduce) returns a wrong low result while a wave of another kernel
// on the same SIMD executes v_mfma_f32_16x16x32_bf16 / _f16.  The library is compiled without packed fp32 instructions
// (build_ext.py, tests/test_device_isa.py); tools/probes/pair_probe.py: 288 of 300 reduces wrong beside the T = 1 form before,
// 0 of 300 after; tests/test_gpu_parity.py::test_other_streams_unharmed_beside_fused_backward keeps the pair under test.
// ------------------------------------------------------------------------------------------
typedef short short4v __attribute__((ext_vector_type(4)));
#ifdef LN_TR_CHECK  // investigation build (tools/probes/fused_b3_stress.py): transposed fragments re-read element by element
__device__ int ln_dbg[4 + 16 * 8];
extern "C" int ln_debug_dump(int* host, int n) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ln_dbg), n * sizeof(int)) != hipSuccess) return 1;
    static int zeros[4 + 16 * 8];
    return hipMemcpyToSymbol(HIP_SYMBOL(ln_dbg), zeros, sizeof zeros) != hipSuccess;
}
#endif
template <int T>
__global__ void __launch_bounds__(256 * T) __attribute__((amdgpu_waves_per_eu(T, T)))
    k_conv_backward_fused_b3(const int* __restrict__ nbr, const float* __restrict__ values, const float* __restrict__ grad_out,
                             const float* __restrict__ filter, int m, float* __restrict__ grad_values, float* __restrict__ slabs,
                             const int* __restrict__ row_part) {
    constexpr int V = 32, F = 32, E = 9, KQ = 8, NT = 2, FT = 2;
    constexpr int THREADS = 256 * T;
    constexpr int BANK = E * V * F;                      // floats of one slab (the parking area at the end)
    constexpr int FRAG16 = E * NT * 3 * 64;              // 16-byte fragments of the split bank
#ifndef LN_BWD_SWZ
#define LN_BWD_SWZ 1  // 0: the padded 80-byte rows of rounds 3-5 (A/B)
#endif
#ifndef LN_BWD_LINE
#define LN_BWD_LINE LN_BWD_SWZ  // line-shaped gathers of the gradient rows (below); 0: the fragment-shaped ones of rounds 3-5 (A/B)
#endif
    // Staged G_e rows.  Rounds 3-5: 64 bytes + 16 of padding per row — conflict-free for the 16-byte staging stores but not for the
    // transposing reads (rows r and r + 3 / r + 8 and r + 11 of a lane group overlap).  Round 6: no padding, the 16-byte piece p of row r
    // sits at position p ^ swz(r), swz(r) = ((r >> 1) & 3) ^ (bit 3 of r) << 1, chosen against the lane groups of MI355X_MICROARCH.md §LDS:
    //   ds_write_b128 — 8 adjacent lanes (rows 8a .. 8a + 7 of one quarter) over 128 bytes: even rows take positions q ^ {0, 1, 2, 3} of
    //                   the first 64 bytes, odd rows of the second — eight different 16-byte columns;
    //   ds_read_b64_tr_b16 — 32 lanes (rows R .. R + 3 and R + 8 .. R + 11, 32 bytes each) over 256 bytes: rows R + j and R + 8 + j share
    //                   a 64-byte column and, bit 3 flipping the half, read its two different halves.
    // 12 KB less LDS per sub-tile.
    constexpr int RS = LN_BWD_SWZ ? 32 : 40;             // bf16 elements per staged G_e row
#if LN_BWD_LINE
    auto swz = [](int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; };  // {0, 2, 3, 1} by row quad (see the line-shaped gathers below)
#else
    auto swz = [](int r) { return ((r >> 1) & 3) ^ ((r >> 2) & 2); };
#endif
    constexpr int PART = 64 * RS;                        // one part of one sub-tile's G_e
    constexpr int STAGE = 2 * 3 * PART;                  // double-buffered, three parts (bf16 elements)
    constexpr int LDS_BYTES_A = FRAG16 * 16 + T * STAGE * 2;
    constexpr int LDS_BYTES_B = (T - 1) * BANK * 4;
    constexpr int LDS_BYTES = LDS_BYTES_A > LDS_BYTES_B ? LDS_BYTES_A : LDS_BYTES_B;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LDS_BYTES];
    u32x4* s_frag = reinterpret_cast<u32x4*>(s_raw);
    unsigned short* s_fh = reinterpret_cast<unsigned short*>(s_raw);
    unsigned short* s_stage = reinterpret_cast<unsigned short*>(s_raw + FRAG16 * 16);
    const int tid = threadIdx.x;
    const int sub = tid >> 8;
    const int t256 = tid & 255;
    const int lane = tid & 63;
    const int wave = t256 >> 6;
    const int i = lane & 15;
    const int q = lane >> 4;
    const int bx = ln_partition_tile(blockIdx.x, gridDim.x, row_part, 64 * T);  // space-ordered table: XCD x takes the tiles of kd region x
    const int sub0 = bx * (64 * T) + sub * 64;
    const int m0 = sub0 + wave * 16;
    const int my_row = m0 + i;

    constexpr int N4 = E * V * F / 4;
    constexpr int NST = (N4 + THREADS - 1) / THREADS;
    float4 wv[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int x4 = tid + s * THREADS;
        wv[s] = (N4 % THREADS == 0 || x4 < N4) ? reinterpret_cast<const float4*>(filter)[x4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#if !LN_BWD_LINE
    int nb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) nb[e] = (my_row < m) ? nbr[(size_t)my_row * E + (e < E - 1 ? (e ^ 1) : e)] : -1;
#else
    static_assert(LN_BWD_SWZ, "the line-shaped gathers stage unpadded, swizzled rows");
#endif
    const int vt = wave / FT, ft = wave % FT;
    // A operand of the filter gradient, split: step s (32 rows), part p -> 8 bf16 = rows 32s + 8q + j of column vt*16 + i
    u32x4 va[2][3];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = sub0 + 32 * st + 8 * q + j;  // (unconditional load of a clamped row + select: no branch around sixteen loads)
            const float xv = values[(size_t)min(row, m - 1) * V + vt * 16 + i];
            x[j] = (row < m) ? xv : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int h0, m0_, l0, h1, m1_, l1;
            ln_split3_bits(x[2 * j], h0, m0_, l0);
            ln_split3_bits(x[2 * j + 1], h1, m1_, l1);
            va[st][0][j] = ln_pack_hi(h0, h1);
            va[st][1][j] = ln_pack_hi(m0_, m1_);
            va[st][2][j] = ln_pack_hi(l0, l1);
        }
    }
#ifndef LN_BWD_DEPTH
#define LN_BWD_DEPTH 2  // (round 6, line-shaped gathers: 1 / 2 / 3 / 4 / 5 -> 19.0 / 18.7 / 19.3 / 20.4 / 20.8 us on one box)
#endif
    constexpr int DEPTH = LN_BWD_DEPTH;  // gathers in flight per lane
    auto slot_of = [](int k) { return k == 0 ? E - 1 : k - 1; };
#if LN_BWD_LINE
    // Line-shaped gathers (round 6; the forward has them since round 5).  A load shaped like the MFMA fragment — lane (i, q) reads its
    // 32-byte quarter of row i — touches 16 lines per wave-instruction, half of each; here lane l loads piece l & 7 (16 bytes) of the rows
    // l >> 3 and 8 + (l >> 3) of the wave's 16: 8 whole lines per instruction.  The rows are staged for the filter gradient anyway; the lane
    // splits ITS piece, stores 8 bytes per plane, and takes the A operand of the value gradient — its (i, q) quarter, all three parts —
    // back from the staged rows behind the slot's barrier.  The swizzle {0, 2, 3, 1} by row quad serves all three access shapes:
    //   ds_write_b64   (16 adjacent lanes = two whole rows, 64 bytes each, on the two halves of the 32 banks): any per-row permutation;
    //   ds_read_b128   (lane groups {0-3, 12-15, 20-27} ...: quarter q of rows i, i + 12 and quarter q ^ 1 of rows i + 4, i + 8 share
    //                   the 64-byte column i & 3): positions s(0), s(3), 1 ^ s(1), 1 ^ s(2) = 0, 1, 3, 2;
    //   ds_read_b64_tr (rows R + j and R + 8 + j share a column and read one 32-byte half each): bit 1 of s differs between quads h, h ^ 2.
    const int lr = lane >> 3, c8 = lane & 7;
    int nb2[2][E];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) {  // (clamped row + select, as above)
            const int id = nbr[(size_t)min(m0 + 8 * j + lr, m - 1) * E + (e < E - 1 ? (e ^ 1) : e)];
            nb2[j][e] = (m0 + 8 * j + lr < m) ? id : -1;
        }
    floatx4 a[DEPTH + 1][2];
    auto gather = [&](int e, floatx4 (&dst)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) dst[j] = *reinterpret_cast<const floatx4*>(grad_out + (size_t)(nb2[j][e] >= 0 ? nb2[j][e] : 0) * F + c8 * 4);
    };
#pragma unroll
    for (int j = 0; j < 2; ++j)  // the centre slot is the row itself: no id to wait for
        a[0][j] = *reinterpret_cast<const floatx4*>(grad_out + (size_t)(m0 + 8 * j + lr < m ? m0 + 8 * j + lr : 0) * F + c8 * 4);
#pragma unroll
    for (int k = 1; k < DEPTH && k < E; ++k) gather(slot_of(k), a[k % (DEPTH + 1)]);
#else
    float a[DEPTH + 1][KQ];
    ln_load_quarter<KQ>(grad_out + (size_t)(my_row < m ? my_row : 0) * F + q * KQ, a[0]);
#pragma unroll
    for (int k = 1; k < DEPTH && k < E; ++k)
        ln_load_quarter<KQ>(grad_out + (size_t)(nb[slot_of(k)] >= 0 ? nb[slot_of(k)] : 0) * F + q * KQ, a[k % (DEPTH + 1)]);
#endif
    // bank -> split -> LDS fragments of W_e^T: x = (e*V + v)*F + f, four consecutive f = elements j..j+3 of ONE fragment
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int x4 = tid + s * THREADS;
        if (N4 % THREADS == 0 || x4 < N4) {
            const int x = x4 * 4;
            const int ev = x / F;
            const int f0 = x - ev * F;
            const int e = ev / V;
            const int v = ev - e * V;
            const float v4[4] = {wv[s].x, wv[s].y, wv[s].z, wv[s].w};
            unsigned int h[4], md[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ln_split3_bits(v4[j], h[j], md[j], lo[j]);
            // (the 16-byte unit of (q = f0 >> 3, i = v & 15) sits at column i ^ 2q of its 16-unit row.  ds_write_b64 is served in groups
            // of 16 adjacent lanes over 32 banks: here two v times (four q, two halves), whose columns i ^ 2q are distinct modulo 8, so a
            // group covers the 128 bytes once; unswizzled the four q of one v sat 256 bytes apart on the same banks (4-way).  The
            // fragment reads below — ds_read_b128, lane groups {0-3,12-15,20-27} ... over 64 banks — stay conflict-free under this XOR:
            // masks 0/2 keep a lane's column inside its quarter pair, masks 4/6 swap both halves of a group together.)
            const int fr = (((e * NT + (v >> 4)) * 3) * 64 + (f0 >> 3) * 16 + ((v & 15) ^ (2 * (f0 >> 3)))) * 8 + (f0 & 7);
            *reinterpret_cast<uint2*>(s_fh + fr) = make_uint2(ln_pack_hi(h[0], h[1]), ln_pack_hi(h[2], h[3]));
            *reinterpret_cast<uint2*>(s_fh + fr + 64 * 8) = make_uint2(ln_pack_hi(md[0], md[1]), ln_pack_hi(md[2], md[3]));
            *reinterpret_cast<uint2*>(s_fh + fr + 2 * 64 * 8) = make_uint2(ln_pack_hi(lo[0], lo[1]), ln_pack_hi(lo[2], lo[3]));
        }
    }
    floatx4 acc_v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc_v[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 acc_w[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc_w[e] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int e = slot_of(k);
        unsigned short* sg = s_stage + (size_t)sub * STAGE + (k & 1) * (3 * PART);
#if LN_BWD_LINE
        if (k + DEPTH < E) gather(slot_of(k + DEPTH), a[(k + DEPTH) % (DEPTH + 1)]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // this lane's piece of rows lr and 8 + lr: split once, staged for both gradients
            const floatx4 x = a[k % (DEPTH + 1)][j];
            const bool there = nb2[j][e] >= 0;
            unsigned int h[4], md[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) ln_split3_bits(there ? x[c] : 0.f, h[c], md[c], lo[c]);
            const int row = wave * 16 + 8 * j + lr;
            unsigned short* dst = sg + row * RS + (((c8 >> 1) ^ swz(row)) * 8 + (c8 & 1) * 4);
            *reinterpret_cast<uint2*>(dst) = make_uint2(ln_pack_hi(h[0], h[1]), ln_pack_hi(h[2], h[3]));
            *reinterpret_cast<uint2*>(dst + PART) = make_uint2(ln_pack_hi(md[0], md[1]), ln_pack_hi(md[2], md[3]));
            *reinterpret_cast<uint2*>(dst + 2 * PART) = make_uint2(ln_pack_hi(lo[0], lo[1]), ln_pack_hi(lo[2], lo[3]));
        }
        __syncthreads();
        u32x4 p1, p2, p3;
        {
            const unsigned short* src = sg + (wave * 16 + i) * RS + ((q ^ swz(wave * 16 + i)) * KQ);
            p1 = *reinterpret_cast<const u32x4*>(src);
            p2 = *reinterpret_cast<const u32x4*>(src + PART);
            p3 = *reinterpret_cast<const u32x4*>(src + 2 * PART);
        }
#else
        float (&ae)[KQ] = a[k % (DEPTH + 1)];
        if (k + DEPTH < E)
            ln_load_quarter<KQ>(grad_out + (size_t)(nb[slot_of(k + DEPTH)] >= 0 ? nb[slot_of(k + DEPTH)] : 0) * F + q * KQ, a[(k + DEPTH) % (DEPTH + 1)]);
        // this lane's quarter of G_e[row], split once: the A fragments of the value gradient AND what is staged for the filter gradient
        u32x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int h0, m0_, l0, h1, m1_, l1;
            ln_split3_bits(nb[e] >= 0 ? ae[2 * j] : 0.f, h0, m0_, l0);
            ln_split3_bits(nb[e] >= 0 ? ae[2 * j + 1] : 0.f, h1, m1_, l1);
            p1[j] = ln_pack_hi(h0, h1);
            p2[j] = ln_pack_hi(m0_, m1_);
            p3[j] = ln_pack_hi(l0, l1);
        }
        {
            unsigned short* dst = sg + (wave * 16 + i) * RS + (LN_BWD_SWZ ? ((q ^ swz(wave * 16 + i)) * KQ) : q * KQ);
            *reinterpret_cast<u32x4*>(dst) = p1;
            *reinterpret_cast<u32x4*>(dst + PART) = p2;
            *reinterpret_cast<u32x4*>(dst + 2 * PART) = p3;
        }
        __syncthreads();
#ifdef LN_TR_CHECK
        {
            const u32x4 rb = *reinterpret_cast<volatile u32x4*>(sg + (wave * 16 + i) * RS + (LN_BWD_SWZ ? ((q ^ swz(wave * 16 + i)) * KQ) : q * KQ));
            if (rb[0] != p1[0] || rb[1] != p1[1] || rb[2] != p1[2] || rb[3] != p1[3]) atomicAdd(&ln_dbg[1], 1);
        }
#endif
#endif
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, p1), a2 = __builtin_bit_cast(bf16x8, p2), a3 = __builtin_bit_cast(bf16x8, p3);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const u32x4* pb = s_frag + ((e * NT + nt) * 3) * 64 + q * 16 + (i ^ (2 * q));
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, pb[0]), b2 = __builtin_bit_cast(bf16x8, pb[64]), b3 = __builtin_bit_cast(bf16x8, pb[128]);
#ifdef LN_CONV_PROBE_NO_MFMA  // attribution build (wrong results): one matrix instruction instead of six
            asm volatile("" ::"v"(a2), "v"(a3), "v"(b2), "v"(b3));
            acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc_v[nt], 0, 0, 0);
#else
            acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc_v[nt], 0, 0, 0);
            acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc_v[nt], 0, 0, 0);
            acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc_v[nt], 0, 0, 0);
            acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc_v[nt], 0, 0, 0);
            acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc_v[nt], 0, 0, 0);
            acc_v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc_v[nt], 0, 0, 0);
#endif
        }
        // filter gradient: two 32-row steps; B fragments through the LDS transpose read (lane: 4 contiguous bf16 of row
        // row0 + (i >> 2), columns ft*16 + 4 (i & 3) ..; it receives rows row0..row0+3 of column ft*16 + i)
        floatx4 w0 = floatx4{0.f, 0.f, 0.f, 0.f}, w1 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 b[3];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                // lane: 4 contiguous bf16 (half a 16-byte piece) of row r0 + (i >> 2), elements ft * 16 + 4 (i & 3) ..
                const int r_lo = 32 * st + 8 * q + (i >> 2), r_hi = r_lo + 4;
                const int piece = ft * 2 + ((i & 3) >> 1), inner = (i & 1) * 4;
                const unsigned short* base = sg + part * PART + r_lo * RS + (LN_BWD_SWZ ? ((piece ^ swz(r_lo)) * 8 + inner) : (ft * 16 + (i & 3) * 4));
                const unsigned short* base_hi = sg + part * PART + r_hi * RS + (LN_BWD_SWZ ? ((piece ^ swz(r_hi)) * 8 + inner) : (ft * 16 + (i & 3) * 4));
                const short4v lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(base));
                const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(base_hi));
                u32x4 packed;
                packed[0] = (unsigned int)(unsigned short)lo4[0] | ((unsigned int)(unsigned short)lo4[1] << 16);
                packed[1] = (unsigned int)(unsigned short)lo4[2] | ((unsigned int)(unsigned short)lo4[3] << 16);
                packed[2] = (unsigned int)(unsigned short)hi4[0] | ((unsigned int)(unsigned short)hi4[1] << 16);
                packed[3] = (unsigned int)(unsigned short)hi4[2] | ((unsigned int)(unsigned short)hi4[3] << 16);
                b[part] = __builtin_bit_cast(bf16x8, packed);
#ifdef LN_TR_CHECK
                {
                    unsigned int badm = 0, ex2 = 0, got2 = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int rr_ = 32 * st + 8 * q + j, ee_ = ft * 16 + i;
                        const unsigned int ex = *(volatile unsigned short*)(sg + part * PART + rr_ * RS + (LN_BWD_SWZ ? ((((ee_ >> 3) ^ swz(rr_)) << 3) | (ee_ & 7)) : ee_));
                        const unsigned int got = (packed[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                        if (ex != got) { badm |= 1u << j; ex2 = ex; got2 = got; }
                    }
                    if (badm) {
                        const int slot = atomicAdd(&ln_dbg[0], 1);
                        if (slot < 16) {
                            int* r = ln_dbg + 4 + slot * 8;
                            r[0] = blockIdx.x;
                            r[1] = lane | wave << 8 | k << 16 | part << 24 | st << 28;
                            r[2] = (int)badm;
                            r[3] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 6);   // HW_REG_LDS_ALLOC
                            r[4] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
                            r[5] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC id
                            r[6] = (int)(ex2 | got2 << 16);
                            r[7] = (int)(size_t)(sg - reinterpret_cast<unsigned short*>(s_raw));
                        }
                    }
                }
#endif
            }
            const bf16x8 v1 = __builtin_bit_cast(bf16x8, va[st][0]), v2 = __builtin_bit_cast(bf16x8, va[st][1]), v3 = __builtin_bit_cast(bf16x8, va[st][2]);
            floatx4& acc = st ? w1 : w0;
#ifdef LN_CONV_PROBE_NO_MFMA
            asm volatile("" ::"v"(v2), "v"(v3), "v"(b[1]), "v"(b[2]));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, b[0], acc, 0, 0, 0);
#else
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v3, b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v2, b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v2, b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, b[0], acc, 0, 0, 0);
#endif
        }
        acc_w[e] = w0 + w1;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + q * 4 + r;
            if (row < m) grad_values[(size_t)row * V + nt * 16 + i] = acc_v[nt][r];
        }
    }
    __syncthreads();
    float* park_all = reinterpret_cast<float*>(s_raw);
    const int colp = (ft * 16 + i) ^ ((q & 1) << 4);  // parked column (rows are multiples of 32 floats: the swizzle only ever touches the column)
    if (sub > 0) {
        float* park = park_all + (size_t)(sub - 1) * BANK;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) park[(e * V + vt * 16 + q * 4 + r) * F + colp] = acc_w[e][r];  // (lanes of
        // quarters 0/1 and 2/3 are served together: their rows, 4 apart, would share the 16 banks of the column tile)
    }
    if constexpr (T > 1) __syncthreads();
    if (sub == 0) {
        float* dst = slabs + (size_t)blockIdx.x * BANK;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (e * V + vt * 16 + q * 4 + r) * F + ft * 16 + i;
                float sum = acc_w[e][r];
#pragma unroll
                for (int k2 = 1; k2 < T; ++k2) sum += park_all[(k2 - 1) * BANK + (e * V + vt * 16 + q * 4 + r) * F + colp];
                dst[o] = sum;
            }
    }
}

static bool ln_bwd_fused_enabled() { return !(ln_debug_mask() & 2048); }  // LN_DEBUG_MASK & 2048: the two-launch backward (A/B)

// Both gradients of out = conv(values_neigh; nbr_q, filter[E*V, F]):
//   grad_filter[E*V, F] = im2row(values_neigh; nbr_q)^T @ grad_out            (ln_conv_grad_filter)
//   grad_values[mn, V]  = conv(grad_out; nbr_n, filter, FLIP | TRANSPOSED)     (ln_conv_forward)
// For the small-filter shapes (whole bank in LDS) the slab sum of the filter gradient rides in the convolution launch.
extern "C" int ln_conv_backward(const int* nbr_q, const int* nbr_n, const float* values_neigh, const float* grad_out, const float* filter, int mq,
                                int mn, int filter_extent, int val_dim, int nr_filters, float* grad_values, float* grad_filter,
                                void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    bool fused = false;
    const int V = nr_filters, F = val_dim;  // roles in the value-gradient convolution: V channels in, F channels out
    if (filter_extent == 9 && mq > 0 && mn > 0 && ln_gf_mfma_supported(val_dim, nr_filters) && nbr_q && nbr_n && values_neigh && grad_out &&
        filter && grad_values && grad_filter && workspace &&
        workspace_bytes >= ln_conv_grad_filter_workspace_bytes(mq, filter_extent, val_dim, nr_filters) &&
        ((reinterpret_cast<uintptr_t>(values_neigh) | reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(filter)) & 15) == 0) {
        float* partial = static_cast<float*>(workspace);
        const int total = filter_extent * val_dim * nr_filters;
        const int bwd_t = ln_bwd_subtiles(mn);
        if (nbr_q == nbr_n && mq == mn && ln_bwd_fused_shape(filter_extent, val_dim, nr_filters) && ln_bwd_fused_enabled()) {
            // same lattice on both sides: one gather per (vertex, slot) serves both gradients
            const int wgs = ln_div_up(mn, 64 * bwd_t);
#define LN_BWD_FUSED(TT)                                                                                                               \
    case TT:                                                                                                                           \
        if constexpr (TT <= LN_BWD_B3_MAX_T) { /* bf16 matrix cores, exactly split operands */                                         \
            if (ln_conv_b3_enabled() && !(ln_debug_mask() & 65536)) {                                                                  \
                LN_LAUNCH("k_conv_backward_fused", (k_conv_backward_fused_b3<TT>), dim3(wgs), dim3(256 * TT), 0, st, nbr_n, values_neigh, grad_out, \
                          filter, mn, grad_values, partial, g_ln_row_partition);                                                       \
                break;                                                                                                                 \
            }                                                                                                                          \
        }                                                                                                                              \
        if (true)                                                                                                                         \
            LN_LAUNCH("k_conv_backward_fused", (k_conv_backward_fused<32, 32, 9, TT>), dim3(wgs), dim3(256 * TT), 0, st, nbr_n, values_neigh, grad_out, \
                      filter, mn, grad_values, partial);                                                                               \
        break;
            switch (bwd_t) { LN_BWD_FUSED(1) LN_BWD_FUSED(2) LN_BWD_FUSED(3) LN_BWD_FUSED(4) }
#undef LN_BWD_FUSED
            (void)ln_reduce_slabs_async(partial, wgs, total, grad_filter, st);
            return ln_check_launch("ln_conv_backward");
        }
        const int conv_blocks = ln_div_up(mn, 64);
        const dim3 grid(conv_blocks + ln_div_up(total, 16)), block(256);
#define LN_BWD_FULL(VV, NN)                                                                                                            \
    if (!fused && V == VV && F == 16 * NN) {                                                                                           \
        const int chunks = ln_gf_launch_partials(nbr_q, values_neigh, grad_out, mq, filter_extent, val_dim, nr_filters, partial, st);  \
        LN_LAUNCH("k_conv_mfma", (k_conv_mfma_full<VV, NN, 9, true, true>), grid, block, 0, st, nbr_n, grad_out, filter, mn, grad_values, \
                  conv_blocks, (const float*)partial, chunks, total, grad_filter);                                                      \
        fused = true;                                                                                                                  \
    }
        LN_BWD_FULL(32, 2) LN_BWD_FULL(32, 1) LN_BWD_FULL(16, 1) LN_BWD_FULL(16, 2) LN_BWD_FULL(16, 4) LN_BWD_FULL(8, 1) LN_BWD_FULL(8, 2)
        LN_BWD_FULL(8, 4) LN_BWD_FULL(8, 8)
#undef LN_BWD_FULL
    }
    if (fused) return ln_check_launch("ln_conv_backward");
    static int fuse_sum = -1;  // LN_BWD_SLAB_SUM_IN_SPLIT=0: the slab sum as a launch of its own (A/B; read once)
    if (fuse_sum < 0) {
        const char* ev = getenv("LN_BWD_SLAB_SUM_IN_SPLIT");
        fuse_sum = (ev && ev[0] == '0') ? 0 : 1;
    }
    int rc = ln_conv_grad_filter_impl(nbr_q, values_neigh, grad_out, mq, filter_extent, val_dim, nr_filters, grad_filter, workspace, workspace_bytes,
                                      stream, fuse_sum == 1);
    if (rc) {
        (void)ln_take_slab_job();
        return rc;
    }
    // the value-gradient convolution may split over the filter slots: its partial slabs go behind the filter gradient's
    size_t gf_bytes = (ln_conv_grad_filter_workspace_bytes(mq, filter_extent, val_dim, nr_filters) + 255) & ~size_t(255);
    char* conv_ws = (workspace && workspace_bytes > gf_bytes) ? static_cast<char*>(workspace) + gf_bytes : nullptr;
    rc = ln_conv_forward_ws(nbr_n, grad_out, filter, mn, filter_extent, nr_filters, val_dim, LN_CONV_FLIP_NEIGHBOURS | LN_CONV_TRANSPOSED_FILTER,
                            grad_values, conv_ws, conv_ws ? workspace_bytes - gf_bytes : 0, stream);
    const LnSlabSum left = ln_take_slab_job();  // no bank split in that convolution (fp32 form, small filter): the sum as a launch of its own
    if (left.partial) (void)ln_reduce_slabs_async(left.partial, left.nslabs, left.total, left.out, st);
    return rc ? rc : ln_check_launch("ln_conv_backward");
}

// Both gradients of a per-row linear layer y = x w^T (w [cout, cin]; a 1 x 1 lattice convolution over the identity neighbour list
// `ident` [rows, 1]) in one call: grad_w [cout, cin] = the filter gradient with grad_y as the gathered rows and x as the gradient rows,
// grad_x = grad_y w as a plain-bank convolution — in that order, so that the slab sum of the first rides in the bank split of the
// second (as in ln_conv_backward).  grad_x may be NULL.  workspace: ln_linear_backward_workspace_bytes.
extern "C" size_t ln_linear_backward_workspace_bytes(int rows, int cin, int cout) {
    const size_t gf = (ln_conv_grad_filter_workspace_bytes(rows, 1, cout, cin) + 255) & ~size_t(255);
    return gf + ln_conv_forward_workspace_bytes(rows, 1, cout, cin) + 256;
}

extern "C" int ln_linear_backward(const int* ident, const float* x, const float* grad_y, const float* w, int rows, int cin, int cout,
                                  float* grad_x, float* grad_w, void* workspace, size_t workspace_bytes, void* stream) {
    LN_REQUIRE(rows >= 0 && cin >= 1 && cout >= 1 && grad_w, LN_ERR_ARG, "ln_linear_backward: bad sizes / null output");
    LN_REQUIRE(workspace && workspace_bytes >= ln_linear_backward_workspace_bytes(rows, cin, cout), LN_ERR_WORKSPACE,
               "ln_linear_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t gf_bytes = (ln_conv_grad_filter_workspace_bytes(rows, 1, cout, cin) + 255) & ~size_t(255);
    int rc = ln_conv_grad_filter_impl(ident, grad_y, x, rows, 1, cout, cin, grad_w, workspace, gf_bytes, stream, grad_x != nullptr);
    if (rc) {
        (void)ln_take_slab_job();
        return rc;
    }
    if (grad_x && rows > 0) {
        char* conv_ws = static_cast<char*>(workspace) + gf_bytes;
        rc = ln_conv_forward_ws(ident, grad_y, w, rows, 1, cout, cin, 0, grad_x, conv_ws, workspace_bytes - gf_bytes, stream);
    }
    const LnSlabSum left = ln_take_slab_job();
    if (left.partial) (void)ln_reduce_slabs_async(left.partial, left.nslabs, left.total, left.out, st);
    return rc ? rc : ln_check_launch("ln_linear_backward");
}
