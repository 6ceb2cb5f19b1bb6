// Row gather / scatter kernels: splat accumulate, slice, gather, im2row family, fused
// slice+classify.  All of them move rows of V floats between point space [N, ·] and vertex space
// [M, V]; lanes are mapped along the channel dimension (float4 chunks when V % 4 == 0) so that a
// row is read or written as one contiguous segment — the reference maps one thread per point and
// loops channels, which strides every access by the row length (LatticeGPU.cuh:937-971 etc.).
//
// Compiled with -ffp-contract=off: products and sums are rounded separately so the
// order-deterministic kernels are bit-identical to the CPU oracle.
#include "ln_common.h"

// wave-tiled slice + classifier kernels for the common shapes (ln_classify.hip); return 0 when the shape is not covered
int ln_sc_forward_wave(const float* values, const float* delta_w, const float* lin_w, const float* lin_b, const int* idx, const float* w,
                       int n, int pos_dim, int val_dim, int nr_classes, float* logits, hipStream_t st);
int ln_sc_backward_wave(const float* grad_logits, const float* values, const float* delta_w, const float* lin_w, const int* idx,
                        const float* w, int n, int pos_dim, int val_dim, int nr_classes, float* g_delta_w, float* grad_sliced,
                        float* w_eff, float* slabs, int* grid_out, hipStream_t st);

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    using type = float;
};
template <>
struct VecT<4> {
    using type = float4;
};

__device__ __forceinline__ float ln_zero(float) { return 0.0f; }
__device__ __forceinline__ float4 ln_zero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float ln_mul(float a, float s) { return a * s; }
__device__ __forceinline__ float4 ln_mul(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float ln_add(float a, float b) { return a + b; }
__device__ __forceinline__ float4 ln_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ void ln_atomic_add(float* dst, float v) {
    __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ln_atomic_add(float* dst, float4 v) {
    ln_atomic_add(dst + 0, v.x);
    ln_atomic_add(dst + 1, v.y);
    ln_atomic_add(dst + 2, v.z);
    ln_atomic_add(dst + 3, v.w);
}

#define LN_DISPATCH_VEC(val_dim, ...)                  \
    if (((val_dim) & 3) == 0) {                        \
        constexpr int VEC = 4;                         \
        __VA_ARGS__;                                   \
    } else {                                           \
        constexpr int VEC = 1;                         \
        __VA_ARGS__;                                   \
    }

static int ln_check_rows(const char* who, const void* a, const void* b, const void* c, int n, int pos_dim, int val_dim) {
    LN_REQUIRE(n >= 0 && pos_dim >= 1 && pos_dim <= LN_MAX_POS_DIM && val_dim >= 1, LN_ERR_ARG, "%s: bad sizes n=%d d=%d V=%d",
               who, n, pos_dim, val_dim);
    LN_REQUIRE(n == 0 || (a && b && c), LN_ERR_ARG, "%s: null buffer", who);
    return LN_OK;
}

// ------------------------------------------------------------------------------------------
// scatter-add of per-point rows onto vertex rows: splat accumulate and slice backward
// dst[idx[p,r], :] += src[p, :] * w[p,r]           (LatticeGPU.cuh:937-971 and 3574-3613)
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256)
    k_scatter_point_rows(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ idx,
                         const float* __restrict__ w, long long work, int dp1, int chunks) {
    using T = typename VecT<VEC>::type;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long p = g / chunks;
    const int c = int(g - p * chunks);
    const T x = reinterpret_cast<const T*>(src)[g];
    const int V = chunks * VEC;
    for (int r = 0; r < dp1; ++r) {
        const int row = idx[p * dp1 + r];
        if (row >= 0) ln_atomic_add(dst + (size_t)row * V + c * VEC, ln_mul(x, w[p * dp1 + r]));
    }
}

static int ln_scatter_point_rows(const char* who, float* dst, const float* src, const int* idx, const float* w, int n,
                                 int pos_dim, int val_dim, void* stream) {
    int rc = ln_check_rows(who, dst, src, idx, n, pos_dim, val_dim);
    if (rc) return rc;
    LN_REQUIRE(n == 0 || w, LN_ERR_ARG, "%s: null weights", who);
    if (n == 0) return LN_OK;
    LN_DISPATCH_VEC(val_dim, {
        const int chunks = val_dim / VEC;
        const long long work = (long long)n * chunks;
        LN_LAUNCH("k_scatter_point_rows", k_scatter_point_rows<VEC>, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, idx,
                           w, work, pos_dim + 1, chunks);
    });
    return ln_check_launch(who);
}

extern "C" int ln_splat_accumulate(float* table_values, const float* vals, const int* idx, const float* w, int n, int pos_dim,
                                   int val_dim, void* stream) {
    return ln_scatter_point_rows("ln_splat_accumulate", table_values, vals, idx, w, n, pos_dim, val_dim, stream);
}

extern "C" int ln_slice_backward(const float* grad_sliced, const int* idx, const float* w, int n, int pos_dim, int val_dim,
                                 float* grad_values, void* stream) {
    return ln_scatter_point_rows("ln_slice_backward", grad_values, grad_sliced, idx, w, n, pos_dim, val_dim, stream);
}

// ------------------------------------------------------------------------------------------
// slice forward (LatticeGPU.cuh:2567-2591): out[p,:] = sum_r values[idx_r,:] * w_r, r ascending
// ------------------------------------------------------------------------------------------
// DP1 = d + 1 at compile time: the d+1 (index, weight) pairs of a point are fetched in one round trip and its d+1 row gathers
// in a second one (with a runtime trip count the compiler kept the loop rolled: d+1 dependent {index -> row} pairs).
// CH = 8: eight chunks per row as a compile-time constant (32 fp32 channels on float4 lanes: the headline chain) — the point / chunk split
// of the thread index is a shift instead of a division by a run-time value (~20 instructions per thread); 0: `chunks` from the argument.
template <int VEC, int DP1, int CH = 0>
__global__ void __launch_bounds__(256)
    k_slice_forward(const float* __restrict__ values, const int* __restrict__ idx, const float* __restrict__ w, long long work,
                    int chunks_arg, float* __restrict__ out, float* __restrict__ zero_fill, long long zero_elems) {
    const int chunks = CH ? CH : chunks_arg;
    using T = typename VecT<VEC>::type;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#ifdef LN_PROBE_NO_CLEAR  // timing probe (wrong results): no zero fill
    zero_fill = nullptr;
#endif
    if (zero_fill) {  // the accumulator the backward pass of this slice will scatter into, zeroed on the way
        const long long threads = (long long)gridDim.x * blockDim.x;
        for (long long i = g; i < zero_elems; i += threads) zero_fill[i] = 0.f;
    }
    if (g >= work) return;
    const long long p = g / chunks;
    const int c = int(g - p * chunks);
    int rows[DP1];
    float wt[DP1];
#pragma unroll
    for (int r = 0; r < DP1; ++r) {
        rows[r] = idx[p * DP1 + r];
        wt[r] = w[p * DP1 + r];
    }
    T v[DP1];
#pragma unroll
    for (int r = 0; r < DP1; ++r) v[r] = reinterpret_cast<const T*>(values)[(size_t)(rows[r] >= 0 ? rows[r] : 0) * chunks + c];
    T acc = ln_zero(T());
#pragma unroll
    for (int r = 0; r < DP1; ++r)
        if (rows[r] >= 0) acc = ln_add(acc, ln_mul(v[r], wt[r]));  // same order and the same skips as LatticeGPU.cuh:2567-2591
    reinterpret_cast<T*>(out)[g] = acc;
}

#define LN_SLICE_CASE(DD)                                                                                                             \
    case DD:                                                                                                                          \
        if (VEC == 4 && chunks == 8)                                                                                                  \
            LN_LAUNCH("k_slice_forward", (k_slice_forward<4, DD + 1, 8>), dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream,  \
                      values, idx, w, work, chunks, out, zero_fill, zero_elems);                                                      \
        else                                                                                                                          \
        LN_LAUNCH("k_slice_forward", (k_slice_forward<VEC, DD + 1>), dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream, values, \
                  idx, w, work, chunks, out, zero_fill, zero_elems);                                                                  \
        break;
static int ln_slice_forward_impl(const float* values, const int* idx, const float* w, int n, int pos_dim, int val_dim, float* out,
                                 float* zero_fill, long long zero_elems, void* stream) {
    int rc = ln_check_rows("ln_slice_forward", values, idx, out, n, pos_dim, val_dim);
    if (rc) return rc;
    LN_REQUIRE(n == 0 || w, LN_ERR_ARG, "ln_slice_forward: null weights");
    if (n == 0) {
        if (zero_fill && zero_elems > 0) (void)ln_zero_async(zero_fill, sizeof(float) * zero_elems, (hipStream_t)stream);
        return LN_OK;
    }
    LN_DISPATCH_VEC(val_dim, {
        const int chunks = val_dim / VEC;
        const long long work = (long long)n * chunks;
        switch (pos_dim) { LN_SLICE_CASE(1) LN_SLICE_CASE(2) LN_SLICE_CASE(3) LN_SLICE_CASE(4) LN_SLICE_CASE(5) LN_SLICE_CASE(6) }

    });
    return ln_check_launch("ln_slice_forward");
}

extern "C" int ln_slice_forward(const float* values, const int* idx, const float* w, int n, int pos_dim, int val_dim, float* out,
                                void* stream) {
    return ln_slice_forward_impl(values, idx, w, n, pos_dim, val_dim, out, nullptr, 0, stream);
}

extern "C" int ln_slice_forward_prepare_backward(const float* values, const int* idx, const float* w, int n, int pos_dim, int val_dim,
                                                 float* out, float* grad_accumulator, long long grad_accumulator_elems, void* stream) {
    return ln_slice_forward_impl(values, idx, w, n, pos_dim, val_dim, out, grad_accumulator, grad_accumulator_elems, stream);
}

// ------------------------------------------------------------------------------------------
// The same slice with the points taken in the order of the build's slot CSR instead of input order (round 6).
// k_slice_forward walks the points as the caller stored them: on a shuffled cloud every workgroup touches value rows all over the
// table, every XCD pulls the whole [M, V] table through its 4 MB L2 (C3: 23 MB fetched for 5.9 MB of rows).  The CSR of a bucketed
// build lists the tokens bucket by bucket, and over a space-ordered table (LnTable.planes) bucket by bucket means region by region:
// here a point is sliced by whoever meets its remainder-0 token in that list, the d+1 rows it reads are rows of the same region (its
// simplex), and XCD x walks the x-th eighth of the list — a value row is then fetched by ~1.2 L2s instead of 3.3.
// One workgroup takes 256 consecutive CSR entries (one coalesced load), compacts the owners (token % (d+1) == 0, ~64 of them) into
// an LDS list with ballots, and slices them `chunks` lanes per point, two points per lane group in flight; per point the arithmetic
// is that of k_slice_forward (same order, same skips: bit-identical rows).  (First version: one WAVE per 256 entries walking its ~64
// points eight at a time — 1900 waves of eight dependent rounds: 28.6 us against 9.7 us for k_slice_forward.)  Points whose remainder-0 token never reached the CSR (key out of the packable range, a build that
// overflowed) exist only when the table's status word is non-zero: then — and only then — every workgroup also scans its share of
// idx[p * (d+1)] < 0 and slices those points the old way (a point met on both paths is written twice with the same row).
template <int VEC, int DP1>
__global__ void __launch_bounds__(256)
    k_slice_forward_ordered(const float* __restrict__ values, const int* __restrict__ idx, const float* __restrict__ w, int n, int chunks,
                            const int* __restrict__ csr_tok, const int* __restrict__ csr_len, const int* __restrict__ status,
                            float* __restrict__ out, float* __restrict__ zero_fill, long long zero_elems) {
    using T = typename VecT<VEC>::type;
    __shared__ int s_list[256];
    __shared__ int s_wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const long long g = (long long)blockIdx.x * blockDim.x + tid;
        if (zero_fill) {  // the accumulator the backward pass of this slice will scatter into, zeroed on the way
            const long long threads = (long long)gridDim.x * blockDim.x;
            for (long long i = g; i < zero_elems; i += threads) zero_fill[i] = 0.f;
        }
    }
    const int lpp = chunks;      // lanes per point (a power of two <= 64: checked by the host)
    const int G = 256 / lpp;     // points the workgroup slices side by side
    const int grp = tid / lpp, c = tid - grp * lpp;
    // two points per lane group and trip: both sets of d+1 (index, weight) pairs in one round trip, both sets of d+1 row gathers in the next
    auto slice_two = [&](int p0, int p1) {
        int rows[2][DP1];
        float wt[2][DP1];
#pragma unroll
        for (int r = 0; r < DP1; ++r) {
            rows[0][r] = idx[(size_t)p0 * DP1 + r];
            wt[0][r] = w[(size_t)p0 * DP1 + r];
            rows[1][r] = p1 >= 0 ? idx[(size_t)p1 * DP1 + r] : -1;
            wt[1][r] = p1 >= 0 ? w[(size_t)p1 * DP1 + r] : 0.f;
        }
        T v[2][DP1];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int r = 0; r < DP1; ++r) v[k][r] = reinterpret_cast<const T*>(values)[(size_t)(rows[k][r] >= 0 ? rows[k][r] : 0) * chunks + c];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            T acc = ln_zero(T());
#pragma unroll
            for (int r = 0; r < DP1; ++r)
                if (rows[k][r] >= 0) acc = ln_add(acc, ln_mul(v[k][r], wt[k][r]));  // same order and the same skips as LatticeGPU.cuh:2567-2591
            const int p = k ? p1 : p0;
            if (p >= 0) reinterpret_cast<T*>(out)[(size_t)p * chunks + c] = acc;
        }
    };
    const int len = *csr_len;  // entries of the CSR (tokens the build placed)
    const int tile = ln_xcd_chunk_tile(blockIdx.x, gridDim.x);  // XCD x walks the x-th eighth of the list
    for (long long base = (long long)tile * 256; base < len; base += (long long)gridDim.x * 256) {  // (one trip: the grid covers the list)
        const long long e = base + tid;
        const int tk = e < len ? csr_tok[e] : -1;
        const bool own = tk >= 0 && (tk % DP1) == 0 && tk / DP1 < n;
        const unsigned long long mask = __ballot(own);
        if (lane == 0) s_wcnt[wave] = __popcll(mask);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cw = s_wcnt[k];
            total += cw;
            if (k < wave) before += cw;
        }
        if (own) s_list[before + __popcll(mask & ((1ull << lane) - 1ull))] = tk / DP1;
        __syncthreads();
        for (int j = grp; j < total; j += 2 * G) slice_two(s_list[j], j + G < total ? s_list[j + G] : -1);
        __syncthreads();  // (the list and the counts are rewritten by the next trip)
    }
    if (*status != 0) {  // some token of the build never reached the CSR: find the points nobody has met
        const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
        const int p0 = blockIdx.x * per, p1 = min(n, p0 + per);
        for (int pb = p0 + wave * 64; pb < p1; pb += 256) {
            const int p = pb + lane;
            const bool orphan = p < p1 && idx[(size_t)p * DP1] < 0;
            unsigned long long mask = __ballot(orphan);
            while (mask) {  // wave-uniform
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if ((lane / lpp) == 0) {
                    // (lanes 0 .. lpp-1 of the wave take the point: c = lane for them only when the wave is a whole number of groups,
                    // which lpp <= 64 guarantees)
                    const int cc = lane;
                    int rows[DP1];
                    float wt[DP1];
#pragma unroll
                    for (int r = 0; r < DP1; ++r) {
                        rows[r] = idx[(size_t)(pb + src) * DP1 + r];
                        wt[r] = w[(size_t)(pb + src) * DP1 + r];
                    }
                    T acc = ln_zero(T());
#pragma unroll
                    for (int r = 0; r < DP1; ++r)
                        if (rows[r] >= 0) acc = ln_add(acc, ln_mul(reinterpret_cast<const T*>(values)[(size_t)rows[r] * chunks + cc], wt[r]));
                    reinterpret_cast<T*>(out)[(size_t)(pb + src) * chunks + cc] = acc;
                }
            }
        }
    }
}

#define LN_SLICE_ORD_CASE(DD)                                                                                                          \
    case DD:                                                                                                                           \
        LN_LAUNCH("k_slice_forward", (k_slice_forward_ordered<VEC, DD + 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, values, idx, w, n, \
                  chunks, csr->csr_tok, csr->grp_start + t->capacity, t->status, out, grad_accumulator, grad_accumulator_elems);       \
        break;
extern "C" int ln_slice_forward_ordered(const LnTable* t, const LnCsr* csr, const float* values, const int* idx, const float* w, int n,
                                        int val_dim, float* out, float* grad_accumulator, long long grad_accumulator_elems, void* stream) {
    LN_REQUIRE(t && csr && csr->csr_tok && csr->grp_start && t->status && t->capacity > 0, LN_ERR_ARG, "ln_slice_forward_ordered: null table / CSR");
    const int pos_dim = t->pos_dim;
    int rc = ln_check_rows("ln_slice_forward_ordered", values, idx, out, n, pos_dim, val_dim);
    if (rc) return rc;
    LN_REQUIRE(n == 0 || w, LN_ERR_ARG, "ln_slice_forward_ordered: null weights");
    const int chunks4 = val_dim / 4;
    // rows of 4 .. 256 floats whose float4 count is a power of two; everything else goes the plain way
    if (n == 0 || (val_dim & 3) != 0 || chunks4 > 64 || (chunks4 & (chunks4 - 1)) != 0)
        return ln_slice_forward_impl(values, idx, w, n, pos_dim, val_dim, out, grad_accumulator, grad_accumulator_elems, stream);
    {
        constexpr int VEC = 4;
        const int chunks = chunks4;
        const long long tokens = (long long)n * (pos_dim + 1);
        // one workgroup per 256 CSR entries (~64 owners: two trips of the 32 lane groups at 32 channels)
        int grid = ln_div_up(tokens, 256);
        if (grid > 65536) grid = 65536;
        if (grid < 1) grid = 1;
        switch (pos_dim) { LN_SLICE_ORD_CASE(1) LN_SLICE_ORD_CASE(2) LN_SLICE_ORD_CASE(3) LN_SLICE_ORD_CASE(4) LN_SLICE_ORD_CASE(5) LN_SLICE_ORD_CASE(6) }
    }
    return ln_check_launch("ln_slice_forward_ordered");
}
#undef LN_SLICE_ORD_CASE

// fp16 lattice values -> fp16 sliced rows (fp32 arithmetic): thread = (point, HV channels), HV = 8 (16-byte words) when the width
// allows.  As in k_slice_forward: all indices and weights first, then the d+1 row gathers together (the first version fetched
// index -> row -> weight vertex after vertex: four dependent round trips, 52 us at C5 against 25 us now), same summation order;
// `zero_fill`: the fp32 accumulator of this slice's backward pass, zeroed on the way.
template <int DP1, int HV, int CH = 0>  // CH = 8: eight chunks per row at compile time (64 fp16 channels), as in k_slice_forward
__global__ void __launch_bounds__(256)
    k_slice_forward_f16(const _Float16* __restrict__ values, const int* __restrict__ idx, const float* __restrict__ w, long long work,
                        int chunks_arg, _Float16* __restrict__ out, float* __restrict__ zero_fill, long long zero_elems) {
    const int chunks = CH ? CH : chunks_arg;
    typedef _Float16 hv __attribute__((ext_vector_type(HV)));
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_fill) {
        const long long threads = (long long)gridDim.x * blockDim.x;
        for (long long i = g; i < zero_elems; i += threads) zero_fill[i] = 0.f;
    }
    if (g >= work) return;
    const long long p = g / chunks;
    const int c = int(g - p * chunks);
    int rows[DP1];
    float wt[DP1];
#pragma unroll
    for (int r = 0; r < DP1; ++r) {
        rows[r] = idx[p * DP1 + r];
        wt[r] = w[p * DP1 + r];
    }
    hv v[DP1];
#pragma unroll
    for (int r = 0; r < DP1; ++r) v[r] = reinterpret_cast<const hv*>(values)[(size_t)(rows[r] >= 0 ? rows[r] : 0) * chunks + c];
    float acc[HV];
#pragma unroll
    for (int k = 0; k < HV; ++k) acc[k] = 0.f;
#pragma unroll
    for (int r = 0; r < DP1; ++r)
        if (rows[r] >= 0) {
#pragma unroll
            for (int k = 0; k < HV; ++k) acc[k] += (float)v[r][k] * wt[r];
        }
    hv o;
#pragma unroll
    for (int k = 0; k < HV; ++k) o[k] = (_Float16)acc[k];
    reinterpret_cast<hv*>(out)[g] = o;
}

static int ln_slice_forward_f16_impl(const void* values_f16, const int* idx, const float* w, int n, int pos_dim, int val_dim, void* out_f16,
                                     float* zero_fill, long long zero_elems, void* stream) {
    LN_REQUIRE(n >= 0 && pos_dim >= 1 && pos_dim <= LN_MAX_POS_DIM && val_dim >= 4 && val_dim % 4 == 0, LN_ERR_UNSUPPORTED,
               "ln_slice_forward_f16: val_dim must be a multiple of 4 (got %d)", val_dim);
    LN_REQUIRE(n == 0 || (values_f16 && idx && w && out_f16), LN_ERR_ARG, "ln_slice_forward_f16: null buffer");
    LN_REQUIRE(((reinterpret_cast<uintptr_t>(values_f16) | reinterpret_cast<uintptr_t>(out_f16)) & 7) == 0, LN_ERR_ARG,
               "ln_slice_forward_f16: buffers must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (zero_fill && zero_elems > 0) (void)ln_zero_async(zero_fill, sizeof(float) * zero_elems, st);
        return LN_OK;
    }
    const bool wide = val_dim % 8 == 0 && ((reinterpret_cast<uintptr_t>(values_f16) | reinterpret_cast<uintptr_t>(out_f16)) & 15) == 0;
    const int chunks = val_dim / (wide ? 8 : 4);
    const long long work = (long long)n * chunks;
    const _Float16* vv = static_cast<const _Float16*>(values_f16);
    _Float16* oo = static_cast<_Float16*>(out_f16);
#define LN_SLICE16_CASE(DD)                                                                                                            \
    case DD:                                                                                                                          \
        if (wide && chunks == 8)                                                                                                      \
            LN_LAUNCH("k_slice_forward_f16", (k_slice_forward_f16<DD + 1, 8, 8>), dim3(ln_div_up(work, 256)), dim3(256), 0, st, vv, idx, w, work, \
                      chunks, oo, zero_fill, zero_elems);                                                                             \
        else if (wide)                                                                                                                \
            LN_LAUNCH("k_slice_forward_f16", (k_slice_forward_f16<DD + 1, 8>), dim3(ln_div_up(work, 256)), dim3(256), 0, st, vv, idx, w, work, \
                      chunks, oo, zero_fill, zero_elems);                                                                             \
        else                                                                                                                          \
            LN_LAUNCH("k_slice_forward_f16", (k_slice_forward_f16<DD + 1, 4>), dim3(ln_div_up(work, 256)), dim3(256), 0, st, vv, idx, w, work, \
                      chunks, oo, zero_fill, zero_elems);                                                                             \
        break;
    switch (pos_dim) { LN_SLICE16_CASE(1) LN_SLICE16_CASE(2) LN_SLICE16_CASE(3) LN_SLICE16_CASE(4) LN_SLICE16_CASE(5) LN_SLICE16_CASE(6) }
#undef LN_SLICE16_CASE
    return ln_check_launch("ln_slice_forward_f16");
}

extern "C" int ln_slice_forward_f16(const void* values_f16, const int* idx, const float* w, int n, int pos_dim, int val_dim, void* out_f16,
                                    void* stream) {
    return ln_slice_forward_f16_impl(values_f16, idx, w, n, pos_dim, val_dim, out_f16, nullptr, 0, stream);
}

extern "C" int ln_slice_forward_f16_prepare_backward(const void* values_f16, const int* idx, const float* w, int n, int pos_dim, int val_dim,
                                                     void* out_f16, float* grad_accumulator, long long grad_accumulator_elems, void* stream) {
    return ln_slice_forward_f16_impl(values_f16, idx, w, n, pos_dim, val_dim, out_f16, grad_accumulator, grad_accumulator_elems, stream);
}

int ln_retrieve_points(const LnTable* t, const float* positions_raw, const float* sigmas_host, int n, int* idx, float* w,
                       void* stream);

extern "C" int ln_slice_no_precomputation(const LnTable* t, const float* values, const float* positions_raw,
                                          const float* sigmas_host, int n, int val_dim, float* out, int* idx, float* w,
                                          void* stream) {
    LN_REQUIRE(t && (n == 0 || (values && positions_raw && out && idx && w)), LN_ERR_ARG,
               "ln_slice_no_precomputation: null buffer");
    int rc = ln_retrieve_points(t, positions_raw, sigmas_host, n, idx, w, stream);
    if (rc) return rc;
    // found vertices carry w = barycentric[remainder]; absent ones are skipped (LatticeGPU.cuh:2739-2745)
    return ln_slice_forward(values, idx, w, n, t->pos_dim, val_dim, out, stream);
}

// ------------------------------------------------------------------------------------------
// gather (LatticeGPU.cuh:2901-2925) and its backward (LatticeGPU.cuh:3778-3814)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_gather_forward(const float* __restrict__ values, const int* __restrict__ idx, const float* __restrict__ w, long long work,
                     int V, float* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long pr = g / (V + 1);  // p*(d+1)+r
    const int j = int(g - pr * (V + 1));
    const int row = idx[pr];
    float o = 0.0f;  // absent vertices stay 0 (Lattice.cu:899)
    if (row >= 0) {
        const float wt = w[pr];
        o = (j < V) ? values[(size_t)row * V + j] * wt : wt;
    }
    out[g] = o;
}

extern "C" int ln_gather_forward(const float* values, const int* idx, const float* w, int n, int pos_dim, int val_dim, float* out,
                                 void* stream) {
    int rc = ln_check_rows("ln_gather_forward", values, idx, out, n, pos_dim, val_dim);
    if (rc) return rc;
    if (n == 0) return LN_OK;
    const long long work = (long long)n * (pos_dim + 1) * (val_dim + 1);
    LN_LAUNCH("k_gather_forward", k_gather_forward, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream, values, idx, w, work, val_dim,
                       out);
    return ln_check_launch("ln_gather_forward");
}

__global__ void __launch_bounds__(256)
    k_gather_backward(const float* __restrict__ grad, const int* __restrict__ idx, const float* __restrict__ w, long long work,
                      int V, float* __restrict__ grad_values) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long pr = g / (V + 1);
    const int j = int(g - pr * (V + 1));
    if (j >= V) return;  // the weight column's gradient is dropped
    const int row = idx[pr];
    if (row >= 0) ln_atomic_add(grad_values + (size_t)row * V + j, grad[g] * w[pr]);
}

extern "C" int ln_gather_backward(const float* grad_gathered, const int* idx, const float* w, int n, int pos_dim, int val_dim,
                                  float* grad_values, void* stream) {
    int rc = ln_check_rows("ln_gather_backward", grad_gathered, idx, grad_values, n, pos_dim, val_dim);
    if (rc) return rc;
    if (n == 0) return LN_OK;
    const long long work = (long long)n * (pos_dim + 1) * (val_dim + 1);
    LN_LAUNCH("k_gather_backward", k_gather_backward, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream, grad_gathered, idx, w, work,
                       val_dim, grad_values);
    return ln_check_launch("ln_gather_backward");
}

// ------------------------------------------------------------------------------------------
// im2row family from a neighbour list
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256)
    k_im2row(const int* __restrict__ nbr, const float* __restrict__ values, long long work, int chunks, float* __restrict__ out) {
    using T = typename VecT<VEC>::type;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long me = g / chunks;  // m*E+e
    const int c = int(g - me * chunks);
    const int row = nbr[me];
    T v = ln_zero(T());
    if (row >= 0) v = reinterpret_cast<const T*>(values)[(size_t)row * chunks + c];
    reinterpret_cast<T*>(out)[g] = v;
}

extern "C" int ln_im2row(const int* nbr, const float* values_neigh, int m, int filter_extent, int val_dim, float* out,
                         void* stream) {
    LN_REQUIRE(m >= 0 && filter_extent >= 3 && val_dim >= 1, LN_ERR_ARG, "ln_im2row: bad sizes");
    LN_REQUIRE(m == 0 || (nbr && values_neigh && out), LN_ERR_ARG, "ln_im2row: null buffer");
    if (m == 0) return LN_OK;
    LN_DISPATCH_VEC(val_dim, {
        const int chunks = val_dim / VEC;
        const long long work = (long long)m * filter_extent * chunks;
        LN_LAUNCH("k_im2row", k_im2row<VEC>, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream, nbr, values_neigh, work,
                           chunks, out);
    });
    return ln_check_launch("ln_im2row");
}

__global__ void __launch_bounds__(256) k_im2rowindices(const int* __restrict__ nbr, long long work, int E, int V, int* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long me = g / V;
    const int e = int(me % E);
    int code = nbr[me];
    // neighbour slots hold -1 when visited-but-absent; the centre is only written when found;
    // never-visited slots keep the tensor's initial 0 (LatticeGPU.cuh:1844-1915, Lattice.cu:600)
    if (code == LN_NOT_VISITED) code = 0;
    if (e == E - 1 && code < 0) code = 0;
    out[g] = code;
}

extern "C" int ln_im2rowindices(const int* nbr, int m, int filter_extent, int val_dim, int* out, void* stream) {
    LN_REQUIRE(m >= 0 && filter_extent >= 3 && val_dim >= 1, LN_ERR_ARG, "ln_im2rowindices: bad sizes");
    LN_REQUIRE(m == 0 || (nbr && out), LN_ERR_ARG, "ln_im2rowindices: null buffer");
    if (m == 0) return LN_OK;
    const long long work = (long long)m * filter_extent * val_dim;
    LN_LAUNCH("k_im2rowindices", k_im2rowindices, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream, nbr, work, filter_extent,
                       val_dim, out);
    return ln_check_launch("ln_im2rowindices");
}

// row2im (LatticeGPU.cuh:2187-2284): out[m] = sum_a rows[np_a][slot 2a+1] + rows[nm_a][slot 2a], then the centre.
template <int VEC>
__global__ void __launch_bounds__(256)
    k_row2im(const int* __restrict__ nbr, const float* __restrict__ rowified, long long work, int E, int chunks,
             float* __restrict__ out) {
    using T = typename VecT<VEC>::type;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long m = g / chunks;
    const int c = int(g - m * chunks);
    const int axes = (E - 1) / 2;
    const T* rows = reinterpret_cast<const T*>(rowified);
    T acc = ln_zero(T());
    for (int a = 0; a < axes; ++a) {
        int r = nbr[m * E + 2 * a];
        if (r >= 0) acc = ln_add(acc, rows[((size_t)r * E + 2 * a + 1) * chunks + c]);
        r = nbr[m * E + 2 * a + 1];
        if (r >= 0) acc = ln_add(acc, rows[((size_t)r * E + 2 * a) * chunks + c]);
    }
    const int r = nbr[m * E + E - 1];
    if (r >= 0) acc = ln_add(acc, rows[((size_t)r * E + E - 1) * chunks + c]);
    reinterpret_cast<T*>(out)[g] = acc;
}

extern "C" int ln_row2im(const int* nbr, const float* rowified, int m, int filter_extent, int val_dim, float* out, void* stream) {
    LN_REQUIRE(m >= 0 && filter_extent >= 3 && (filter_extent & 1) && val_dim >= 1, LN_ERR_ARG, "ln_row2im: bad sizes");
    LN_REQUIRE(m == 0 || (nbr && rowified && out), LN_ERR_ARG, "ln_row2im: null buffer");
    if (m == 0) return LN_OK;
    LN_DISPATCH_VEC(val_dim, {
        const int chunks = val_dim / VEC;
        const long long work = (long long)m * chunks;
        LN_LAUNCH("k_row2im", k_row2im<VEC>, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream, nbr, rowified, work,
                           filter_extent, chunks, out);
    });
    return ln_check_launch("ln_row2im");
}

// ------------------------------------------------------------------------------------------
// fused slice + linear classifier (LatticeGPU.cuh:3405-3460, 3648-3751)
// ------------------------------------------------------------------------------------------
// A workgroup walks tiles of PB points.  The sliced features h[PB, V] (barycentric weights w + delta_w) are
// computed ONCE per point into LDS — the reference recomputes the d+1 row gathers for every class — next to the
// classifier W[C, V] (row stride V+1: conflict-free when lanes run over classes).  The per-(point, class) sums
// keep the reference's order (rows r ascending, then channels v ascending; separate multiply and add), so the
// logits are bit-identical to the serial evaluation.
template <int PB>
__global__ void __launch_bounds__(256)
    k_slice_classify_forward(const float* __restrict__ values, const float* __restrict__ delta_w, const float* __restrict__ lin_w,
                             const float* __restrict__ lin_b, const int* __restrict__ idx, const float* __restrict__ w, int n,
                             int dp1, int V, int C, float* __restrict__ logits) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w = smem;                 // [C, V+1]
    float* s_h = s_w + C * (V + 1);    // [PB, V]
    const int tid = threadIdx.x;
    for (int i = tid; i < C * V; i += 256) {
        const int c = i / V;
        s_w[c * (V + 1) + (i - c * V)] = lin_w[i];
    }
    const int tiles = (n + PB - 1) / PB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long p0 = (long long)tile * PB;
        __syncthreads();  // s_w staged / previous tile's s_h consumed
        for (int i = tid; i < PB * V; i += 256) {
            const int lp = i / V, v = i - lp * V;
            const long long p = p0 + lp;
            float h = 0.0f;
            if (p < n) {
                for (int r = 0; r < dp1; ++r) {
                    const int row = idx[p * dp1 + r];
                    if (row >= 0) h = h + values[(size_t)row * V + v] * (w[p * dp1 + r] + delta_w[p * dp1 + r]);
                }
            }
            s_h[i] = h;
        }
        __syncthreads();
        for (int o = tid; o < PB * C; o += 256) {
            const int lp = o / C, c = o - lp * C;
            const long long p = p0 + lp;
            if (p >= n) continue;
            float acc = 0.0f;
            const float* hw = s_h + lp * V;
            const float* ww = s_w + c * (V + 1);
            for (int v = 0; v < V; ++v) acc = acc + ww[v] * hw[v];
            logits[p * C + c] = acc + lin_b[c];
        }
    }
}

// Same result, bit for bit (each logit is still the serial sum over v of W[c,v] * h[p,v], multiply and add kept apart), for
// V % 4 == 0: the d+1 row indices and effective weights of a tile are staged in LDS once (the scalar kernel re-reads them
// for every channel — three quarters of its L1 traffic), rows are gathered as float4, and a thread owns one point and every
// TPP-th class, so a channel of h is read from LDS once for all of the thread's classes.
#define LN_SC_FWD_MAX_CPT 16  // classes per thread
template <int PB>
__global__ void __launch_bounds__(256)
    k_slice_classify_forward_v4(const float* __restrict__ values, const float* __restrict__ delta_w, const float* __restrict__ lin_w,
                                const float* __restrict__ lin_b, const int* __restrict__ idx, const float* __restrict__ w, int n,
                                int dp1, int V, int C, float* __restrict__ logits) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int SH = V + 1;                // row stride of W and h: lanes that differ in class / point hit different banks
    float* s_w = smem;                   // [C, V+1]
    float* s_h = s_w + C * SH;           // [PB, V+1]
    float* s_we = s_h + PB * SH;         // [PB, dp1]  w + delta_w
    int* s_idx = reinterpret_cast<int*>(s_we + PB * dp1);  // [PB, dp1]
    constexpr int TPP = 256 / PB;        // threads per point in the classifier phase
    const int tid = threadIdx.x;
    const int V4 = V >> 2;
    for (int i = tid; i < C * V; i += 256) {
        const int c = i / V;
        s_w[c * SH + (i - c * V)] = lin_w[i];
    }
    const int tiles = (n + PB - 1) / PB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long p0 = (long long)tile * PB;
        __syncthreads();  // s_w staged / previous tile consumed
        for (int i = tid; i < PB * dp1; i += 256) {
            const long long t = p0 * dp1 + i;
            const bool ok = t < (long long)n * dp1;
            s_idx[i] = ok ? idx[t] : -1;
            s_we[i] = ok ? w[t] + delta_w[t] : 0.0f;
        }
        __syncthreads();
        for (int i = tid; i < PB * V4; i += 256) {
            const int lp = i / V4, v4 = i - lp * V4;
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
            // all d+1 row gathers in flight before the first use (a rolled loop with the load inside its branch waits for
            // each row in turn); summation order and skips unchanged
            int rows[LN_MAX_POS_DIM + 1];
            float4 x[LN_MAX_POS_DIM + 1];
#pragma unroll
            for (int r = 0; r <= LN_MAX_POS_DIM; ++r) {
                rows[r] = -1;
                x[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < dp1) {  // wave-uniform
                    rows[r] = s_idx[lp * dp1 + r];
                    x[r] = reinterpret_cast<const float4*>(values + (size_t)(rows[r] >= 0 ? rows[r] : 0) * V)[v4];
                }
            }
#pragma unroll
            for (int r = 0; r <= LN_MAX_POS_DIM; ++r) {
                if (rows[r] >= 0) {
                    const float wt = s_we[lp * dp1 + r];
                    h.x = h.x + x[r].x * wt; h.y = h.y + x[r].y * wt; h.z = h.z + x[r].z * wt; h.w = h.w + x[r].w * wt;
                }
            }
            float* d = s_h + lp * SH + v4 * 4;
            d[0] = h.x; d[1] = h.y; d[2] = h.z; d[3] = h.w;
        }
        __syncthreads();
        {
            const int lp = tid / TPP, q = tid - lp * TPP;
            const long long p = p0 + lp;
            if (p < n) {
                float acc[LN_SC_FWD_MAX_CPT];
#pragma unroll
                for (int k = 0; k < LN_SC_FWD_MAX_CPT; ++k) acc[k] = 0.0f;
                const float* hw = s_h + lp * SH;
                for (int v = 0; v < V; ++v) {
                    const float hv = hw[v];
#pragma unroll
                    for (int k = 0; k < LN_SC_FWD_MAX_CPT; ++k) {
                        const int c = q + k * TPP;
                        if (c < C) acc[k] = acc[k] + s_w[c * SH + v] * hv;
                    }
                }
#pragma unroll
                for (int k = 0; k < LN_SC_FWD_MAX_CPT; ++k) {
                    const int c = q + k * TPP;
                    if (c < C) logits[p * C + c] = acc[k] + lin_b[c];
                }
            }
        }
    }
}

static int ln_sc_points_per_tile(int V, int C, int arrays_of_v) {  // largest PB in {64,32,16,8} whose tiles fit 64 KiB of LDS
    for (int pb = 64; pb >= 8; pb >>= 1) {
        const size_t lds = sizeof(float) * ((size_t)C * (V + 1) + (size_t)pb * ((size_t)arrays_of_v * V + C));
        if (lds <= 64 * 1024) return pb;
    }
    return 0;
}

extern "C" int ln_slice_classify_forward(const float* values, const float* delta_w, const float* lin_w, const float* lin_b,
                                         const int* idx, const float* w, int n, int pos_dim, int val_dim, int nr_classes,
                                         float* logits, void* stream) {
    int rc = ln_check_rows("ln_slice_classify_forward", values, idx, logits, n, pos_dim, val_dim);
    if (rc) return rc;
    LN_REQUIRE(nr_classes >= 1 && (n == 0 || (delta_w && lin_w && lin_b && w)), LN_ERR_ARG, "ln_slice_classify_forward: bad args");
    if (n == 0) return LN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!(ln_debug_mask() & 1048576) && ln_sc_forward_wave(values, delta_w, lin_w, lin_b, idx, w, n, pos_dim, val_dim, nr_classes, logits, st))
        return ln_check_launch("ln_slice_classify_forward");  // wave-tiled kernel (ln_classify.hip): V % 32 == 0, C <= 32, d in {2, 3}
    if (val_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(values) & 15) == 0) {
        // float4 kernel: largest tile whose [C + PB, V+1] + 2 [PB, d+1] arrays fit 64 KiB and whose threads own <= 16 classes each
        for (int pb = 64; pb >= 16; pb >>= 1) {
            const size_t lds4 = sizeof(float) * ((size_t)(nr_classes + pb) * (val_dim + 1) + 2 * (size_t)pb * (pos_dim + 1));
            const int cpt = ln_div_up(nr_classes, 256 / pb);
            if (lds4 > 64 * 1024 || cpt > LN_SC_FWD_MAX_CPT) continue;
            int grid4 = ln_div_up(n, pb);
            if (grid4 > 2048) grid4 = 2048;
            if (pb == 64)
                LN_LAUNCH("k_slice_classify_forward", k_slice_classify_forward_v4<64>, dim3(grid4), dim3(256), lds4, st, values, delta_w, lin_w, lin_b,
                          idx, w, n, pos_dim + 1, val_dim, nr_classes, logits);
            else if (pb == 32)
                LN_LAUNCH("k_slice_classify_forward", k_slice_classify_forward_v4<32>, dim3(grid4), dim3(256), lds4, st, values, delta_w, lin_w, lin_b,
                          idx, w, n, pos_dim + 1, val_dim, nr_classes, logits);
            else
                LN_LAUNCH("k_slice_classify_forward", k_slice_classify_forward_v4<16>, dim3(grid4), dim3(256), lds4, st, values, delta_w, lin_w, lin_b,
                          idx, w, n, pos_dim + 1, val_dim, nr_classes, logits);
            return ln_check_launch("ln_slice_classify_forward");
        }
    }
    const int pb = ln_sc_points_per_tile(val_dim, nr_classes, 1);
    LN_REQUIRE(pb > 0, LN_ERR_UNSUPPORTED, "ln_slice_classify_forward: V=%d C=%d do not fit 64 KiB of LDS", val_dim, nr_classes);
    const size_t lds = sizeof(float) * ((size_t)nr_classes * (val_dim + 1) + (size_t)pb * val_dim);
    int grid = ln_div_up(n, pb);
    if (grid > 2048) grid = 2048;
#define LN_SC_FWD(P)                                                                                                          \
    if (pb == P)                                                                                                                \
        LN_LAUNCH("k_slice_classify_forward", k_slice_classify_forward<P>, dim3(grid), dim3(256), lds, st, values, delta_w, lin_w, lin_b, idx, w, \
                  n, pos_dim + 1, val_dim, nr_classes, logits);
    LN_SC_FWD(64) LN_SC_FWD(32) LN_SC_FWD(16) LN_SC_FWD(8)
#undef LN_SC_FWD
    return ln_check_launch("ln_slice_classify_forward");
}

// Backward.  Per tile of PB points, in LDS: h[PB,V] (as forward), g[PB,C] (incoming gradient), W[C,V+1] and
// gh[PB,V] = g @ W (the gradient wrt the sliced features).  Outputs:
//   grad_sliced[p,:] = gh            -> the caller scatters it onto the lattice with weights w_eff = w + delta_w
//                                       (ln_csr_reduce_rows: no atomics; the reference issues N(d+1)V of them)
//   g_delta_w[p,r]  += values[idx[p,r],:] . gh[p,:]                       (one writer each)
//   classifier weight / bias gradients: accumulated in registers over all tiles of the workgroup, written as one
//   slab per workgroup and summed by k_sc_reduce_slabs (the reference: N*C*V atomics onto C*V addresses).
#define LN_SC_MAX_ACC 16  // C*V <= 256 * LN_SC_MAX_ACC
template <int PB>
__global__ void __launch_bounds__(256)
    k_slice_classify_backward(const float* __restrict__ grad_logits, const float* __restrict__ values,
                              const float* __restrict__ delta_w, const float* __restrict__ lin_w, const int* __restrict__ idx,
                              const float* __restrict__ w, int n, int dp1, int V, int C, float* __restrict__ g_delta_w,
                              float* __restrict__ grad_sliced, float* __restrict__ w_eff, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w = smem;                  // [C, V+1]
    float* s_h = s_w + C * (V + 1);     // [PB, V]
    float* s_gh = s_h + PB * V;         // [PB, V]
    float* s_g = s_gh + PB * V;         // [PB, C]
    const int tid = threadIdx.x;
    const int CV = C * V;
    for (int i = tid; i < CV; i += 256) {
        const int c = i / V;
        s_w[c * (V + 1) + (i - c * V)] = lin_w[i];
    }
    float acc_w[LN_SC_MAX_ACC];
#pragma unroll
    for (int k = 0; k < LN_SC_MAX_ACC; ++k) acc_w[k] = 0.0f;
    float acc_b = 0.0f;
    const int tiles = (n + PB - 1) / PB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long p0 = (long long)tile * PB;
        __syncthreads();
        for (int i = tid; i < PB * C; i += 256) {
            const long long p = p0 + i / C;
            s_g[i] = p < n ? grad_logits[p0 * C + i] : 0.0f;
        }
        for (int i = tid; i < PB * V; i += 256) {
            const int lp = i / V, v = i - lp * V;
            const long long p = p0 + lp;
            float h = 0.0f;
            if (p < n) {
                for (int r = 0; r < dp1; ++r) {
                    const int row = idx[p * dp1 + r];
                    if (row >= 0) h = h + values[(size_t)row * V + v] * (w[p * dp1 + r] + delta_w[p * dp1 + r]);
                }
            }
            s_h[i] = h;
        }
        for (int i = tid; i < PB * dp1; i += 256) {
            const long long t = p0 * dp1 + i;
            if (t < (long long)n * dp1) w_eff[t] = w[t] + delta_w[t];
        }
        __syncthreads();
        for (int i = tid; i < PB * V; i += 256) {
            const int lp = i / V, v = i - lp * V;
            float gw = 0.0f;
            for (int c = 0; c < C; ++c) gw = gw + s_g[lp * C + c] * s_w[c * (V + 1) + v];
            s_gh[i] = gw;
            if (p0 + lp < n) grad_sliced[(size_t)(p0 + lp) * V + v] = gw;
        }
        __syncthreads();
        for (int i = tid; i < PB * dp1; i += 256) {
            const int lp = i / dp1, r = i - lp * dp1;
            const long long p = p0 + lp;
            if (p >= n) continue;
            const int row = idx[p * dp1 + r];
            if (row < 0) continue;
            const float* vr = values + (size_t)row * V;
            const float* gh = s_gh + lp * V;
            float dot = 0.0f;
            for (int v = 0; v < V; ++v) dot = dot + vr[v] * gh[v];
            g_delta_w[p * dp1 + r] += dot;
        }
#pragma unroll
        for (int k = 0; k < LN_SC_MAX_ACC; ++k) {
            const int j = tid + k * 256;
            if (j < CV) {
                const int c = j / V, v = j - c * V;
                float a = acc_w[k];
                for (int lp = 0; lp < PB; ++lp) a = a + s_h[lp * V + v] * s_g[lp * C + c];
                acc_w[k] = a;
            }
        }
        if (tid < C)
            for (int lp = 0; lp < PB; ++lp) acc_b = acc_b + s_g[lp * C + tid];
    }
    float* slab = slabs + (size_t)blockIdx.x * (CV + C);
#pragma unroll
    for (int k = 0; k < LN_SC_MAX_ACC; ++k) {
        const int j = tid + k * 256;
        if (j < CV) slab[j] = acc_w[k];
    }
    if (tid < C) slab[CV + tid] = acc_b;
}

// Same contract for V % 4 == 0 (the usual case), with the LDS-heavy loops register-blocked: the classifier-gradient
// accumulation owns 4x4 blocks of (class, channel) pairs per thread (two float4 LDS reads per 16 FMAs instead of two
// scalar reads per FMA), gh = g @ W and the delta-weight dots run on float4 lanes.  LDS: h[PB,V] | gh[PB,V] | g[PB,CP] | W[C,V]
// with CP = C rounded up to 4 (zero padded).
#define LN_SC_BLOCKS 3  // 4x4 blocks per thread: CP*V <= 256 * 3 * 16 = 12288
template <int PB>
__global__ void __launch_bounds__(256)
    k_slice_classify_backward_v4(const float* __restrict__ grad_logits, const float* __restrict__ values,
                                 const float* __restrict__ delta_w, const float* __restrict__ lin_w, const int* __restrict__ idx,
                                 const float* __restrict__ w, int n, int dp1, int V, int C, float* __restrict__ g_delta_w,
                                 float* __restrict__ grad_sliced, float* __restrict__ w_eff, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CP = (C + 3) & ~3;
    float* s_h = smem;                  // [PB, V]
    float* s_gh = s_h + PB * V;         // [PB, V]
    float* s_g = s_gh + PB * V;         // [PB, CP]
    float* s_w = s_g + PB * CP;         // [C, V]
    float* s_we = s_w + C * V;          // [PB, dp1]  w + delta_w
    int* s_idx = reinterpret_cast<int*>(s_we + PB * dp1);  // [PB, dp1]
    const int tid = threadIdx.x;
    const int CV = C * V;
    const int V4 = V >> 2;
    for (int i = tid; i < CV; i += 256) s_w[i] = lin_w[i];
    const int cblocks = CP >> 2;
    const int nblocks = cblocks * V4;   // 4x4 blocks of (class, channel)
    float acc[LN_SC_BLOCKS][16];
#pragma unroll
    for (int k = 0; k < LN_SC_BLOCKS; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.0f;
    float acc_b = 0.0f;
    const int tiles = (n + PB - 1) / PB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long p0 = (long long)tile * PB;
        __syncthreads();
        for (int i = tid; i < PB * CP; i += 256) {
            const int lp = i / CP, c = i - lp * CP;
            const long long p = p0 + lp;
            s_g[i] = (p < n && c < C) ? grad_logits[p * C + c] : 0.0f;
        }
        for (int i = tid; i < PB * dp1; i += 256) {  // the tile's row indices and effective weights, once
            const long long t = p0 * dp1 + i;
            const bool ok = t < (long long)n * dp1;
            const float we = ok ? w[t] + delta_w[t] : 0.0f;
            s_idx[i] = ok ? idx[t] : -1;
            s_we[i] = we;
            if (ok) w_eff[t] = we;
        }
        __syncthreads();
        for (int i = tid; i < PB * V4; i += 256) {
            const int lp = i / V4, v4 = i - lp * V4;
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
            // all d+1 row gathers in flight before the first use (a rolled loop with the load inside its branch waits for
            // each row in turn); summation order and skips unchanged
            int rows[LN_MAX_POS_DIM + 1];
            float4 x[LN_MAX_POS_DIM + 1];
#pragma unroll
            for (int r = 0; r <= LN_MAX_POS_DIM; ++r) {
                rows[r] = -1;
                x[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < dp1) {  // wave-uniform
                    rows[r] = s_idx[lp * dp1 + r];
                    x[r] = reinterpret_cast<const float4*>(values + (size_t)(rows[r] >= 0 ? rows[r] : 0) * V)[v4];
                }
            }
#pragma unroll
            for (int r = 0; r <= LN_MAX_POS_DIM; ++r) {
                if (rows[r] >= 0) {
                    const float wt = s_we[lp * dp1 + r];
                    h.x = h.x + x[r].x * wt; h.y = h.y + x[r].y * wt; h.z = h.z + x[r].z * wt; h.w = h.w + x[r].w * wt;
                }
            }
            reinterpret_cast<float4*>(s_h)[i] = h;
        }
        __syncthreads();
        for (int i = tid; i < PB * V4; i += 256) {  // gh = g @ W
            const int lp = i / V4, v4 = i - lp * V4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* gr = s_g + lp * CP;
            for (int c = 0; c < C; ++c) {
                const float g = gr[c];
                const float4 w4 = reinterpret_cast<const float4*>(s_w + c * V)[v4];
                a.x = fmaf(g, w4.x, a.x); a.y = fmaf(g, w4.y, a.y); a.z = fmaf(g, w4.z, a.z); a.w = fmaf(g, w4.w, a.w);
            }
            reinterpret_cast<float4*>(s_gh)[i] = a;
            if (p0 + lp < n) reinterpret_cast<float4*>(grad_sliced + (size_t)(p0 + lp) * V)[v4] = a;
        }
        __syncthreads();
        for (int i = tid; i < PB * dp1; i += 256) {  // delta-weight gradients
            const int lp = i / dp1, r = i - lp * dp1;
            const long long p = p0 + lp;
            if (p >= n) continue;
            const int row = s_idx[i];
            if (row < 0) continue;
            const float4* vr = reinterpret_cast<const float4*>(values + (size_t)row * V);
            const float4* gh = reinterpret_cast<const float4*>(s_gh + lp * V);
            float dot = 0.0f;
            for (int v4 = 0; v4 < V4; ++v4) {
                const float4 a = vr[v4], b = gh[v4];
                dot = fmaf(a.x, b.x, dot); dot = fmaf(a.y, b.y, dot); dot = fmaf(a.z, b.z, dot); dot = fmaf(a.w, b.w, dot);
            }
            g_delta_w[p * dp1 + r] += dot;
        }
#pragma unroll
        for (int k = 0; k < LN_SC_BLOCKS; ++k) {  // classifier weight gradient, 4 classes x 4 channels per block
            const int blk = tid + k * 256;
            if (blk < nblocks) {
                const int cb = blk / V4, vb = blk - cb * V4;
                const float* pg = s_g + cb * 4;
                const float* ph = s_h + vb * 4;
#pragma unroll 4
                for (int lp = 0; lp < PB; ++lp) {
                    const float4 g4 = *reinterpret_cast<const float4*>(pg + lp * CP);
                    const float4 h4 = *reinterpret_cast<const float4*>(ph + lp * V);
                    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
                    const float hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[k][a * 4 + b] = fmaf(gg[a], hh[b], acc[k][a * 4 + b]);
                }
            }
        }
        if (tid < C)
            for (int lp = 0; lp < PB; ++lp) acc_b = acc_b + s_g[lp * CP + tid];
    }
    float* slab = slabs + (size_t)blockIdx.x * (CV + C);
#pragma unroll
    for (int k = 0; k < LN_SC_BLOCKS; ++k) {
        const int blk = tid + k * 256;
        if (blk < nblocks) {
            const int cb = blk / V4, vb = blk - cb * V4;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (cb * 4 + a < C)
#pragma unroll
                    for (int b = 0; b < 4; ++b) slab[(cb * 4 + a) * V + vb * 4 + b] = acc[k][a * 4 + b];
        }
    }
    if (tid < C) slab[CV + tid] = acc_b;
}

// grad_sliced rows scattered with global atomics (callers without a CSR adjacency)
__global__ void __launch_bounds__(256)
    k_sc_scatter_atomic(const float* __restrict__ grad_sliced, const float* __restrict__ w_eff, const int* __restrict__ idx, long long work,
                        int dp1, int V, float* __restrict__ g_values) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long t = g / V;
    const int v = int(g - t * V);
    const int row = idx[t];
    if (row >= 0) ln_atomic_add(g_values + (size_t)row * V + v, grad_sliced[(t / dp1) * V + v] * w_eff[t]);
}

static int ln_sc_backward_grid(int n, int pb) {
    int grid = ln_div_up(n, pb);
    return grid > 512 ? 512 : grid;
}

extern "C" size_t ln_slice_classify_backward_workspace_bytes(int n, int pos_dim, int val_dim, int nr_classes) {
    if (n < 1) n = 1;
    return (size_t)LN_SC_MAX_SLABS * ((size_t)nr_classes * val_dim + nr_classes) * sizeof(float) + 256;
}

extern "C" int ln_slice_classify_backward(const float* grad_logits, const float* values, const float* delta_w, const float* lin_w,
                                          const int* idx, const float* w, int n, int pos_dim, int val_dim, int nr_classes,
                                          float* g_values, float* g_delta_w, float* g_lin_w, float* g_lin_b, float* grad_sliced,
                                          float* w_eff, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = ln_check_rows("ln_slice_classify_backward", grad_logits, values, idx, n, pos_dim, val_dim);
    if (rc) return rc;
    LN_REQUIRE(nr_classes >= 1 && (n == 0 || (delta_w && lin_w && w && g_delta_w && g_lin_w && g_lin_b && grad_sliced && w_eff)), LN_ERR_ARG,
               "ln_slice_classify_backward: bad args");
    if (n == 0) return LN_OK;
    LN_REQUIRE(workspace && workspace_bytes >= ln_slice_classify_backward_workspace_bytes(n, pos_dim, val_dim, nr_classes), LN_ERR_WORKSPACE,
               "ln_slice_classify_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int dp1 = pos_dim + 1;
    float* slabs = static_cast<float*>(workspace);
    const int cv = nr_classes * val_dim;
    int grid = 0;
    const bool wave_kernel = !(ln_debug_mask() & 1048576) && ln_sc_backward_wave(grad_logits, values, delta_w, lin_w, idx, w, n, pos_dim, val_dim, nr_classes,
                                                                          g_delta_w, grad_sliced, w_eff, slabs, &grid, st);
    if (!wave_kernel) {
    const int cp = (nr_classes + 3) & ~3;
    const bool v4 = (val_dim % 4 == 0) && (cp * val_dim <= 256 * LN_SC_BLOCKS * 16) &&
                    ((reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(grad_sliced)) & 15) == 0;
    int pb = 0;
    size_t lds = 0;
    if (v4) {
        for (int cand = 64; cand >= 8 && !pb; cand >>= 1) {
            lds = sizeof(float) * ((size_t)cand * (2 * val_dim + cp + 2 * (pos_dim + 1)) + (size_t)nr_classes * val_dim);
            if (lds <= 64 * 1024) pb = cand;
        }
    } else {
        LN_REQUIRE((long long)nr_classes * val_dim <= 256 * LN_SC_MAX_ACC, LN_ERR_UNSUPPORTED,
                   "ln_slice_classify_backward: nr_classes*val_dim = %d exceeds %d (val_dim %% 4 != 0 path)", nr_classes * val_dim,
                   256 * LN_SC_MAX_ACC);
        pb = ln_sc_points_per_tile(val_dim, nr_classes, 2);
        lds = sizeof(float) * ((size_t)nr_classes * (val_dim + 1) + (size_t)pb * (2 * val_dim + nr_classes));
    }
    LN_REQUIRE(pb > 0, LN_ERR_UNSUPPORTED, "ln_slice_classify_backward: V=%d C=%d do not fit 64 KiB of LDS", val_dim, nr_classes);
    grid = ln_sc_backward_grid(n, pb);
#define LN_SC_BWD(P)                                                                                                               \
    if (pb == P) {                                                                                                                 \
        if (v4)                                                                                                                    \
            LN_LAUNCH("k_slice_classify_backward", k_slice_classify_backward_v4<P>, dim3(grid), dim3(256), lds, st, grad_logits, values, delta_w, \
                      lin_w, idx, w, n, dp1, val_dim, nr_classes, g_delta_w, grad_sliced, w_eff, slabs);                           \
        else                                                                                                                       \
            LN_LAUNCH("k_slice_classify_backward", k_slice_classify_backward<P>, dim3(grid), dim3(256), lds, st, grad_logits, values, delta_w, \
                      lin_w, idx, w, n, dp1, val_dim, nr_classes, g_delta_w, grad_sliced, w_eff, slabs);                           \
    }
    LN_SC_BWD(64) LN_SC_BWD(32) LN_SC_BWD(16) LN_SC_BWD(8)
#undef LN_SC_BWD
    }
    // the gradient tensors are accumulated into (Lattice.cu:1091-1115)
    LN_LAUNCH("k_sc_reduce_slabs", ln_k_sum_slabs2<true>, dim3(ln_div_up(cv + nr_classes, 16)), dim3(256), 0, st, slabs, grid,
              (long long)(cv + nr_classes), cv + nr_classes, g_lin_w, g_lin_b, cv);
    if (g_values) {
        const long long work = (long long)n * dp1 * val_dim;
        LN_LAUNCH("k_sc_scatter_atomic", k_sc_scatter_atomic, dim3(ln_div_up(work, 256)), dim3(256), 0, st, grad_sliced, w_eff, idx, work, dp1,
                  val_dim, g_values);
    }
    return ln_check_launch("ln_slice_classify_backward");
}
