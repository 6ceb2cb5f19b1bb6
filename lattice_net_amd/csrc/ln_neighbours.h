// Neighbour traversal of one (query vertex, filter slot) pair (LatticeGPU.cuh:1479-1684), shared by k_neighbours
// (ln_table.hip) and the fused post-build launch (ln_csr.hip).
#pragma once
#include "ln_common.h"

__device__ __forceinline__ bool ln_coord_is_integer(float v) {
    float ip;
    const float frac = fabsf(modff(v, &ip));
    return !(frac > 0.0001f);  // LatticeGPU.cuh:467
}

// g = global index over query_rows_upper * E (vertex-major, slot-minor); map_n = tn's slot map in registers (loaded at kernel entry), .on = false: hashed
template <int D>
__device__ __forceinline__ void ln_neighbours_body(long long g, const LnTable& tq, int query_rows_upper, const LnTable& tn, float scale,
                                                   int dilation, int flip, int* __restrict__ nbr, const LnSlotMap& map_n) {
#pragma clang fp contract(off)
    constexpr int E = 2 * (D + 1) + 1;
    const int m = int(g / E);
    const int e = int(g - (long long)m * E);
    if (m >= query_rows_upper) return;
    int mq = *tq.nr_filled;
    if (m >= mq) {  // rows beyond the filled part: reference kernels return early (LatticeGPU.cuh:1471)
        nbr[g] = LN_NOT_VISITED;
        return;
    }
    // Same level (scale 1): every quantity of LG:1479-1684 is an integer — the scaled key, the steps +-dilation and -+dilation * D, and the
    // integrality tests (the "all coordinates integer" centre rule, the odd-(d+1) neighbour rule) hold trivially.  In float arithmetic
    // these are exact while |coordinate| < 2^24; the integer form below is taken where every coordinate of the vertex stays below 2^22
    // (then every neighbour coordinate is below 2^23 for dilation * D < 2^22) and returns what the float form returns — at a third of
    // its instructions (round 6: the traversal was 2.7 M of the chain's 22 M vector-ALU instructions per scan; no modff / roundf /
    // float compares here).  Larger coordinates (unpackable at d >= 3; possible at d <= 2) take the float form, as the reference does.
    if (scale == 1.0f && dilation < (1 << 18)) {
        int ki[D + 1];
        int isum = 0;
        bool small = true;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ki[i] = tq.keys[(size_t)m * D + i];
            isum += ki[i];
            small = small && ki[i] > -(1 << 22) && ki[i] < (1 << 22);
        }
        ki[D] = -isum;
        small = small && ki[D] > -(1 << 22) && ki[D] < (1 << 22);
        if (small) {
            int key[D + 1];
            if (e == E - 1) {  // centre, LatticeGPU.cuh:1534-1540
#pragma unroll
                for (int i = 0; i <= D; ++i) key[i] = ki[i];
            } else {
                const int axis = e >> 1;
                const bool is_np = ((e & 1) == (flip ? 1 : 0));
                const int step = is_np ? dilation : -dilation;
#pragma unroll
                for (int i = 0; i <= D; ++i) key[i] = (i == axis) ? ki[i] - step * D : ki[i] + step;
            }
            nbr[g] = ln_retrieve<D>(tn, key, map_n);
            return;
        }
    }
    float kf[D + 1];
    float ksum = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        kf[i] = float(tq.keys[(size_t)m * D + i]);
        ksum = ksum + kf[i];
    }
    kf[D] = -ksum;
    bool all_int = true;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        kf[i] = kf[i] * scale;
        if (scale < 1.0f) all_int = all_int && ln_coord_is_integer(kf[i]);
    }
    int result = LN_NOT_VISITED;
    if (e == E - 1) {  // centre, LatticeGPU.cuh:1534-1540
        if (all_int) {
            int key[D + 1];
#pragma unroll
            for (int i = 0; i <= D; ++i) key[i] = int(roundf(kf[i]));
            result = ln_retrieve<D>(tn, key, map_n);
        }
    } else {
        const bool check = (scale >= 1.0f) || !all_int;  // LatticeGPU.cuh:1547-1554
        if (check) {
            const int axis = e >> 1;
            const bool is_np = ((e & 1) == (flip ? 1 : 0));
            const float mm = (scale < 1.0f) ? scale : 1.0f;
            const float step = mm * float(dilation);
            const float big = mm * float(dilation) * float(D);
            float nf[D + 1];
            bool ok = true;
#pragma unroll
            for (int i = 0; i <= D; ++i) {
                nf[i] = is_np ? (kf[i] + step) : (kf[i] - step);
                if (i == axis) nf[i] = is_np ? (kf[i] - big) : (kf[i] + big);
            }
            if ((D + 1) % 2 != 0) {  // odd d+1: the neighbour itself must be all-integer (LatticeGPU.cuh:1581-1601)
#pragma unroll
                for (int i = 0; i <= D; ++i) ok = ok && ln_coord_is_integer(nf[i]);
            }
            if (ok) {
                int key[D + 1];
#pragma unroll
                for (int i = 0; i <= D; ++i) key[i] = int(roundf(nf[i]));
                result = ln_retrieve<D>(tn, key, map_n);
            }
        }
    }
    nbr[g] = result;
}

