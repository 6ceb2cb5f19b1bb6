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

