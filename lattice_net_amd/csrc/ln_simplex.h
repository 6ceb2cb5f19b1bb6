// Simplex location on the permutohedral lattice: elevate -> nearest remainder-0 point -> rank ->
// barycentric weights -> the d+1 vertex keys.  Restates the arithmetic of kernel_splat
// (reference LatticeGPU.cuh:718-806) with strict IEEE fp32 semantics (no FMA contraction, the
// two double-precision products kept in double), so keys AND weights are bit-identical to the
// CPU oracle.  Pure functions, usable from host code for unit tests.
#pragma once
#include "ln_common.h"

#include <math.h>

template <int D>
struct LnScale {
    float sf[D];     // LatticeGPU.cuh:725-729, computed on the host in fp32
    float sigma[D];  // Lattice.cu:226 divisor
};

template <int D>
inline LnScale<D> ln_make_scale(const float* sigmas_host) {
    LnScale<D> s;
    const float inv_std_dev = float(D + 1) * sqrtf(2.0f / 3);
    for (int i = 0; i < D; ++i) {
        s.sf[i] = 1.0f / sqrtf(float(i + 1) * float(i + 2)) * inv_std_dev;
        s.sigma[i] = sigmas_host ? sigmas_host[i] : 1.0f;
    }
    return s;
}

template <int D>
struct LnSimplex {
    int rem0[D + 1];
    int rank[D + 1];
    float bary[D + 2];
};

template <int D>
LN_HD void ln_simplex(const float* pos_raw, const LnScale<D>& sc, LnSimplex<D>& out) {
#pragma clang fp contract(off)
    float elevated[D + 1];
    float sm = 0.0f;
#pragma unroll
    for (int i = D; i > 0; --i) {
        const float p = pos_raw[i - 1] / sc.sigma[i - 1];  // Lattice.cu:226 (IEEE division)
        const float cf = p * sc.sf[i - 1];
        const float icf = float(i) * cf;
        elevated[i] = sm - icf;
        sm = sm + cf;
    }
    elevated[0] = sm;

    constexpr double inv = 1.0 / (D + 1);
    int sum = 0;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float v = float(double(elevated[i]) * inv);  // LatticeGPU.cuh:748
        const float up = ceilf(v) * float(D + 1);
        const float down = floorf(v) * float(D + 1);
        const float du = up - elevated[i];
        const float dd = elevated[i] - down;
        out.rem0[i] = (du < dd) ? int(up) : int(down);
        sum += out.rem0[i];
    }
    sum /= (D + 1);

    float diff[D + 1];
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        out.rank[i] = 0;
        diff[i] = elevated[i] - float(out.rem0[i]);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = i + 1; j <= D; ++j) {
            if (diff[i] < diff[j])
                out.rank[i]++;
            else
                out.rank[j]++;  // ties go to j, LatticeGPU.cuh:767-770
        }
    }
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        out.rank[i] += sum;
        if (out.rank[i] < 0) {
            out.rank[i] += D + 1;
            out.rem0[i] += D + 1;
        } else if (out.rank[i] > D) {
            out.rank[i] -= D + 1;
            out.rem0[i] -= D + 1;
        }
    }

#pragma unroll
    for (int i = 0; i <= D + 1; ++i) out.bary[i] = 0.0f;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float dlt = float(double(elevated[i] - float(out.rem0[i])) * inv);  // LatticeGPU.cuh:790
        // bary[D - rank] += dlt; bary[D + 1 - rank] -= dlt, with rank in [0, D] (kept in registers)
#pragma unroll
        for (int k = 0; k <= D + 1; ++k) {
            if (k == D - out.rank[i]) out.bary[k] = out.bary[k] + dlt;
            if (k == D + 1 - out.rank[i]) out.bary[k] = out.bary[k] - dlt;
        }
    }
    out.bary[0] = float(double(out.bary[0]) + (1.0 + double(out.bary[D + 1])));  // LatticeGPU.cuh:795
}

// Batch of independent clouds in one table (LnTable.batch_points / batch_key_step): the simplex of point p of cloud c = p / batch_points
// is translated by c * step along the first lattice coordinate.  step is a multiple of D + 1, so the translated point is a lattice point
// with the same remainder, ranks and barycentric weights: the cloud's lattice is the one a build of the cloud alone produces, moved.
template <int D>
LN_HD void ln_simplex_of_cloud(LnSimplex<D>& s, long long point, int batch_points, int batch_key_step) {
    if (batch_points > 0) s.rem0[0] += int(point / batch_points) * batch_key_step;
}

// LatticeGPU.cuh:798-806: first d coordinates of simplex vertex `remainder`.
template <int D>
LN_HD void ln_vertex_key(const LnSimplex<D>& s, int remainder, int* key) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        int k = s.rem0[i] + remainder;
        if (s.rank[i] > D - remainder) k -= (D + 1);
        key[i] = k;
    }
}
