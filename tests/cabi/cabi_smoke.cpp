// Drives the C ABI (include/latticenet_hip.h) from plain C++ + the HIP runtime — no Python, no torch: the boundary a
// compiled host (the reference's src/Lattice.cu) would bind.  Build: hipcc cabi_smoke.cpp -I<repo>/include -L<pkg> -llatticenet_hip
// Checks (exit code 0 = all hold):
//   * build: every point gets d+1 valid rows, barycentric weights sum to 1, row ids < nr_filled, keys of the rows are distinct
//   * splat of constant-one features: column sums equal the number of points (partition of unity)
//   * convolution with a bank that is the identity on the centre slot reproduces the lattice values
//   * slice of a constant field is that constant
extern "C" {
#include "latticenet_hip.h"
}
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define HIPCHECK(x)                                                              \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));       \
            return 2;                                                            \
        }                                                                        \
    } while (0)
#define LNCHECK(x)                                                               \
    do {                                                                         \
        int rc_ = (x);                                                           \
        if (rc_ != LN_OK) {                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, ln_last_error_string()); \
            return 3;                                                            \
        }                                                                        \
    } while (0)
#define EXPECT(cond, ...)                      \
    do {                                       \
        if (!(cond)) {                         \
            fprintf(stderr, "CHECK FAILED: "); \
            fprintf(stderr, __VA_ARGS__);      \
            fprintf(stderr, "\n");             \
            return 4;                          \
        }                                      \
    } while (0)

template <class T>
static T* dmalloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    return static_cast<T*>(p);
}

int main() {
    const int n = 20000, d = 3, V = 32, F = 32, E = 9, cap = 60000;
    const int tokens = n * (d + 1);
    std::vector<float> pos(n * d), ones((size_t)n * V, 1.0f);
    unsigned s = 12345u;
    for (auto& x : pos) {
        s = s * 1664525u + 1013904223u;
        x = ((s >> 8) / 16777216.0f - 0.5f) * 6.0f;
    }
    const float sigmas[3] = {0.4f, 0.4f, 0.4f};

    LnTable t{};
    t.capacity = cap;
    t.pos_dim = d;
    t.slot_keys = dmalloc<unsigned long long>(cap);
    t.slot_tok = dmalloc<unsigned int>(cap);
    t.slot_cnt = dmalloc<int>(cap);
    t.entries = dmalloc<int>(cap);
    t.keys = dmalloc<int>((size_t)cap * d);
    int* counters = dmalloc<int>(2);
    t.nr_filled = counters;
    t.status = counters + 1;
    t.host_counters = nullptr;
    t.host_seq = 0;
    HIPCHECK(hipMemset(t.slot_cnt, 0, cap * sizeof(int)));  // the table's scratch starts (and stays) zero between builds

    float* d_pos = dmalloc<float>(pos.size());
    float* d_vals = dmalloc<float>(ones.size());
    float* d_table_values = dmalloc<float>((size_t)cap * V);
    int* d_idx = dmalloc<int>(tokens);
    float* d_w = dmalloc<float>(tokens);
    HIPCHECK(hipMemcpy(d_pos, pos.data(), pos.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_vals, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));

    const long long S = ln_csr_max_segments(tokens, cap);
    // seg_desc[G*S*4] first (16-byte aligned: hipMalloc is) | grp_start[cap+1] | csr_tok[tokens] | seg_count[G+2]
    const size_t G = LN_XCD_GROUPS;
    int* csr_buf = dmalloc<int>(4 * G * S + (size_t)cap + 1 + tokens + G + 2);
    int* c0 = csr_buf + 4 * G * S;
    LnCsr csr{c0, c0 + cap + 1, csr_buf, c0 + cap + 1 + tokens, S, nullptr};
    const size_t ws_bytes = ln_build_workspace_bytes(tokens, cap);
    void* ws = nullptr;
    HIPCHECK(hipMalloc(&ws, ws_bytes));
    hipStream_t st = nullptr;

    // begin_splat + splat_standalone (Lattice.cu:185-241): clear + build + accumulate
    LNCHECK(ln_build_splat(&t, d_pos, sigmas, n, d_idx, d_w, LN_BUILD_WRITE_IDX | LN_BUILD_CLEAR_FIRST, &csr, ws, ws_bytes, d_table_values,
                           (long long)cap * V, st));
    int host_counters[2];
    HIPCHECK(hipMemcpy(host_counters, counters, sizeof(host_counters), hipMemcpyDeviceToHost));
    const int m = host_counters[0];
    EXPECT(host_counters[1] == 0, "status %d", host_counters[1]);
    EXPECT(m > 1000 && m <= cap, "nr_filled %d", m);
    LNCHECK(ln_csr_reduce_rows(&csr, t.entries, S, d_vals, d_w, V, d + 1, V, d_table_values, st));

    std::vector<int> idx(tokens);
    std::vector<float> w(tokens);
    HIPCHECK(hipMemcpy(idx.data(), d_idx, tokens * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(w.data(), d_w, tokens * sizeof(float), hipMemcpyDeviceToHost));
    for (int p = 0; p < n; ++p) {
        float sum = 0.f;
        for (int r = 0; r <= d; ++r) {
            EXPECT(idx[p * 4 + r] >= 0 && idx[p * 4 + r] < m, "point %d vertex %d -> row %d", p, r, idx[p * 4 + r]);
            sum += w[p * 4 + r];
        }
        EXPECT(std::fabs(sum - 1.f) < 1e-4f, "weights of point %d sum to %f", p, sum);
    }
    std::vector<int> keys((size_t)m * d);
    HIPCHECK(hipMemcpy(keys.data(), t.keys, keys.size() * sizeof(int), hipMemcpyDeviceToHost));
    std::set<std::vector<int>> uniq;
    for (int r = 0; r < m; ++r) uniq.insert({keys[r * 3], keys[r * 3 + 1], keys[r * 3 + 2]});
    EXPECT((int)uniq.size() == m, "%d distinct keys for %d rows", (int)uniq.size(), m);

    std::vector<float> tv((size_t)m * V);
    HIPCHECK(hipMemcpy(tv.data(), d_table_values, tv.size() * sizeof(float), hipMemcpyDeviceToHost));
    double col0 = 0.0;
    for (int r = 0; r < m; ++r) col0 += tv[(size_t)r * V];
    EXPECT(std::fabs(col0 - n) < 1e-2 * n, "splat of ones sums to %f, expected %d", col0, n);

    // neighbour list + convolution with "identity on the centre slot"
    int* d_nbr = dmalloc<int>((size_t)m * E);
    LNCHECK(ln_neighbours(&t, m, &t, 1, 1, 1, 0, d_nbr, st));
    std::vector<float> bank((size_t)E * V * F, 0.f);
    for (int v = 0; v < V; ++v) bank[((size_t)(E - 1) * V + v) * F + v] = 1.f;
    float* d_bank = dmalloc<float>(bank.size());
    float* d_conv = dmalloc<float>((size_t)m * F);
    HIPCHECK(hipMemcpy(d_bank, bank.data(), bank.size() * sizeof(float), hipMemcpyHostToDevice));
    LNCHECK(ln_conv_forward(d_nbr, d_table_values, d_bank, m, E, V, F, 0, d_conv, st));
    std::vector<float> conv((size_t)m * F);
    HIPCHECK(hipMemcpy(conv.data(), d_conv, conv.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < conv.size(); ++i) EXPECT(conv[i] == tv[i], "identity convolution differs at %zu: %f vs %f", i, conv[i], tv[i]);

    // slice of a constant field
    std::vector<float> constant((size_t)m * V, 2.5f);
    HIPCHECK(hipMemcpy(d_table_values, constant.data(), constant.size() * sizeof(float), hipMemcpyHostToDevice));
    float* d_sliced = dmalloc<float>((size_t)n * V);
    LNCHECK(ln_slice_forward(d_table_values, d_idx, d_w, n, d, V, d_sliced, st));
    std::vector<float> sliced((size_t)n * V);
    HIPCHECK(hipMemcpy(sliced.data(), d_sliced, sliced.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < sliced.size(); i += 997) EXPECT(std::fabs(sliced[i] - 2.5f) < 1e-4f, "slice of a constant: %f", sliced[i]);

    printf("CABI OK: %s, n=%d, vertices=%d\n", ln_version(), n, m);
    return 0;
}
