// TEST INFRASTRUCTURE — container-only.  Drives the reference's own device kernels
// (included from /root/reference, never copied) serially on the host through
// cuda_qualifier_shim.h, and exposes them as plain C entry points for ctypes so that
// tests/golden/make_goldens.py can dump golden vectors.
//
// Each entry point mirrors the host glue of the reference around that kernel:
//   launch geometry  <<<ceil(items/256), 256>>>          LatticeGPU.cuh:45-46 etc.
//   table reset      values=0, keys=0, entries=-1, n=0    HashTable.cu:49-57
// (positions are expected already divided by sigma, as Lattice.cu:226 does before launch).
#include "cuda_qualifier_shim.h"
#define val_full_dim val_dim
#include "lattice_net/kernels/LatticeGPU.cuh"
#undef val_full_dim

#include <cstring>

namespace {

HashTableGPU make_table(int capacity, int pos_dim, int* keys, int* entries, float* values, int* nr_filled) {
    HashTableGPU t(capacity, pos_dim);
    t.m_keys = keys;
    t.m_entries = entries;
    t.m_values = values;
    t.m_nr_filled = nr_filled;
    return t;
}

template <typename F>
void serial_launch(int items, F&& body) {
    gridDim.x = (items - 1) / 256 + 1;
    for (int i = 0; i < items; ++i) {
        blockIdx.x = i / 256;
        threadIdx.x = i % 256;
        body();
    }
}

}  // namespace

#define REF_DISPATCH_V(V, CALL)                                   \
    switch (V) {                                                  \
        case 1: { constexpr int VV = 1; CALL; } break;            \
        case 2: { constexpr int VV = 2; CALL; } break;            \
        case 3: { constexpr int VV = 3; CALL; } break;            \
        case 4: { constexpr int VV = 4; CALL; } break;            \
        case 5: { constexpr int VV = 5; CALL; } break;            \
        case 8: { constexpr int VV = 8; CALL; } break;            \
        case 16: { constexpr int VV = 16; CALL; } break;          \
        case 32: { constexpr int VV = 32; CALL; } break;          \
        default: return -2;                                       \
    }

#define REF_DISPATCH_D(D, CALL)                                   \
    switch (D) {                                                  \
        case 2: { constexpr int DD = 2; CALL; } break;            \
        case 3: { constexpr int DD = 3; CALL; } break;            \
        default: return -1;                                       \
    }

extern "C" {

// kernel_splat (LatticeGPU.cuh:707) with write flag; table arrays are caller-owned.
int ref_kernel_splat(const float* positions, int n, int d, int capacity, int* keys, int* entries, float* values,
                     int* nr_filled, int* idx, float* w, int write_idx) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d, serial_launch(n, [&] { kernel_splat<DD, 1>(positions, n, idx, w, t, write_idx != 0); }));
    return 0;
}

// splatCacheNaive (LatticeGPU.cuh:926)
int ref_splat_accumulate(float* vals, int n, int d, int v, int capacity, int* keys, int* entries, float* values,
                         int* nr_filled, int* idx, float* w) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(n, [&] { splatCacheNaive<DD, VV>(n, vals, idx, w, t); })));
    return 0;
}

// distribute (LatticeGPU.cuh:534)
int ref_distribute(float* positions, float* vals, int n, int d, int v, int capacity, int* keys, int* entries,
                   float* values, int* nr_filled, int* idx, float* w, float* distributed) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d,
                   REF_DISPATCH_V(v, serial_launch(n, [&] { distribute<DD, VV>(positions, vals, n, idx, w, distributed, t); })));
    return 0;
}

// im2row (LatticeGPU.cuh:1464); query and neighbour tables may alias.
int ref_im2row(int nr_vertices, int d, int v, float* out, int filter_extent, int dilation, int cap_q, int* keys_q,
               int* entries_q, float* values_q, int* nr_q, int cap_n, int* keys_n, int* entries_n, float* values_n,
               int* nr_n, int lvl_q, int lvl_n, int flip) {
    HashTableGPU tq = make_table(cap_q, d, keys_q, entries_q, values_q, nr_q);
    HashTableGPU tn = make_table(cap_n, d, keys_n, entries_n, values_n, nr_n);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(nr_vertices, [&] {
                                          im2row<DD, VV>(nr_vertices, out, filter_extent, dilation, tq, tn, lvl_q, lvl_n,
                                                         flip != 0, false);
                                      })));
    return 0;
}

// im2rowindices (LatticeGPU.cuh:1690)
int ref_im2rowindices(int nr_vertices, int d, int v, int* out, int filter_extent, int dilation, int cap_q, int* keys_q,
                      int* entries_q, float* values_q, int* nr_q, int cap_n, int* keys_n, int* entries_n,
                      float* values_n, int* nr_n, int lvl_q, int lvl_n, int flip) {
    HashTableGPU tq = make_table(cap_q, d, keys_q, entries_q, values_q, nr_q);
    HashTableGPU tn = make_table(cap_n, d, keys_n, entries_n, values_n, nr_n);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(nr_vertices, [&] {
                                          im2rowindices<DD, VV>(nr_vertices, out, filter_extent, dilation, tq, tn, lvl_q,
                                                                lvl_n, flip != 0, false);
                                      })));
    return 0;
}

// row2im (LatticeGPU.cuh:2067); launched over capacity as Lattice.cu:664 does.
int ref_row2im(int d, int v, float* rowified, int filter_extent, int dilation, int cap_q, int* keys_q, int* entries_q,
               float* values_q, int* nr_q, int cap_n, int* keys_n, int* entries_n, float* values_n, int* nr_n, int lvl_q,
               int lvl_n) {
    HashTableGPU tq = make_table(cap_q, d, keys_q, entries_q, values_q, nr_q);
    HashTableGPU tn = make_table(cap_n, d, keys_n, entries_n, values_n, nr_n);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(cap_q, [&] {
                                          row2im<DD, VV>(cap_q, rowified, filter_extent, dilation, tq, tn, lvl_q, lvl_n,
                                                         false);
                                      })));
    return 0;
}

// coarsen (LatticeGPU.cuh:2314)
int ref_coarsen(int d, int cap_f, int* keys_f, int* entries_f, float* values_f, int* nr_f, int cap_c, int* keys_c,
                int* entries_c, float* values_c, int* nr_c) {
    HashTableGPU tf = make_table(cap_f, d, keys_f, entries_f, values_f, nr_f);
    HashTableGPU tc = make_table(cap_c, d, keys_c, entries_c, values_c, nr_c);
    REF_DISPATCH_D(d, serial_launch(cap_f, [&] { coarsen<DD>(cap_f, tf, tc); }));
    return 0;
}

// slice_with_precomputation (LatticeGPU.cuh:2552)
int ref_slice_with_precomputation(const float* positions, float* out, int n, int d, int v, int capacity, int* keys,
                                  int* entries, float* values, int* nr_filled, int* idx, float* w) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(n, [&] {
                                          slice_with_precomputation<DD, VV>(positions, out, n, idx, w, t);
                                      })));
    return 0;
}

// slice_no_precomputation (LatticeGPU.cuh:2598)
int ref_slice_no_precomputation(const float* positions, float* out, int n, int d, int v, int capacity, int* keys,
                                int* entries, float* values, int* nr_filled, int* idx, float* w) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(n, [&] {
                                          slice_no_precomputation<DD, VV>(positions, out, n, idx, w, t);
                                      })));
    return 0;
}

// gather_with_precomputation (LatticeGPU.cuh:2886)
int ref_gather_with_precomputation(const float* positions, float* out, int n, int d, int v, int capacity, int* keys,
                                   int* entries, float* values, int* nr_filled, int* idx, float* w) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(n, [&] {
                                          gather_with_precomputation<DD, VV>(positions, out, n, idx, w, t);
                                      })));
    return 0;
}

// slice_backwards_with_precomputation_no_homogeneous (LatticeGPU.cuh:3540); `values` is the
// zeroed [M, V] gradient buffer (Lattice.cu:1079).
int ref_slice_backwards(float* grad, int n, int d, int v, int capacity, int* keys, int* entries, float* values,
                        int* nr_filled, int* idx, float* w) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(n, [&] {
                                          slice_backwards_with_precomputation_no_homogeneous<DD, VV>(n, grad, idx, w, t);
                                      })));
    return 0;
}

// gather_backwards_with_precomputation (LatticeGPU.cuh:3761)
int ref_gather_backwards(float* grad, int n, int d, int v, int capacity, int* keys, int* entries, float* values,
                         int* nr_filled, int* idx, float* w) {
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    REF_DISPATCH_D(d, REF_DISPATCH_V(v, serial_launch(n, [&] {
                                          gather_backwards_with_precomputation<DD, VV>(n, grad, idx, w, t);
                                      })));
    return 0;
}

// slice_classify_with_precomputation<d, V, C> (LatticeGPU.cuh:3387); fixtures use d=3, C=5.
int ref_slice_classify(const float* positions, float* logits, const float* delta_w, const float* lin_w,
                       const float* lin_b, int n, int d, int v, int nr_classes, int capacity, int* keys, int* entries,
                       float* values, int* nr_filled, int* idx, float* w) {
    if (d != 3) return -3;
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    if (nr_classes == 20) {  // the SemanticKITTI head (fixture F11): V = 32 / 64
        if (v == 64)
            serial_launch(n, [&] { slice_classify_with_precomputation<3, 64, 20>(positions, logits, delta_w, lin_w, lin_b, n, idx, w, t); });
        else if (v == 32)
            serial_launch(n, [&] { slice_classify_with_precomputation<3, 32, 20>(positions, logits, delta_w, lin_w, lin_b, n, idx, w, t); });
        else
            return -2;
        return 0;
    }
    if (nr_classes != 5) return -3;
    REF_DISPATCH_V(v, serial_launch(n, [&] {
                       slice_classify_with_precomputation<3, VV, 5>(positions, logits, delta_w, lin_w, lin_b, n, idx, w,
                                                                    t);
                   }));
    return 0;
}

// slice_classify_backwards_with_precomputation<d, V, C> (LatticeGPU.cuh:3628)
int ref_slice_classify_backwards(float* grad_logits, float* initial_values, int n, int d, int v, int nr_classes,
                                 float* delta_w, float* lin_w, float* lin_b, float* g_values, float* g_delta_w,
                                 float* g_lin_w, float* g_lin_b, int capacity, int* keys, int* entries, float* values,
                                 int* nr_filled, int* idx, float* w) {
    if (d != 3) return -3;
    HashTableGPU t = make_table(capacity, d, keys, entries, values, nr_filled);
    if (nr_classes == 20) {
        if (v == 64)
            serial_launch(n, [&] {
                slice_classify_backwards_with_precomputation<3, 64, 20>(n, grad_logits, initial_values, idx, w, delta_w, lin_w, lin_b, g_values,
                                                                        g_delta_w, g_lin_w, g_lin_b, t);
            });
        else if (v == 32)
            serial_launch(n, [&] {
                slice_classify_backwards_with_precomputation<3, 32, 20>(n, grad_logits, initial_values, idx, w, delta_w, lin_w, lin_b, g_values,
                                                                        g_delta_w, g_lin_w, g_lin_b, t);
            });
        else
            return -2;
        return 0;
    }
    if (nr_classes != 5) return -3;
    REF_DISPATCH_V(v, serial_launch(n, [&] {
                       slice_classify_backwards_with_precomputation<3, VV, 5>(n, grad_logits, initial_values, idx, w,
                                                                              delta_w, lin_w, lin_b, g_values, g_delta_w,
                                                                              g_lin_w, g_lin_b, t);
                   }));
    return 0;
}

}  // extern "C"
