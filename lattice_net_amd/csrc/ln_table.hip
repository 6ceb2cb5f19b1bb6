// Hash-table build (canonical vertex numbering), key-based coarsening, neighbour traversal and
// simplex retrieval for gfx950.  See include/latticenet_hip.h for the reference interfaces these
// replace.
//
// Build design (differs from the reference's CAS spin lock, HashTableGPU.cuh:425-484):
//   1. insert   : one 64-bit CAS claims a slot AND publishes its packed key; every insertion
//                 carries a token (its position in the serial insertion order) and the slot
//                 remembers the smallest token that touched it (atomicMin).
//   2. mark     : token t is a "first occurrence" iff it is the smallest token of a NEW slot;
//                 one __ballot per wave writes 64 flags as a single 8-byte word (no atomics).
//   3. scan     : exclusive prefix over per-block first-occurrence counts (one workgroup).
//   4. finalize : row id = nr_filled_before + rank of the slot's smallest token; the first
//                 occurrence writes entries[slot] and keys[row]; every token gets its row.
// Rows therefore come out numbered by first occurrence in token order — bit-identical to a
// serial run of the reference — without any lock, spin or fence.
#include "ln_common.h"
#include "ln_simplex.h"
#include "ln_csr.h"
#include "ln_neighbours.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_ln_error[512] = "";

void ln_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ln_error, sizeof(g_ln_error), fmt, ap);
    va_end(ap);
}

int ln_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ln_set_error("%s: %s", what, hipGetErrorString(e));
        return LN_ERR_LAUNCH;
    }
    return LN_OK;
}

extern "C" const char* ln_last_error_string(void) { return g_ln_error; }

#include <stdlib.h>
int ln_debug_mask() {
    static int mask = -1;
    if (mask < 0) {
        const char* e = getenv("LN_DEBUG_MASK");
        mask = e ? atoi(e) : 0;
    }
    return mask;
}

// ------------------------------------------------------------------------------------------
// live per-kernel timing: a (start, stop) event pair bound to each dispatch of the armed kernels
// (bench.py's roofline block is computed from these)
// ------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct LnProfState {
    char names[256] = "";  // ",name1,name2," — the kernels whose launches are timed
    int max_samples = 0;
    std::vector<hipEvent_t> starts, stops;
    std::vector<const char*> sample_names;  // launch name of every sample (string literals of the LN_LAUNCH sites)
    size_t used = 0;
    long long dropped = 0;  // matching launches that found every event pair taken
};
LnProfState g_prof;
}  // namespace

LnProfEvents ln_prof_next(const char* name) {
    LnProfEvents ev{nullptr, nullptr, false};
    if (g_prof.max_samples > 0) {
        char key[72];
        snprintf(key, sizeof(key), ",%s,", name);
        if (!strcmp(g_prof.names, ",*,") || strstr(g_prof.names, key)) {
            if (g_prof.used >= g_prof.starts.size()) {
                ++g_prof.dropped;
                return ev;
            }
            ev.start = g_prof.starts[g_prof.used];
            ev.stop = g_prof.stops[g_prof.used];
            ev.armed = true;
            g_prof.sample_names[g_prof.used] = name;
            ++g_prof.used;
        }
    }
    return ev;
}

extern "C" const char* ln_kernel_names(void) { return "k_bucket_rows,k_canon_idx,k_canon_mark,k_canon_segs,k_canon_slots,k_conv_backward_fused,k_conv_generic,k_conv_generic_f16,k_conv_mfma,k_conv_mfma_f16,k_conv_split_bank,k_conv_sum_partials,k_csr_count,k_csr_fill,k_csr_group_sizes,k_csr_reduce_segments,k_csr_scan_local,k_csr_scan_top,k_csr_copy_tokens,k_csr_segment_max,k_csr_segment_max_decode,k_csr_sort_groups,k_distribute_centre,k_finalize,k_gather_backward,k_gather_forward,k_gn_apply,k_gn_backward_apply,k_gn_stats,k_grad_filter_f16,k_grad_filter_generic,k_grad_filter_mfma,k_grad_filter_mfma_f16,k_im2row,k_im2rowindices,k_insert_coarse,k_insert_points,k_linear_act_backward_w,k_linear_act_backward_x,k_linear_act_forward,k_linear_reduce_slabs,k_mark_first,k_max_centre_backward,k_max_centre_forward,k_max_centre_sum,k_neighbours,k_nll_backward,k_nll_finish,k_nll_partials,k_point_keys,k_pointnet_reduce_backward,k_pointnet_reduce_decode,k_reduce_and_neighbours,k_reduce_slabs,k_rehash_clear,k_rehash_rows,k_retrieve_points,k_row2im,k_sc_reduce_slabs,k_sc_scatter_atomic,k_scan_blocks,k_scatter_point_rows,k_seg_min,k_slice_classify_backward,k_slice_classify_forward,k_slice_forward,k_slice_forward_f16,k_table_clear,k_weight_norm_backward,k_weight_norm_forward,ln_k_arena_init"; }

extern "C" int ln_profile_begin(const char* kernel_names, int max_samples) {
    LN_REQUIRE(kernel_names && strlen(kernel_names) + 3 < sizeof(g_prof.names) && max_samples > 0, LN_ERR_ARG, "ln_profile_begin: bad args");
    LN_REQUIRE(g_prof.max_samples == 0, LN_ERR_ARG, "ln_profile_begin: profiling already armed for %s", g_prof.names);
    g_prof.starts.resize(max_samples);
    g_prof.stops.resize(max_samples);
    g_prof.sample_names.assign(max_samples, nullptr);
    for (int i = 0; i < max_samples; ++i) {
        if (hipEventCreate(&g_prof.starts[i]) != hipSuccess || hipEventCreate(&g_prof.stops[i]) != hipSuccess) {
            ln_set_error("ln_profile_begin: hipEventCreate failed");
            return LN_ERR_LAUNCH;
        }
    }
    snprintf(g_prof.names, sizeof(g_prof.names), ",%s,", kernel_names);
    g_prof.used = 0;
    g_prof.dropped = 0;
    g_prof.max_samples = max_samples;
    return LN_OK;
}

extern "C" int ln_profile_end(double* total_ms, int* launches) {
    LN_REQUIRE(total_ms && launches, LN_ERR_ARG, "ln_profile_end: null output");
    double total = 0.0;
    int rc = LN_OK;
    for (size_t i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.stops[i]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.starts[i], g_prof.stops[i]) != hipSuccess) {
            ln_set_error("ln_profile_end: event query failed");
            rc = LN_ERR_LAUNCH;
            break;
        }
        total += ms;
    }
    *total_ms = total;
    *launches = int(g_prof.used);
    for (size_t i = 0; i < g_prof.starts.size(); ++i) {
        (void)hipEventDestroy(g_prof.starts[i]);
        (void)hipEventDestroy(g_prof.stops[i]);
    }
    g_prof.starts.clear();
    g_prof.stops.clear();
    g_prof.used = 0;
    g_prof.max_samples = 0;
    g_prof.names[0] = 0;
    return rc;
}
extern "C" int ln_profile_end_table(char* out, int out_bytes) {
    LN_REQUIRE(out && out_bytes > 0, LN_ERR_ARG, "ln_profile_end_table: null output");
    struct Row { const char* name; int launches; double ms; };
    std::vector<Row> rows;
    int rc = LN_OK;
    for (size_t i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.stops[i]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.starts[i], g_prof.stops[i]) != hipSuccess) {
            ln_set_error("ln_profile_end_table: event query failed");
            rc = LN_ERR_LAUNCH;
            break;
        }
        size_t r = 0;
        while (r < rows.size() && strcmp(rows[r].name, g_prof.sample_names[i])) ++r;
        if (r == rows.size()) rows.push_back(Row{g_prof.sample_names[i], 0, 0.0});
        rows[r].launches += 1;
        rows[r].ms += ms;
    }
    int at = 0;
    out[0] = 0;
    bool truncated = false;
    for (const Row& r : rows) {
        const int w = snprintf(out + at, size_t(out_bytes - at), "%s %d %.6f\n", r.name, r.launches, r.ms);
        if (w < 0 || w >= out_bytes - at) {
            out[at] = 0;
            truncated = true;
            break;
        }
        at += w;
    }
    // samples that found no free event pair (ln_profile_begin's max_samples) and rows that did not fit `out` are reported, not dropped
    // silently: a last line "DROPPED <samples> <rows>" and LN_ERR_ARG
    const long long dropped = g_prof.dropped;
    if (dropped > 0 || truncated) {
        char tail[64];
        const int w = snprintf(tail, sizeof(tail), "DROPPED %lld %d\n", dropped, truncated ? 1 : 0);
        if (w > 0 && w < out_bytes) {
            if (at + w >= out_bytes) at = out_bytes - w - 1;
            memcpy(out + at, tail, size_t(w) + 1);
        }
        ln_set_error("ln_profile_end_table: %lld launches were not sampled (max_samples) / the table did not fit %d bytes", dropped, out_bytes);
        if (!rc) rc = LN_ERR_ARG;
    }
    double total;
    int launches;
    const int rc2 = ln_profile_end(&total, &launches);  // releases the events
    return rc ? rc : rc2;
}
extern "C" const char* ln_version(void) { return "latticenet_hip 0.2 (gfx950)"; }
#ifndef LN_ABI_HASH
#define LN_ABI_HASH "unknown"
#endif
extern "C" const char* ln_abi_hash(void) { return LN_ABI_HASH; }

// Phase stamps (tools/kernel_timeline.py; compiled in with -DLN_STAMPS only): thread 0 of every workgroup stores the
// 100 MHz wall clock at the marked points of k_point_keys / k_bucket_build.
#ifdef LN_STAMPS
__device__ unsigned long long* g_ln_stamps = nullptr;
extern "C" int ln_debug_set_stamps(void* buffer) {
    unsigned long long* p = static_cast<unsigned long long*>(buffer);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ln_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define LN_STAMP(slot)                                                                                         \
    do {                                                                                                       \
        if (g_ln_stamps && threadIdx.x == 0) g_ln_stamps[(size_t)blockIdx.x * 24 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define LN_STAMP(slot) do { } while (0)
#endif

#define LN_DISPATCH_D(d, ...)                                                    \
    switch (d) {                                                                 \
        case 1: { constexpr int D = 1; __VA_ARGS__; } break;                     \
        case 2: { constexpr int D = 2; __VA_ARGS__; } break;                     \
        case 3: { constexpr int D = 3; __VA_ARGS__; } break;                     \
        case 4: { constexpr int D = 4; __VA_ARGS__; } break;                     \
        case 5: { constexpr int D = 5; __VA_ARGS__; } break;                     \
        case 6: { constexpr int D = 6; __VA_ARGS__; } break;                     \
        default:                                                                 \
            ln_set_error("pos_dim %d unsupported (1..%d)", d, LN_MAX_POS_DIM);   \
            return LN_ERR_UNSUPPORTED;                                           \
    }

static int ln_check_table(const LnTable* t, const char* who) {
    LN_REQUIRE(t != nullptr, LN_ERR_ARG, "%s: null table", who);
    LN_REQUIRE(t->capacity > 0, LN_ERR_ARG, "%s: capacity %d", who, t->capacity);
    LN_REQUIRE(t->pos_dim >= 1 && t->pos_dim <= LN_MAX_POS_DIM, LN_ERR_UNSUPPORTED, "%s: pos_dim %d unsupported", who,
               t->pos_dim);
    LN_REQUIRE(t->slot_keys && t->slot_tok && t->slot_cnt && t->entries && t->keys && t->nr_filled && t->status, LN_ERR_ARG,
               "%s: table has a null buffer", who);
    return LN_OK;
}

__global__ void __launch_bounds__(256) ln_k_zero_words(unsigned int* __restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = 0u;
}

int ln_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return LN_OK;
    LN_REQUIRE(p != nullptr && (reinterpret_cast<uintptr_t>(p) & 3) == 0 && (bytes & 3) == 0, LN_ERR_ARG, "ln_zero_async: unaligned fill");
    const size_t words = bytes / 4;
    int blocks = ln_div_up((long long)words, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ln_k_zero_words, dim3(blocks), dim3(256), 0, st, static_cast<unsigned int*>(p), words);
    return ln_check_launch("ln_zero_async");
}

// One launch for the freshly allocated structure buffers of a table (keys = 0 | entries = -1 | slot counters, device counters and
// a placeholder values row = 0, carved from ONE allocation by the host): words [minus_begin, minus_end) become -1, the rest 0.
__global__ void __launch_bounds__(256) ln_k_arena_init(int* __restrict__ p, size_t words, size_t minus_begin, size_t minus_end) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = (i >= minus_begin && i < minus_end) ? -1 : 0;
}

extern "C" int ln_arena_init(int* arena, long long words, long long minus_begin, long long minus_end, void* stream) {
    LN_REQUIRE(words >= 0 && minus_begin >= 0 && minus_begin <= minus_end && minus_end <= words, LN_ERR_ARG, "ln_arena_init: bad ranges");
    if (words == 0) return LN_OK;
    LN_REQUIRE(arena != nullptr, LN_ERR_ARG, "ln_arena_init: null buffer");
    int blocks = ln_div_up(words, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    LN_LAUNCH("ln_k_arena_init", ln_k_arena_init, dim3(blocks), dim3(256), 0, (hipStream_t)stream, arena, (size_t)words, (size_t)minus_begin,
              (size_t)minus_end);
    return ln_check_launch("ln_arena_init");
}

// {vertex count, status bits, the build's sequence number} as ONE 64-bit word into pinned host memory (LnTable.host_counters):
// count | status << 32 | seq << 40.  A single aligned 8-byte store needs no release fence in front of it — the fence of the
// three-word form made the last workgroup of a build write back its whole L2 (buffer_wbl2) twice.
__device__ __forceinline__ void ln_report_to_host(int* host_counters, int count, int status, int seq) {
    if (!host_counters) return;
    const unsigned long long word = (unsigned long long)(unsigned int)count | ((unsigned long long)(status & 0xFF) << 32) |
                                    ((unsigned long long)(seq & 0xFFFFFF) << 40);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_counters), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------
// clear
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_table_clear(LnTable t, float* values, long long values_elems) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = g; i < t.capacity; i += stride) {
        t.slot_keys[i] = LN_EMPTY_KEY;
        t.slot_tok[i] = LN_EMPTY_TOK;
        t.slot_cnt[i] = 0;
        t.entries[i] = -1;
    }
    const long long nk = (long long)t.capacity * t.pos_dim;
    for (long long i = g; i < nk; i += stride) t.keys[i] = 0;
    if (values) {
        const long long n4 = values_elems >> 2;
        float4* v4 = reinterpret_cast<float4*>(values);
        for (long long i = g; i < n4; i += stride) v4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long i = (n4 << 2) + g; i < values_elems; i += stride) values[i] = 0.f;
    }
    if (g == 0) {
        *t.nr_filled = 0;
        *t.status = 0;
    }
}

extern "C" int ln_table_clear(const LnTable* t, float* values, long long values_elems, void* stream) {
    int rc = ln_check_table(t, "ln_table_clear");
    if (rc) return rc;
    LN_REQUIRE(values == nullptr || (reinterpret_cast<uintptr_t>(values) & 15) == 0, LN_ERR_ARG,
               "ln_table_clear: values must be 16-byte aligned");
    long long work = (long long)t->capacity * t->pos_dim;
    if (values && values_elems / 4 > work) work = values_elems / 4;
    int blocks = ln_div_up(work, 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    LN_LAUNCH("k_table_clear", k_table_clear, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *t, values, values_elems);
    return ln_check_launch("ln_table_clear");
}

// ------------------------------------------------------------------------------------------
// insert
// ------------------------------------------------------------------------------------------
// Returns the slot of `key` (inserting it if new) and, in `pos`, this token's rank among the tokens
// of the current build that landed on that slot (one returning int atomic: it is both the CSR
// position of the token and, summed, the slot's degree).
template <int D>
__device__ __forceinline__ int ln_insert(const LnTable& t, const int* key, int& pos) {
    pos = -1;
    if (!KeyPack<D>::in_range(key, t.key_format)) {
        atomicOr(t.status, LN_STATUS_KEY_RANGE);
        return -1;
    }
    const uint64_t pk = KeyPack<D>::pack(key, t.key_format);
    const LnProbe pr = LnProbe::of_key<D>(key, t, ln_bucket_slots(t.capacity));
    for (int probes = 0; probes < t.capacity; ++probes) {
        const int h = pr.slot(probes);
        // Plain (cacheable) pre-check: a slot only ever changes EMPTY -> key, so a stale read can only
        // show EMPTY, in which case the CAS below decides.  Duplicates of an already-inserted key are
        // then served by L1/L2 instead of each costing a memory-side atomic.
        unsigned long long cur = t.slot_keys[h];
        if (cur == LN_EMPTY_KEY) {
            cur = atomicCAS(&t.slot_keys[h], (unsigned long long)LN_EMPTY_KEY, (unsigned long long)pk);
            if (cur == LN_EMPTY_KEY) cur = pk;  // we claimed it
        }
        if (cur == pk) {
            pos = atomicAdd(&t.slot_cnt[h], 1);
            return h;
        }
    }
    atomicOr(t.status, LN_STATUS_TABLE_FULL);
    return -1;
}

// Token producer 1: the d+1 simplex vertices of every point (kernel_splat / distribute).
// One thread per TOKEN (point, remainder): the d+1 lanes of a point recompute the same cheap simplex
// arithmetic, but every hash probe chain runs in its own lane, which gives (d+1)x the memory-level
// parallelism of the reference's thread-per-point loop, and idx / w stores are lane-linear.
template <int D>
__global__ void __launch_bounds__(256)
    k_insert_points(LnTable t, const float* __restrict__ pos_raw, LnScale<D> sc, int n, int* __restrict__ tok_slot,
                    int* __restrict__ tok_pos, float* __restrict__ w, const float* __restrict__ vals, int val_dim,
                    float* __restrict__ distributed) {
    const long long tk = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = int(tk / (D + 1));
    const int r = int(tk - (long long)p * (D + 1));
    if (p >= n) return;
    float pr[D];
#pragma unroll
    for (int i = 0; i < D; ++i) pr[i] = pos_raw[(size_t)p * D + i];
    LnSimplex<D> s;
    ln_simplex<D>(pr, sc, s);
    ln_simplex_of_cloud<D>(s, p, t.batch_points, t.batch_key_step);
    int key[D];
    ln_vertex_key<D>(s, r, key);
    int pos;
    const int h = ln_insert<D>(t, key, pos);
    tok_slot[tk] = h;
    tok_pos[tk] = pos;
    float b = s.bary[0];
#pragma unroll
    for (int k = 1; k <= D; ++k) b = (r == k) ? s.bary[k] : b;
    if (w) w[tk] = h >= 0 ? b : -1.0f;
    if (distributed) {  // LatticeGPU.cuh:626-637: [pos_scaled(d) | val(V) | bary[r]] per simplex vertex
        const int row_len = D + val_dim + 1;
        float* o = distributed + (size_t)tk * row_len;
#pragma unroll
        for (int i = 0; i < D; ++i) o[i] = pr[i] / sc.sigma[i];
        for (int j = 0; j < val_dim; ++j) o[D + j] = vals[(size_t)p * val_dim + j];
        o[D + val_dim] = b;
    }
}

// ------------------------------------------------------------------------------------------
// Bucketed build of a freshly cleared table (the splat / distribute hot path)
// ------------------------------------------------------------------------------------------
// Memory-side atomics are the scarce resource on this part (every device-scope atomic executes at
// the memory controller), and the insert above spends one returning atomic per TOKEN.  Here the
// tokens are first partitioned by the bucket their hash lands in (two LDS-histogram passes; one
// global atomic per (block, bucket) instead of one per token); then ONE workgroup per bucket stages
// the bucket's slots in LDS and resolves everything there — claim (64-bit LDS CAS), per-slot token
// count and position (LDS add), smallest token (LDS min) — and emits the slot range, the slot CSR
// and its segments directly.  A bucket that fills up raises LN_STATUS_BUCKET_OVERFLOW: the caller
// re-runs the build with LN_BUILD_ATOMIC_PATH, whose inserts spill past the bucket.

// Pass 1: thread per point -> the d+1 packed keys and weights (+ distribute rows), scattered straight into the
// fixed-capacity region of the bucket each key hashes to.  Positions inside a region come from an LDS count per
// (block, bucket) plus ONE returning global atomic per (block, bucket) on the bucket's cursor.
// The kernel also does what HashTable::clear would (HT.cu:49-57) for everything the later passes do not overwrite
// anyway: values and keys zeroed, nr_filled = status = 0, and this build's first-occurrence bitmap.  The bucket
// cursors live in the first words of t.slot_cnt (zero between builds: the scan pass resets them), word nbk of
// it collects "a key did not fit the packed format".
// points per thread: 2 keeps the (block, bucket) global atomics low; wide lattices take 1 so that the staging arrays fit
// the 64 KB of static LDS
#ifndef LN_KEYS_PTS_SMALL_D
#define LN_KEYS_PTS_SMALL_D 2
#endif
#define LN_KEYS_PTS_PER_THREAD ((D) <= 3 ? LN_KEYS_PTS_SMALL_D : 1)
#define LN_KEYS_PTS_PER_BLOCK (256 * LN_KEYS_PTS_PER_THREAD)
// TH = 256: a thread owns LN_KEYS_PTS_PER_THREAD whole points (all d+1 tokens each).
// TH = 1024 (d <= 3): the SAME 512-point tile — i.e. the same number of (block, bucket) cursor atomics — on four times the waves: two
// threads per point, each owning half of its d+1 tokens (both evaluate the point's simplex: a few hundred ALU operations against a
// chain of returning LDS atomics and global round trips that four waves per SIMD overlap where one could not).
template <int D, int TH>
__global__ void __launch_bounds__(TH)
    k_point_keys(LnTable t, const float* __restrict__ pos_raw, LnScale<D> sc, int n, int sb, int nbk, int capb,
                 int* __restrict__ part_tok, unsigned long long* __restrict__ part_pk, int* __restrict__ tok_slot,
                 float* __restrict__ w, const float* __restrict__ vals, int val_dim, float* __restrict__ distributed,
                 int* __restrict__ seg_count, int seg_regions, float* __restrict__ clear_values, long long clear_values_elems,
                 unsigned int* __restrict__ pub) {
    constexpr bool SPLIT = TH == 1024;                        // two threads per point
    constexpr int PTS = TH == 256 ? LN_KEYS_PTS_PER_THREAD : 1;   // points a thread touches (TH = 512: one whole point per thread)
    constexpr int RH = SPLIT ? (D + 2) / 2 : D + 1;           // tokens of a point a thread owns
    constexpr int PTS_PER_BLOCK = SPLIT ? 512 : TH * PTS;
    __shared__ int s_cnt[LN_BKT_MAX];
    __shared__ int s_lbase[LN_BKT_MAX];
    __shared__ int s_scan_tmp[TH / 64];
    __shared__ unsigned long long s_stage_pk[PTS_PER_BLOCK * (D + 1)];
    __shared__ int s_stage_tok[PTS_PER_BLOCK * (D + 1)];
    __shared__ int s_stage_dst[PTS_PER_BLOCK * (D + 1)];
    int* cursor = t.slot_cnt;
    LN_STAMP(0);
    for (int b = threadIdx.x; b < nbk; b += TH) s_cnt[b] = 0;
    // space-ordered slots: the table's slot map in registers (25 wave-uniform words: scalar loads), walked d+1 times per point
    const LnSlotMap smap = ln_load_slot_map(t.slot_map);
    ln_lds_barrier();
    LN_STAMP(1);
    unsigned long long pk[PTS][RH];
    int bkt[PTS][RH];
    int rank[PTS][RH];
    bool bad_key = false;
    const int r_first = SPLIT ? (threadIdx.x & 1) * RH : 0;
#pragma unroll
    for (int it = 0; it < PTS; ++it) {
        const int p = SPLIT ? blockIdx.x * PTS_PER_BLOCK + (threadIdx.x >> 1) : blockIdx.x * PTS_PER_BLOCK + it * TH + threadIdx.x;
#pragma unroll
        for (int k = 0; k < RH; ++k) bkt[it][k] = -1;
        if (p >= n) continue;
        float pr[D];
#pragma unroll
        for (int i = 0; i < D; ++i) pr[i] = pos_raw[(size_t)p * D + i];
        LnSimplex<D> s;
        ln_simplex<D>(pr, sc, s);
        ln_simplex_of_cloud<D>(s, p, t.batch_points, t.batch_key_step);
#pragma unroll
        for (int k = 0; k < RH; ++k) {
            const int r = r_first + k;
            if (SPLIT && r > D) continue;  // (odd d + 1: the second thread of a point owns one token less)
            int key[D];
            ln_vertex_key<D>(s, r, key);
            float bary = s.bary[0];
#pragma unroll
            for (int q = 1; q <= D; ++q) bary = (r == q) ? s.bary[q] : bary;
            const size_t tk = (size_t)p * (D + 1) + r;
            const bool lat_fmt = t.key_format == LN_KEYS_LATTICE;  // (vertex r of a simplex has remainder r: no modulo needed)
            const bool ok = lat_fmt ? KeyPack<D>::lattice_in_range(key, r) : KeyPack<D>::in_range(key, t.key_format);
            if (ok) {
                pk[it][k] = lat_fmt ? KeyPack<D>::lattice_pack(key, r) : KeyPack<D>::pack(key, t.key_format);
                bkt[it][k] = LnProbe::bucket_of_key<D>(key, t.capacity, sb, smap);
                rank[it][k] = atomicAdd(&s_cnt[bkt[it][k]], 1);
            } else {
                bad_key = true;
                if (tok_slot) tok_slot[tk] = -1;
            }
            if (w) w[tk] = ok ? bary : -1.0f;
            if (distributed) {  // LatticeGPU.cuh:626-637: [pos_scaled(d) | val(V) | bary[r]] per simplex vertex
                const int row_len = D + val_dim + 1;
                float* o = distributed + tk * row_len;
#pragma unroll
                for (int i = 0; i < D; ++i) o[i] = pr[i] / sc.sigma[i];
                for (int j = 0; j < val_dim; ++j) o[D + j] = vals[(size_t)p * val_dim + j];
                o[D + val_dim] = bary;
            }
        }
    }
    if (bad_key) atomicOr(&t.slot_cnt[nbk], 1);
    LN_STAMP(2);
    ln_lds_barrier();
    LN_STAMP(3);
    // Region cursors (one returning global atomic per (block, bucket), all of a thread row in flight together) and, for the
    // staged write below, the exclusive prefix of this block's per-bucket counts (one TH-wide scan per row of buckets).
    int block_tokens = 0;
    for (int k0 = 0; k0 < nbk; k0 += TH) {  // block-uniform trip count
        const int b = k0 + threadIdx.x;
        const int c = b < nbk ? s_cnt[b] : 0;
        const int at = c ? atomicAdd(&cursor[b], c) : 0;  // in flight across the scan below
        int row_total;
        const int ex = ln_block_excl_scan<TH / 64>(c, s_scan_tmp, &row_total);  // (has its own barriers)
        if (b < nbk) {
            s_lbase[b] = block_tokens + ex;
            if (c) s_cnt[b] = at;
        }
        block_tokens += row_total;
    }
    LN_STAMP(4);
    ln_lds_barrier();
    LN_STAMP(5);
    // Stage the block's tokens in LDS sorted by bucket, then write them out in that order: the lanes of a wave then cover a
    // few runs of consecutive region entries instead of 64 different cache lines per store instruction (the scattered form
    // of these two stores was a third of this kernel's time).
#pragma unroll
    for (int it = 0; it < PTS; ++it) {
        const int p = SPLIT ? blockIdx.x * PTS_PER_BLOCK + (threadIdx.x >> 1) : blockIdx.x * PTS_PER_BLOCK + it * TH + threadIdx.x;
#pragma unroll
        for (int k = 0; k < RH; ++k) {
            if (bkt[it][k] < 0) continue;
            const int tk = p * (D + 1) + r_first + k;
            const int at = s_cnt[bkt[it][k]] + rank[it][k];
            const int j = s_lbase[bkt[it][k]] + rank[it][k];
            s_stage_tok[j] = tk;
            s_stage_pk[j] = pk[it][k];
            s_stage_dst[j] = at < capb ? bkt[it][k] * capb + at : -1;  // (regions are < 2^31 entries in total: checked by the host)
        }
    }
    ln_lds_barrier();
    LN_STAMP(6);
    for (int j = threadIdx.x; j < block_tokens; j += TH) {
        const int dst = s_stage_dst[j];
        const int tk = s_stage_tok[j];
        if (dst >= 0) {
            part_tok[dst] = tk;
            part_pk[dst] = s_stage_pk[j];
        } else if (tok_slot) {
            tok_slot[tk] = -1;  // region full (heavily skewed cloud): the bucket workgroup reports it, the build is replayed
        }
    }
    {  // Clear duties, LAST: nothing in this kernel reads what they write, and a workgroup barrier waits for the stores issued
       // before it (1.6 us at C3 when they came first); here only the end of the kernel does.
        const long long stride = (long long)gridDim.x * TH;
        const long long g = (long long)blockIdx.x * TH + threadIdx.x;
#ifdef LN_PROBE_NO_CLEAR  // timing probe (wrong results): the values are not cleared
        clear_values = nullptr;
#endif
        if (clear_values) {
            const long long n4 = clear_values_elems >> 2;
            float4* v4 = reinterpret_cast<float4*>(clear_values);
            for (long long i = g; i < n4; i += stride) v4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (long long i = (n4 << 2) + g; i < clear_values_elems; i += stride) clear_values[i] = 0.f;
        }
        const long long nk = (long long)t.capacity * D;
        for (long long i = g; i < nk; i += stride) t.keys[i] = 0;
        for (long long i = g; i <= nbk; i += stride) pub[i] = 0u;  // the bucket workgroups' look-back words + their ticket counter (k_bucket_rows)
        if (g == 0) {
            *t.nr_filled = 0;
            *t.status = 0;   // later passes (bucket build, scan) raise the error bits of this build
            for (int gi = 0; gi < LN_XCD_GROUPS; ++gi) seg_count[gi] = 0;  // the bucket workgroups of the next launch add to them
            seg_count[LN_XCD_GROUPS] = seg_regions;
            seg_count[LN_XCD_GROUPS + 1] = 1;  // the segment descriptors of this build carry rows (k_bucket_rows)
        }
    }
    LN_STAMP(7);
}

// Pass 2: one workgroup per bucket.  LDS: keys[sb] | count[sb] | min token[sb] | token offset[sb] | segment offset[sb] | row[sb] | rank[sb] | compacted minima.
// Its barriers order LDS traffic only (ln_lds_barrier) wherever nothing else is needed: __syncthreads() also waits for every global
// operation a wave has in flight — the agent-scope store that publishes the bucket's count (~1.5 us to the memory side) held the
// whole workgroup at the next barrier, the slot stores of the emit phase at the last one.
// This is the LAST kernel of a bucketed build.  Rows are numbered in SLOT order: row of a new vertex = (new vertices of all
// buckets before this one) + (occupied slots before it inside the bucket).  The only cross-workgroup quantity is that per-
// bucket count, handed on through `pub` (one word per bucket: ready bit | error bit | count) with relaxed agent-scope
// atomics — no fence (the word IS the payload), published right after the bucket's scans and read just before the emit
// phase (decoupled look-back over all earlier buckets; buckets are handed out as tickets, so a workgroup only ever waits for
// workgroups that are resident or done).  The last bucket closes the build: nr_filled, status, the pinned host
// pair, and the bucket cursors back to zero (every workgroup read them before it published).
// The reference numbers vertices by thread-arrival order (atomicAdd(m_nr_filled), HashTableGPU.cuh:454: not reproducible);
// ln_canonicalize relabels a table built here into first-occurrence order (= a serial run of the reference) on request.
#ifndef LN_BKT_THREADS
#define LN_BKT_THREADS 1024
#endif
#define LN_BKT_WAVES (LN_BKT_THREADS / 64)
#ifndef LN_BKT_NARROW_TOKENS
#define LN_BKT_NARROW_TOKENS 4096  // tokens per bucket up to which overlapping builds take 512-thread bucket workgroups
#endif
// ln_build_concurrency: how many builds / scans the caller keeps in flight on this GPU (thread-local, 1 = one at a time)
static thread_local int g_ln_build_concurrency = 1;
extern "C" int ln_build_concurrency(int scans_in_flight) {
    g_ln_build_concurrency = scans_in_flight > 1 ? scans_in_flight : 1;
    return LN_OK;
}
#define LN_BKT_LDS_LIMIT (150 * 1024)  // dynamic LDS one k_bucket_rows workgroup may ask for (160 KB per CU minus its static arrays)
#define LN_BKT_LDS_PER_SLOT (sizeof(unsigned long long) + 7 * sizeof(int))
#define LN_BKT_LDS_EXTRA 32  // alignment of the compacted token list + its padding to a multiple of four entries
#ifndef LN_BKT_REG_TOK
#define LN_BKT_REG_TOK 4  // tokens per thread whose (token, slot, position) stay in registers between the two sweeps
#endif
#define LN_PUB_READY 0x80000000u
#define LN_PUB_ERR 0x40000000u
#define LN_PUB_CNT 0x3FFFFFFFu
template <int D, int TH>
__global__ void __launch_bounds__(TH)
    k_bucket_rows(LnTable t, int sb, int nbk, int capb, int* __restrict__ cursor, const int* __restrict__ part_tok,
                  const unsigned long long* __restrict__ part_pk, int* __restrict__ part_slot, int* __restrict__ part_pos,
                  int* __restrict__ idx_out, LnCsr csr, unsigned int* __restrict__ pub) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
    unsigned long long* skeys = s_mem;
    int* scnt = reinterpret_cast<int*>(skeys + sb);
    unsigned int* smin = reinterpret_cast<unsigned int*>(scnt + sb);
    int* soff = reinterpret_cast<int*>(smin + sb);
    int* sseg = soff + sb;
    int* srow = sseg + sb;
    // smallest tokens of the occupied slots, compacted (16-byte aligned: read as uint4; LN_BKT_LDS_EXTRA covers the alignment and the padding)
    int* srank = srow + sb;  // in-bucket rank of the compacted vertices (partial counts added up)
    // (an OFFSET from the 16-byte aligned base, not an address rounded up through an integer cast: the compiler then keeps the LDS
    // address space — the cast turned every access into a FLAT instruction, whose s_waitcnt vmcnt(0) waited ~1.5 us for the
    // agent-scope store that publishes the bucket's count)
    unsigned int* slist = reinterpret_cast<unsigned int*>(s_mem) + ((8 * sb + 3) & ~3);  // 2 sb words of keys + six int arrays
    __shared__ int s_wave_tok[16], s_wave_seg[16], s_wave_new[16];
    __shared__ int s_run_tok, s_run_seg, s_run_new, s_err;
    __shared__ int s_rcnt[LN_XCD_GROUPS], s_rbase[LN_XCD_GROUPS];
    __shared__ int s_ticket;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // The bucket is NOT blockIdx.x: it is a ticket taken when the workgroup starts running.  Whoever holds ticket b knows that
    // the holders of all smaller tickets are resident (or done), whatever order workgroups are dispatched in and however other
    // kernels share the CUs — with blockIdx.x as the bucket, three of these kernels running side by side on different queues
    // can fill one XCD each with workgroups that wait for a bucket whose workgroup has nowhere to go (measured: a 0.6 s stall).
    if (tid == 0) s_ticket = atomicAdd(reinterpret_cast<int*>(&pub[nbk]), 1);
    LN_STAMP(8);
    for (int i = tid; i < sb; i += TH) {  // (while the ticket is in flight)
        skeys[i] = LN_EMPTY_KEY;  // the table was cleared by this build call
        scnt[i] = 0;
        smin[i] = LN_EMPTY_TOK;
    }
    ln_lds_barrier();
    const int b = s_ticket;
    // slot range of this bucket: b * sb for a hashed table; under a slot map the leaf's run cut into the leaf's own bucket size
    // (`sb` is then the LARGEST bucket of the table: what the LDS arrays above are carved for)
    const bool mapped = t.slot_map != nullptr;
    const LnBucket bk(b, t.capacity, sb, t.slot_map);
    const int lo = bk.lo;
    const int size = bk.size;
    // CSR offset of this bucket = tokens of all buckets before it.  The cursors of the earlier buckets and the bucket's own are
    // fetched together (one round trip), the register-resident tokens behind them (fetching those unconditionally, 4 x 1024 entries
    // whatever the cursor says, saved the dependency and cost more in reads: 20.2 -> 21.3 us).
    const size_t in0 = (size_t)b * capb;
    int before = 0;
    for (int i = tid; i < b; i += TH) before += min(cursor[i], capb);
    const int my_cursor = cursor[b];
    const int ntok = min(my_cursor, capb);
    int r_tk[LN_BKT_REG_TOK], r_ls[LN_BKT_REG_TOK], r_pos[LN_BKT_REG_TOK];
    unsigned long long r_pk[LN_BKT_REG_TOK];
#pragma unroll
    for (int k = 0; k < LN_BKT_REG_TOK; ++k) {
        const int j = tid + k * TH;
        r_ls[k] = -1;
        r_pos[k] = -1;
        r_tk[k] = j < ntok ? part_tok[in0 + j] : -1;
        r_pk[k] = j < ntok ? part_pk[in0 + j] : LN_EMPTY_KEY;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) s_wave_tok[wave] = before;
    if (tid == 0) {
        s_run_tok = 0;
        s_run_seg = 0;
        s_run_new = 0;
        s_err = my_cursor > capb ? 1 : 0;  // region overflow: tokens were dropped by pass 1
    }
    if (tid < LN_XCD_GROUPS) s_rcnt[tid] = 0;
    __syncthreads();  // (every cursor[] load of this workgroup has returned by now: the last bucket relies on it)
    int base = 0;
#pragma unroll
    for (int k = 0; k < (TH / 64); ++k) base += s_wave_tok[k];
    ln_lds_barrier();  // s_wave_tok is reused by the scans below
    LN_STAMP(9);

    // Called by every lane of the wave (invalid lanes carry no token): the wave-level grouping below uses shuffles.
    auto place = [&](bool valid, int tk, unsigned long long pk, int& ls, int& pos) {
        ls = -1;
        pos = -1;
        if (valid) {
            int key[D];
            KeyPack<D>::unpack(pk, key, t.key_format);
            int o = bk.offset_of<D>(key, t.capacity);
            for (int i = 0; i < size; ++i) {
                unsigned long long cur = skeys[o];
                if (cur == LN_EMPTY_KEY) {
                    cur = atomicCAS(&skeys[o], (unsigned long long)LN_EMPTY_KEY, pk);
                    if (cur == LN_EMPTY_KEY) cur = pk;
                }
                if (cur == pk) {
                    ls = o;
                    break;
                }
                if (++o >= size) o = 0;
            }
            if (ls < 0) s_err = 1;  // every slot of the bucket is taken (benign race: all writers store 1)
        }
        // Hot vertices (coarse lattices: thousands of tokens on one slot) would serialise on one LDS address.  When a
        // large part of the wave landed on the slot of its first placed lane, that group takes its positions with ONE
        // add and ONE min (ballot + rank among the matching lanes); everybody else issues its own pair.
        const unsigned long long placed_mask = __ballot(ls >= 0);
        const int lead = placed_mask ? __ffsll((long long)placed_mask) - 1 : 0;
        const int lead_ls = __shfl(ls, lead, 64);
        const unsigned long long match = __ballot(ls >= 0 && ls == lead_ls);
        const bool grouped = __popcll(match) >= 16 && ls >= 0 && ls == lead_ls;
        if (__popcll(match) >= 16) {  // wave-uniform
            unsigned int mn = grouped ? (unsigned int)tk : 0xFFFFFFFFu;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) mn = min(mn, (unsigned int)__shfl_xor((int)mn, off, 64));
            int base_pos = 0;
            if ((threadIdx.x & 63) == lead) {
                base_pos = atomicAdd(&scnt[lead_ls], __popcll(match));
                atomicMin(&smin[lead_ls], mn);
            }
            base_pos = __shfl(base_pos, lead, 64);
            if (grouped) pos = base_pos + __popcll(match & ((1ull << (threadIdx.x & 63)) - 1ull));
        }
        if (ls >= 0 && !grouped) {
            pos = atomicAdd(&scnt[ls], 1);
            atomicMin(&smin[ls], (unsigned int)tk);
        }
    };
#pragma unroll
    for (int k = 0; k < LN_BKT_REG_TOK; ++k)
        if (__ballot(r_tk[k] >= 0)) place(r_tk[k] >= 0, r_tk[k], r_pk[k], r_ls[k], r_pos[k]);  // (wave-uniform: the votes inside need whole waves)
    for (int j0 = LN_BKT_REG_TOK * TH; j0 < ntok; j0 += TH) {
        const int j = j0 + tid;
        const bool valid = j < ntok;
        int ls, pos;
        place(valid, valid ? part_tok[in0 + j] : -1, valid ? part_pk[in0 + j] : LN_EMPTY_KEY, ls, pos);
        if (valid) {
            part_slot[in0 + j] = ls;
            part_pos[in0 + j] = pos;
        }
    }
    ln_lds_barrier();
    LN_STAMP(10);
    // exclusive scans of the per-slot token counts, segment counts and occupancy (-> row of the slot inside the bucket)
    for (int start = 0; start < size; start += TH) {
        const int i = start + tid;
        const int c = (i < size) ? scnt[i] : 0;
        const int g = (c + LN_CSR_SEG - 1) / LN_CSR_SEG;
        const int nw = c ? 1 : 0;
        const int ic = ln_wave_incl_scan(c), ig = ln_wave_incl_scan(g);
        const unsigned long long occ = __ballot(nw);
        const int in_wave_new = __popcll(occ & ((1ull << lane) - 1ull));
        if (lane == 63) {
            s_wave_tok[wave] = ic;
            s_wave_seg[wave] = ig;
            s_wave_new[wave] = __popcll(occ);
        }
        ln_lds_barrier();
        int wt = 0, wg = 0, wn = 0;
#pragma unroll
        for (int k = 0; k < (TH / 64); ++k) {  // all LDS reads issue together (a loop to `wave` waits for each in turn)
            const int a = s_wave_tok[k], g2 = s_wave_seg[k], n2 = s_wave_new[k];
            if (k < wave) {
                wt += a;
                wg += g2;
                wn += n2;
            }
        }
        // (a bucket of at most one trip starts from zero without reading the running totals: no barrier needed before the last
        // thread overwrites them)
        const bool multi = size > TH;
        const int rt = multi ? s_run_tok : 0, rg = multi ? s_run_seg : 0, rn = multi ? s_run_new : 0;
        if (i < size) {
            soff[i] = rt + wt + ic - c;
            sseg[i] = rg + wg + ig - g;
            srow[i] = rn + wn + in_wave_new;
        }
        if (multi) ln_lds_barrier();  // (every thread has read the running totals of this trip)
        if (tid == TH - 1) {
            s_run_tok = rt + wt + ic;
            s_run_seg = rg + wg + ig;
            s_run_new = rn + wn + __popcll(occ);
        }
        ln_lds_barrier();
    }
    LN_STAMP(16);
    // hand the count on as early as it is known
    if (tid == 0)
        __hip_atomic_store(&pub[b], LN_PUB_READY | (s_err ? LN_PUB_ERR : 0u) | (unsigned int)s_run_new, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    // Which slot of the bucket a key ends up in depends on the order the LDS CAS race resolves collisions in, so rows are NOT
    // numbered by slot position: inside the bucket a vertex is numbered by the rank of its smallest token (first occurrence in
    // (point, remainder) order among the bucket's vertices) — a pure function of the cloud, identical run to run.
    {
        const int nv = s_run_new;
        for (int i = tid; i < size; i += TH)
            if (scnt[i]) slist[srow[i]] = smin[i];
        for (int j = nv + tid; j < ((nv + 3) & ~3); j += TH) slist[j] = 0xFFFFFFFFu;  // pad to a multiple of 4
        for (int j = tid; j < nv; j += TH) srank[j] = 0;
        ln_lds_barrier();
        LN_STAMP(17);
        const uint4* l4 = reinterpret_cast<const uint4*>(slist);
        const int nv4 = (nv + 3) / 4;
        // Rank of every vertex's smallest token among the bucket's nv minima (all pairs, ~180 x 180 at C3).  The threads form a
        // (vertex, chunk) grid over the COMPACTED list: chunks = threads / nv, every thread counts over its share of the list with
        // four loads in flight (the lanes of a wave read the same words: LDS broadcast) and adds its partial count to its vertex
        // with one LDS add.  (Two lanes per SLOT, occupied or not, each walking half the list one load at a time: 2.3 us of the pass.)
        const int chunks = max(1, min(TH / max(nv, 1), nv4));
        const int vpp = TH / chunks;      // vertices per trip
        const int per = (nv4 + chunks - 1) / chunks;  // 16-byte words per chunk
        for (int v0 = 0; v0 < nv; v0 += vpp) {        // one trip unless the bucket holds more vertices than the workgroup has threads
            const int v = v0 + tid % vpp, ch = tid / vpp;
            if (v < nv && ch < chunks) {
                const unsigned int mine = slist[v];
                const int j0 = ch * per, j1 = min(nv4, j0 + per);
                int r = 0;
                for (int j = j0; j < j1; j += 4) {
                    uint4 t4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) t4[k] = l4[min(j + k, nv4 - 1)];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (j + k < j1) r += (t4[k].x < mine) + (t4[k].y < mine) + (t4[k].z < mine) + (t4[k].w < mine);
                }
                if (r) atomicAdd(&srank[v], r);
            }
        }
        ln_lds_barrier();
        for (int i = tid; i < size; i += TH)
            if (scnt[i]) srow[i] = srank[srow[i]];  // (srow[i] held the slot's position in the compacted list)
    }
    LN_STAMP(11);
    // Segment ids.  Without region planes: one contiguous run of region 0 per bucket.  Space-ordered table (LnTable.planes): one
    // contiguous run of the bucket's region.  Region planes over a hashed table: every slot files its segments under the kd region
    // of its key (sseg[i] becomes region << 28 | position among this bucket's segments of that region; one returning global atomic
    // per (bucket, region)).
    const bool slot_ordered = mapped;
    const int bucket_region = (slot_ordered && csr.planes) ? ln_region_of_bucket(b, nbk) : 0;
    const int* planes = slot_ordered ? nullptr : csr.planes;
    if (planes) {
        for (int i = tid; i < size; i += TH) {
            const int c = scnt[i];
            if (c) {
                int key[D];
                KeyPack<D>::unpack(skeys[i], key, t.key_format);
                const int r = ln_region_of_key<D>(key, planes);
                sseg[i] = (r << 28) | atomicAdd(&s_rcnt[r], (c + LN_CSR_SEG - 1) / LN_CSR_SEG);
            }
        }
        ln_lds_barrier();
    }
    // first segment id of this bucket (per region): one returning global atomic, left in flight across the look-back below
    int seg_base_reg = 0;
    if (planes) {
        if (tid < LN_XCD_GROUPS && s_rcnt[tid]) seg_base_reg = atomicAdd(&csr.seg_count[tid], s_rcnt[tid]);
    } else if (tid == 0 && s_run_seg) {
        seg_base_reg = atomicAdd(&csr.seg_count[bucket_region], s_run_seg);
    }
    LN_STAMP(15);
    // look-back: new vertices (and error flags) of every earlier bucket
    int rows_before = 0;
    unsigned int err_before = 0u;
    for (int i = tid; i < b; i += TH) {
        unsigned int v = __hip_atomic_load(&pub[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (!(v & LN_PUB_READY)) {
            __builtin_amdgcn_s_sleep(2);
            v = __hip_atomic_load(&pub[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        rows_before += int(v & LN_PUB_CNT);
        err_before |= v & LN_PUB_ERR;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        rows_before += __shfl_xor(rows_before, off, 64);
        err_before |= (unsigned int)__shfl_xor((int)err_before, off, 64);
    }
    if (tid < (planes ? LN_XCD_GROUPS : 1)) s_rbase[tid] = seg_base_reg;
    if (lane == 0) {
        s_wave_tok[wave] = rows_before;
        s_wave_seg[wave] = int(err_before);
    }
    ln_lds_barrier();
    int base_row = 0;
    unsigned int err_all = s_err ? LN_PUB_ERR : 0u;
#pragma unroll
    for (int k = 0; k < (TH / 64); ++k) {
        base_row += s_wave_tok[k];
        err_all |= (unsigned int)s_wave_seg[k];
    }
    const int placed = s_run_tok;
    const int new_here = s_run_new;
    if (t.row_regions && tid == 0 && slot_ordered) {  // first row of each kd region (the argument of ln_conv_row_partition)
        const int bpr = nbk / LN_XCD_GROUPS;
        if (b % bpr == 0) t.row_regions[b / bpr] = base_row;
        if (b == nbk - 1) t.row_regions[LN_XCD_GROUPS] = base_row + new_here;
    }
    LN_STAMP(12);
    for (int i = tid; i < size; i += TH) {
        const int h = lo + i;
        const int beg = base + soff[i];
        const int c = scnt[i];
        int row = -1;
        if (c) {
            row = base_row + srow[i];
            if (t.row_limit > 0 && row >= t.row_limit) row = -1;  // beyond the host's static row bound: stays un-inserted
        }
        srow[i] = row;
        csr.grp_start[h] = beg;
        const unsigned long long pk = skeys[i];
        t.slot_keys[h] = pk;
        t.entries[h] = row;
        t.slot_tok[h] = smin[i];  // smallest token of the slot: what ln_canonicalize orders the rows by
        if (row >= 0) {
            int key[D];
            KeyPack<D>::unpack(pk, key, t.key_format);
#pragma unroll
            for (int k = 0; k < D; ++k) t.keys[(size_t)row * D + k] = key[k];
        }
        const int sr = planes ? (sseg[i] >> 28) : bucket_region;
        long long sid = (long long)sr * csr.seg_region + s_rbase[planes ? sr : 0] + (planes ? (sseg[i] & 0x0FFFFFFF) : sseg[i]);
        for (int e = 0; e < c; e += LN_CSR_SEG, ++sid)
            reinterpret_cast<int4*>(csr.seg_desc)[sid] = make_int4(row, beg + e, c - e, e);  // {ROW, first entry, entries to the end, offset}
    }
    if (b == nbk - 1) {  // close the build
        // (a slot map may leave a few slots behind its last run: they stay empty, and every slot array says so)
        for (int h = lo + size + tid; h < t.capacity; h += TH) {
            csr.grp_start[h] = base + ntok;
            t.slot_keys[h] = LN_EMPTY_KEY;
            t.entries[h] = -1;
            t.slot_tok[h] = LN_EMPTY_TOK;
        }
        if (tid == 0) {
            csr.grp_start[t.capacity] = base + ntok;
            const int total = base_row + new_here;
            int bits = (err_all ? LN_STATUS_BUCKET_OVERFLOW : 0) | (cursor[nbk] ? LN_STATUS_KEY_RANGE : 0);
            *t.nr_filled = total;
            if (bits) atomicOr(t.status, bits);
            ln_report_to_host(t.host_counters, total, bits, t.host_seq);
        }
        __syncthreads();  // (tid 0 has read cursor[nbk])
        for (int i = tid; i <= nbk; i += TH) cursor[i] = 0;  // all-zero between builds; every reader has published
    }
    ln_lds_barrier();  // srow[] now holds the final row of every slot
    LN_STAMP(13);
#pragma unroll
    for (int k = 0; k < LN_BKT_REG_TOK; ++k) {
        if (r_tk[k] < 0) continue;
        if (r_ls[k] >= 0) csr.csr_tok[base + soff[r_ls[k]] + r_pos[k]] = r_tk[k];
        if (idx_out) idx_out[r_tk[k]] = r_ls[k] >= 0 ? srow[r_ls[k]] : -1;
    }
    for (int j = tid + LN_BKT_REG_TOK * TH; j < ntok; j += TH) {
        const int ls = part_slot[in0 + j];
        const int tk = part_tok[in0 + j];
        if (ls >= 0) csr.csr_tok[base + soff[ls] + part_pos[in0 + j]] = tk;
        if (idx_out) idx_out[tk] = ls >= 0 ? srow[ls] : -1;
    }
    for (int j = placed + tid; j < ntok; j += TH)
        csr.csr_tok[base + j] = -1;  // only after an overflow: keeps readers in bounds until the rebuild
    LN_STAMP(14);
}

// Token producer 2: coarsen kernel (LatticeGPU.cuh:2348-2511): per fine vertex with all-even key,
// token 0 = key/2, token 1+2a / 2+2a = the coarse neighbour matching an EXISTING fine np / nm.
template <int D>
__global__ void __launch_bounds__(256)
    k_insert_coarse(LnTable fine, int fine_rows_upper, LnTable coarse, int* __restrict__ tok_slot, int* __restrict__ tok_pos) {
    constexpr int TPR = 2 * (D + 1) + 1;  // tokens per fine row
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int m = *fine.nr_filled;
    if (m > fine_rows_upper) m = fine_rows_upper;
    if (r >= fine_rows_upper) return;
    int* ts = tok_slot + (size_t)r * TPR;
    int* tp = tok_pos + (size_t)r * TPR;
    for (int j = 0; j < TPR; ++j) {
        ts[j] = -1;
        tp[j] = -1;
    }
    if (r >= m) return;
    int fk[D + 1];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        fk[i] = fine.keys[(size_t)r * D + i];
        sum += fk[i];
    }
    fk[D] = -sum;
    bool all_even = true;
#pragma unroll
    for (int i = 0; i <= D; ++i) all_even = all_even && ((fk[i] & 1) == 0);  // |frac(key/2)| <= 0.1, LatticeGPU.cuh:2376
    if (!all_even) return;
    int div[D + 1];
#pragma unroll
    for (int i = 0; i <= D; ++i) div[i] = fk[i] / 2;  // exact: all even
    ts[0] = ln_insert<D>(coarse, div, tp[0]);
#pragma unroll
    for (int axis = 0; axis <= D; ++axis) {
        int nk[D + 1];
        int ck[D + 1];
        // np: +1 everywhere, -D on the axis
#pragma unroll
        for (int i = 0; i <= D; ++i) {
            nk[i] = fk[i] + 1;
            ck[i] = div[i] + 1;
        }
        nk[axis] = fk[axis] - D;
        ck[axis] = div[axis] - D;
        if (ln_retrieve<D>(fine, nk) >= 0) ts[1 + 2 * axis] = ln_insert<D>(coarse, ck, tp[1 + 2 * axis]);
        // nm: -1 everywhere, +D on the axis
#pragma unroll
        for (int i = 0; i <= D; ++i) {
            nk[i] = fk[i] - 1;
            ck[i] = div[i] - 1;
        }
        nk[axis] = fk[axis] + D;
        ck[axis] = div[axis] + D;
        if (ln_retrieve<D>(fine, nk) >= 0) ts[2 + 2 * axis] = ln_insert<D>(coarse, ck, tp[2 + 2 * axis]);
    }
}

// ------------------------------------------------------------------------------------------
// segment minimum -> smallest token per slot (defines the canonical row order)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_seg_min(LnTable t, const int* __restrict__ csr_tok, const int4* __restrict__ seg_desc, const int* __restrict__ seg_count) {
    const int sid = blockIdx.x * blockDim.x + threadIdx.x;
    if (sid >= *seg_count) return;  // (the atomic build path files every segment under region 0)
    const int4 d = seg_desc[sid];  // {slot, first entry, entries to the end of the slot, offset inside the slot}
    const int h = d.x;
    const int beg = d.y;
    const int end = beg + min(LN_CSR_SEG, d.z);
    unsigned int mn = LN_EMPTY_TOK;
    for (int e = beg; e < end; ++e) mn = min(mn, (unsigned int)csr_tok[e]);
    t.slot_cnt[h] = 0;  // the counts were consumed by the scan: keep the invariant "all zero between builds"
    if (d.w == 0 && d.z <= LN_CSR_SEG)
        t.slot_tok[h] = mn;  // the slot's only segment
    else
        atomicMin(&t.slot_tok[h], mn);
}

// ------------------------------------------------------------------------------------------
// mark / scan / finalize (shared by both producers)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_mark_first(LnTable t, const int* __restrict__ tok_slot, long long tokens, unsigned long long* __restrict__ bitmap,
                 int* __restrict__ block_cnt) {
    __shared__ int s_cnt[4];
    const long long tk = (long long)blockIdx.x * 256 + threadIdx.x;
    bool first = false;
    if (tk < tokens) {
        const int h = tok_slot[tk];
        if (h >= 0) first = (t.slot_tok[h] == (unsigned int)tk) && (t.entries[h] < 0);
    }
    const unsigned long long mask = __ballot(first);
    const int lane = threadIdx.x & 63;
    if (lane == 0) {
        bitmap[tk >> 6] = mask;
        s_cnt[threadIdx.x >> 6] = __popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// single workgroup: exclusive scan of the per-block first-occurrence counts (+ rows that existed before)
__global__ void __launch_bounds__(1024) k_scan_blocks(const int* __restrict__ block_cnt, const unsigned long long* __restrict__ bitmap, int nb,
                                                      int* __restrict__ block_prefix, int* nr_filled, int* __restrict__ status,
                                                      int* __restrict__ host_counters, int host_seq, int relabel) {
    __shared__ int s_wave[16];
    __shared__ int s_running;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int base = relabel ? 0 : *nr_filled;  // relabel (ln_canonicalize): ranks among the rows that exist, counters untouched
    if (tid == 0) s_running = 0;
    __syncthreads();
    for (int start = 0; start < nb; start += 1024) {
        const int i = start + tid;
        int v = 0;
        if (i < nb) {
            if (block_cnt) {
                v = block_cnt[i];
            } else {  // ln_canonicalize: the first-occurrence bits were set directly, count them here
                const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(bitmap + (size_t)i * 4);
                const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(bitmap + (size_t)i * 4 + 2);
                v = __popcll(a.x) + __popcll(a.y) + __popcll(c.x) + __popcll(c.y);
            }
        }
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int k = 0; k < wave; ++k) wave_off += s_wave[k];
        const int running = s_running;
        if (i < nb) block_prefix[i] = base + running + wave_off + incl - v;
        __syncthreads();
        if (tid == 1023) s_running = running + wave_off + incl;
        __syncthreads();
    }
    if (tid == 0 && !relabel) {
        *nr_filled = base + s_running;
        ln_report_to_host(host_counters, base + s_running, *status, host_seq);  // (every producer kernel ran before this one)
    }
}

// rank of token `ft` among the marked tokens: prefix of its 256-token block + the set bits below it inside the block (the
// block's four bitmap words are fetched together with the prefix: one round trip)
__device__ __forceinline__ int ln_rank_of_token(unsigned int ft, const unsigned long long* __restrict__ bitmap,
                                                const int* __restrict__ block_prefix) {
    const unsigned int blk = ft >> 8;
    const unsigned int wd = (ft >> 6) & 3;
    int r = block_prefix[blk];
    const ulonglong2 w01 = *reinterpret_cast<const ulonglong2*>(bitmap + (size_t)blk * 4);
    const ulonglong2 w23 = *reinterpret_cast<const ulonglong2*>(bitmap + (size_t)blk * 4 + 2);
    const unsigned long long words[4] = {w01.x, w01.y, w23.x, w23.y};
    unsigned long long mine = 0ull;
#pragma unroll
    for (unsigned int k = 0; k < 4; ++k) {
        if (k < wd) r += __popcll(words[k]);
        if (k == wd) mine = words[k];
    }
    return r + __popcll(mine & ((1ull << (ft & 63)) - 1ull));
}

template <int D>
__global__ void __launch_bounds__(256)
    k_finalize(LnTable t, const int* tok_slot, int* idx_out, long long tokens, const unsigned long long* __restrict__ bitmap,
               const int* __restrict__ block_prefix) {
    const long long tk = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tk >= tokens) return;
    const int h = tok_slot[tk];
    if (h < 0) {
        if (idx_out) idx_out[tk] = -1;
        return;
    }
    const int e = t.entries[h];  // >=0: existed before this build (or already finalized — same value)
    int row = e;
    const unsigned int ft = t.slot_tok[h];
    if (e < 0) {
        const int r = ln_rank_of_token(ft, bitmap, block_prefix);  // rank of the slot's smallest token among the first occurrences
        row = r;
        if (t.row_limit > 0 && row >= t.row_limit) row = -1;  // beyond the host's static row bound: stays un-inserted
        if (ft == (unsigned int)tk && row >= 0) {  // first occurrence publishes the vertex
            t.entries[h] = row;
            int key[D];
            KeyPack<D>::unpack(t.slot_keys[h], key, t.key_format);
#pragma unroll
            for (int i = 0; i < D; ++i) t.keys[(size_t)row * D + i] = key[i];
        }
    }
    if (idx_out) idx_out[tk] = row;
}

// Workspace of one build: first-occurrence bitmap, block counts/prefixes,
// token->slot scratch, token->position, the scratch of the slot-CSR construction (ln_csr.hip) and the bucket regions.
struct BuildWs {
    unsigned long long* bitmap;
    int* block_cnt;
    int* block_prefix;
    int* tok_slot;
    int* tok_pos;
    void* csr_ws;
    size_t csr_ws_bytes;
    int nb;
    // bucket regions: nbk x capb entries
    int capb;
    unsigned long long* part_pk;
    int* part_tok;
    int* part_slot;
    int* part_pos;
    unsigned int* pub;  // [buckets + 1] look-back words of k_bucket_rows, then its ticket counter
};

static size_t ln_align256(size_t x) { return (x + 255) & ~size_t(255); }

// Entries reserved per bucket region: 16x the mean bucket population + slack (coarse lattices put thousands of tokens
// on single vertices; HBM is plentiful: 320 B per token).  Clouds so skewed that one bucket receives more (e.g. all
// points identical) are rebuilt on the atomic path (LN_STATUS_BUCKET_OVERFLOW).
static int ln_bucket_region(long long tokens, int capacity) {
    const int nbk = ln_bucket_count(capacity);
    long long capb = 16 * ((tokens + nbk - 1) / nbk) + 1024;
    capb = (capb + 63) & ~63ll;
    return int(capb);
}

extern "C" int ln_table_bucket_count(int capacity) { return capacity > 0 ? ln_bucket_count(capacity) : 0; }

extern "C" size_t ln_build_workspace_bytes(long long tokens, int capacity) {
    if (tokens < 1) tokens = 1;
    if (capacity < 1) capacity = 1;
    const size_t nb = (size_t)ln_div_up(tokens, 256);
    const size_t region = (size_t)ln_bucket_count(capacity) * ln_bucket_region(tokens, capacity);
    return ln_align256(nb * 4 * sizeof(unsigned long long)) + 2 * ln_align256(nb * sizeof(int)) +
           2 * ln_align256((size_t)tokens * sizeof(int)) + ln_align256(ln_csr_scan_workspace_bytes(capacity)) +
           ln_align256(region * sizeof(unsigned long long)) + 3 * ln_align256(region * sizeof(int)) +
           ln_align256(((size_t)ln_bucket_count(capacity) + 1) * sizeof(unsigned int));
}

static int ln_carve_ws(long long tokens, int capacity, void* workspace, size_t bytes, BuildWs& ws) {
    LN_REQUIRE(workspace != nullptr && bytes >= ln_build_workspace_bytes(tokens, capacity), LN_ERR_WORKSPACE,
               "build workspace too small: %zu < %zu", bytes, ln_build_workspace_bytes(tokens, capacity));
    LN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, LN_ERR_WORKSPACE, "build workspace must be 256-byte aligned");
    LN_REQUIRE(tokens < 0x7FFFFFFFll, LN_ERR_ARG, "too many insertion tokens: %lld", tokens);
    char* p = static_cast<char*>(workspace);
    ws.nb = ln_div_up(tokens, 256);
    ws.bitmap = reinterpret_cast<unsigned long long*>(p);
    p += ln_align256((size_t)ws.nb * 4 * sizeof(unsigned long long));
    ws.block_cnt = reinterpret_cast<int*>(p);
    p += ln_align256((size_t)ws.nb * sizeof(int));
    ws.block_prefix = reinterpret_cast<int*>(p);
    p += ln_align256((size_t)ws.nb * sizeof(int));
    ws.tok_slot = reinterpret_cast<int*>(p);
    p += ln_align256((size_t)tokens * sizeof(int));
    ws.tok_pos = reinterpret_cast<int*>(p);
    p += ln_align256((size_t)tokens * sizeof(int));
    ws.csr_ws = p;
    ws.csr_ws_bytes = ln_csr_scan_workspace_bytes(capacity);
    p += ln_align256(ws.csr_ws_bytes);
    ws.capb = ln_bucket_region(tokens, capacity);
    const size_t region = (size_t)ln_bucket_count(capacity) * ws.capb;
    ws.part_pk = reinterpret_cast<unsigned long long*>(p);
    p += ln_align256(region * sizeof(unsigned long long));
    ws.part_tok = reinterpret_cast<int*>(p);
    p += ln_align256(region * sizeof(int));
    ws.part_slot = reinterpret_cast<int*>(p);
    p += ln_align256(region * sizeof(int));
    ws.part_pos = reinterpret_cast<int*>(p);
    p += ln_align256(region * sizeof(int));
    ws.pub = reinterpret_cast<unsigned int*>(p);
    return LN_OK;
}

template <int D>
static int ln_rank_rows(const LnTable& t, const int* tok_slot, int* idx_out, long long tokens, const BuildWs& ws, hipStream_t st);

// After the producer: slot CSR (scan of slot_cnt + fill) -> per-slot smallest token -> canonical rank.
template <int D>
static int ln_rank_and_finalize(const LnTable& t, const int* tok_slot, const int* tok_pos, int* idx_out, long long tokens,
                                const BuildWs& ws, const LnCsr& csr, hipStream_t st) {
    int rc = ln_csr_from_counts(tok_slot, tok_pos, tokens, t.slot_cnt, t.capacity, csr, ws.csr_ws, ws.csr_ws_bytes, st);
    if (rc) return rc;
    const long long max_seg = ln_csr_max_segments(tokens, t.capacity);
    LN_LAUNCH("k_seg_min", k_seg_min, dim3(ln_div_up(max_seg, 256)), dim3(256), 0, st, t, csr.csr_tok,
              reinterpret_cast<const int4*>(csr.seg_desc), csr.seg_count);
    return ln_rank_rows<D>(t, tok_slot, idx_out, tokens, ws, st);
}

// slot_tok (smallest token per slot) -> canonical row numbers (atomic build path: works on tables that already hold rows)
template <int D>
static int ln_rank_rows(const LnTable& t, const int* tok_slot, int* idx_out, long long tokens, const BuildWs& ws, hipStream_t st) {
    LN_LAUNCH("k_mark_first", k_mark_first, dim3(ws.nb), dim3(256), 0, st, t, tok_slot, tokens, ws.bitmap, ws.block_cnt);
    LN_LAUNCH("k_scan_blocks", k_scan_blocks, dim3(1), dim3(1024), 0, st, (const int*)ws.block_cnt, ws.bitmap, ws.nb, ws.block_prefix, t.nr_filled,
              t.status, t.host_counters, t.host_seq, 0);
    LN_LAUNCH("k_finalize", (k_finalize<D>), dim3(ws.nb), dim3(256), 0, st, t, tok_slot, idx_out, tokens, ws.bitmap, ws.block_prefix);
    return ln_check_launch("ln build (mark/scan/finalize)");
}

// ------------------------------------------------------------------------------------------
// ln_canonicalize: slot-order rows of ONE fresh bucketed build -> first-occurrence order
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_canon_mark(LnTable t, unsigned long long* __restrict__ bitmap) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= t.capacity || t.entries[h] < 0) return;
    const unsigned int ft = t.slot_tok[h];
    atomicOr(&bitmap[ft >> 6], 1ull << (ft & 63));
}

template <int D>
__global__ void __launch_bounds__(256)
    k_canon_slots(LnTable t, const unsigned long long* __restrict__ bitmap, const int* __restrict__ block_prefix, int* __restrict__ perm) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= t.capacity) return;
    const int old_row = t.entries[h];
    if (old_row < 0) return;
    const int row = ln_rank_of_token(t.slot_tok[h], bitmap, block_prefix);
    perm[old_row] = row;
    t.entries[h] = row;
    int key[D];
    KeyPack<D>::unpack(t.slot_keys[h], key, t.key_format);  // (from the slot, not from keys[old_row]: the permutation runs in place)
#pragma unroll
    for (int i = 0; i < D; ++i) t.keys[(size_t)row * D + i] = key[i];
}

// the bucketed build's segment descriptors carry rows (LnCsr.seg_count): relabel them as well
__global__ void __launch_bounds__(256) k_canon_segs(int4* __restrict__ seg_desc, const int* __restrict__ seg_count, long long seg_region,
                                                    const int* __restrict__ perm) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const int region = int(g / seg_region);
    const long long s = g - (long long)region * seg_region;
    if (region >= seg_count[LN_XCD_GROUPS] || s >= seg_count[region] || !(seg_count[LN_XCD_GROUPS + 1] & 1)) return;
    const int r = seg_desc[g].x;
    if (r >= 0) seg_desc[g].x = perm[r];
}

__global__ void __launch_bounds__(256) k_canon_idx(int* __restrict__ idx, long long tokens, const int* __restrict__ perm) {
    const long long tk = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tk >= tokens) return;
    const int r = idx[tk];
    if (r >= 0) idx[tk] = perm[r];
}

template <int D>
static int ln_canonicalize_impl(const LnTable& t, int* idx, long long tokens, const LnCsr* csr, const BuildWs& ws, hipStream_t st) {
    const int dbg = ln_debug_mask();
    if (dbg & 256) return LN_OK;
    if (ln_zero_async(ws.bitmap, (size_t)ws.nb * 4 * sizeof(unsigned long long), st)) return LN_ERR_LAUNCH;
    const int slot_blocks = ln_div_up(t.capacity, 256);
    LN_LAUNCH("k_canon_mark", k_canon_mark, dim3(slot_blocks), dim3(256), 0, st, t, ws.bitmap);
    LN_LAUNCH("k_scan_blocks", k_scan_blocks, dim3(1), dim3(1024), 0, st, (const int*)nullptr, ws.bitmap, ws.nb, ws.block_prefix, t.nr_filled, t.status,
              (int*)nullptr, 0, 1);
    if (!(dbg & 512)) LN_LAUNCH("k_canon_slots", k_canon_slots<D>, dim3(slot_blocks), dim3(256), 0, st, t, ws.bitmap, ws.block_prefix, ws.tok_pos);
    if (idx && !(dbg & 64)) LN_LAUNCH("k_canon_idx", k_canon_idx, dim3(ws.nb), dim3(256), 0, st, idx, tokens, ws.tok_pos);
    if (csr && csr->seg_desc && csr->seg_count && !(dbg & 32))
        LN_LAUNCH("k_canon_segs", k_canon_segs, dim3(ln_div_up((long long)LN_XCD_GROUPS * csr->seg_region, 256)), dim3(256), 0, st,
                  reinterpret_cast<int4*>(csr->seg_desc), csr->seg_count, csr->seg_region, ws.tok_pos);
    return ln_check_launch("ln_canonicalize");
}

static int ln_check_csr(const LnCsr* c, const char* who) {
    LN_REQUIRE(c && c->grp_start && c->csr_tok && c->seg_desc && (reinterpret_cast<uintptr_t>(c->seg_desc) & 15) == 0 && c->seg_count, LN_ERR_ARG, "%s: CSR output has a null buffer", who);
    return LN_OK;
}

static int ln_build_points(const LnTable* t, const float* positions_raw, const float* sigmas_host, int n, int* idx, float* w,
                           int flags, const float* vals, int val_dim, float* distributed, const LnCsr* csr, void* workspace,
                           size_t workspace_bytes, float* clear_values, long long clear_values_elems, void* stream, const char* who) {
    int rc = ln_check_table(t, who);
    if (rc) return rc;
    rc = ln_check_csr(csr, who);
    if (rc) return rc;
    const int write_idx = flags & LN_BUILD_WRITE_IDX;
    LN_REQUIRE(n >= 0, LN_ERR_ARG, "%s: n=%d", who, n);
    LN_REQUIRE(positions_raw != nullptr || n == 0, LN_ERR_ARG, "%s: null positions", who);
    LN_REQUIRE(!write_idx || n == 0 || (idx && w), LN_ERR_ARG, "%s: write_idx set but idx/w null", who);
    const long long tokens = (long long)n * (t->pos_dim + 1);
    BuildWs ws;
    if (n > 0) {
        rc = ln_carve_ws(tokens, t->capacity, workspace, workspace_bytes, ws);
        if (rc) return rc;
    }
    // The bucketed path needs a table it knows to be empty: it is taken when the clear rides in this call, and then does
    // the clearing itself (no k_table_clear launch).  Its cursors use the first nbk+1 words of slot_cnt.
    // Beyond LN_BKT_MAX buckets the buckets grow instead; one bucket's staging area has to fit the 160 KB of LDS of a gfx950 CU
    // (tables past ~8.7M slots — 36 bytes of LDS per slot of a bucket, LN_BKT_LDS_LIMIT — take the atomic path).
    const int sb_lds = (t->slot_map && t->bucket_slots_max > 0) ? t->bucket_slots_max : ln_bucket_slots(t->capacity);  // largest bucket staged in LDS
    const size_t bucket_lds = (size_t)sb_lds * LN_BKT_LDS_PER_SLOT + LN_BKT_LDS_EXTRA;
    const bool bucketed = n > 0 && (flags & LN_BUILD_CLEAR_FIRST) && !(flags & LN_BUILD_ATOMIC_PATH) &&
                          t->capacity > ln_bucket_count(t->capacity) && bucket_lds <= LN_BKT_LDS_LIMIT &&
                          (long long)ln_bucket_count(t->capacity) * ln_bucket_region(tokens, t->capacity) < 0x7FFFFFFFll;  // int region offsets
    LN_REQUIRE(clear_values == nullptr || (reinterpret_cast<uintptr_t>(clear_values) & 15) == 0, LN_ERR_ARG, "%s: clear_values must be 16-byte aligned", who);
    if ((flags & LN_BUILD_CLEAR_FIRST) && !bucketed) {
        rc = ln_table_clear(t, clear_values, clear_values_elems, stream);
        if (rc) return rc;
    }
    if (n == 0) return LN_OK;
    hipStream_t st = (hipStream_t)stream;
    int* tok_slot = write_idx ? idx : ws.tok_slot;  // atomic path: idx doubles as the token->slot scratch
    LN_DISPATCH_D(t->pos_dim, {
        LnScale<D> sc = ln_make_scale<D>(sigmas_host);
        if (bucketed) {
            const int sb = sb_lds;  // (hashed table: THE bucket size; slot map: the largest one — the kernels take each bucket's size from the map)
            // (slot map: 8 leaves x the same number of buckets — ln_bucket_count / 8 each, as the host laid the map out)
            const int nbk = t->slot_map ? (ln_bucket_count(t->capacity) / LN_XCD_GROUPS) * LN_XCD_GROUPS : ln_bucket_count(t->capacity);
            const size_t lds = bucket_lds;
            int* dropped_idx = write_idx ? idx : (int*)nullptr;  // tokens that never reach a bucket get idx = -1 in pass 1
            // the same 512-point tile (the same number of cursor atomics) on 512 threads (a whole point each), 1024 (half a point each) or
            // 256 (two points each): LN_DEBUG_MASK & 2097152 selects 256, & 4194304 selects 1024 (A/B)
            if (D <= 3 && (ln_debug_mask() & 4194304))
                LN_LAUNCH("k_point_keys", (k_point_keys<D, (D <= 3 ? 1024 : 256)>), dim3(ln_div_up(n, 512)), dim3(1024), 0, st, *t, positions_raw, sc, n, sb,
                          nbk, ws.capb, ws.part_tok, ws.part_pk, dropped_idx, write_idx ? w : (float*)nullptr, vals, val_dim, distributed,
                          csr->seg_count, csr->planes ? LN_XCD_GROUPS : 1, clear_values, clear_values_elems, ws.pub);
            else if (D <= 3 && !(ln_debug_mask() & 2097152))
                LN_LAUNCH("k_point_keys", (k_point_keys<D, (D <= 3 ? 512 : 256)>), dim3(ln_div_up(n, 512)), dim3(512), 0, st, *t, positions_raw, sc, n, sb,
                          nbk, ws.capb, ws.part_tok, ws.part_pk, dropped_idx, write_idx ? w : (float*)nullptr, vals, val_dim, distributed,
                          csr->seg_count, csr->planes ? LN_XCD_GROUPS : 1, clear_values, clear_values_elems, ws.pub);
            else
                LN_LAUNCH("k_point_keys", (k_point_keys<D, 256>), dim3(ln_div_up(n, LN_KEYS_PTS_PER_BLOCK)), dim3(256), 0, st, *t, positions_raw, sc, n, sb,
                          nbk, ws.capb, ws.part_tok, ws.part_pk, dropped_idx, write_idx ? w : (float*)nullptr, vals, val_dim, distributed,
                          csr->seg_count, csr->planes ? LN_XCD_GROUPS : 1, clear_values, clear_values_elems, ws.pub);
            // Workgroup size of the bucket pass.  1024 threads finish a bucket soonest (C3, 256 buckets of 1 875 tokens: 19.9 us against 22.7
            // on 512 threads), but one such workgroup takes half a CU's wave slots; with several scans in flight the narrower workgroups
            // pack beside the other scans' kernels: C3 1407 -> 1441 Mpoints/s, C4 (3 100 tokens per bucket) 947 -> 980, C2 with 16 clouds
            // per step 494 -> 508 — C5, 7 500 tokens per bucket, 1421 -> 1401 (LN_BKT_THREADS=512 builds, one box).  The caller says
            // whether builds overlap with other work (ln_build_concurrency); alone, and on big buckets, 1024 stays.
            static int narrow_ok = -1;  // LN_BKT_NARROW=0: 1024-thread bucket workgroups whatever the concurrency (A/B; read once)
            if (narrow_ok < 0) {
                const char* ev = getenv("LN_BKT_NARROW");
                narrow_ok = (ev && ev[0] == '0') ? 0 : 1;
            }
            const bool narrow = narrow_ok && g_ln_build_concurrency > 1 && tokens <= (long long)LN_BKT_NARROW_TOKENS * nbk;
            if (narrow)
                LN_LAUNCH("k_bucket_rows", (k_bucket_rows<D, 512>), dim3(nbk), dim3(512), lds, st, *t, sb, nbk, ws.capb, t->slot_cnt,
                      ws.part_tok, ws.part_pk, ws.part_slot, ws.part_pos, dropped_idx, *csr, ws.pub);
            else
                LN_LAUNCH("k_bucket_rows", (k_bucket_rows<D, LN_BKT_THREADS>), dim3(nbk), dim3(LN_BKT_THREADS), lds, st, *t, sb, nbk, ws.capb, t->slot_cnt,
                      ws.part_tok, ws.part_pk, ws.part_slot, ws.part_pos, dropped_idx, *csr, ws.pub);
            rc = ln_check_launch(who);
            if (rc == LN_OK && (flags & LN_BUILD_CANONICAL_ROWS)) rc = ln_canonicalize_impl<D>(*t, dropped_idx, tokens, csr, ws, st);
            if (rc == LN_OK && (flags & LN_BUILD_SORTED_CSR)) rc = ln_csr_sort_groups(*csr, t->capacity, ws.tok_slot, st);
        } else {
            LN_LAUNCH("k_insert_points", k_insert_points<D>, dim3(ln_div_up(tokens, 256)), dim3(256), 0, st, *t, positions_raw, sc, n, tok_slot,
                      ws.tok_pos, write_idx ? w : (float*)nullptr, vals, val_dim, distributed);
            rc = ln_rank_and_finalize<D>(*t, tok_slot, ws.tok_pos, write_idx ? idx : (int*)nullptr, tokens, ws, *csr, st);
            // (tok_pos is free again behind the fill: the scratch of the sort; tok_slot may be the caller's idx)
            if (rc == LN_OK && (flags & LN_BUILD_SORTED_CSR)) rc = ln_csr_sort_groups(*csr, t->capacity, ws.tok_pos, st);
        }
    });
    return rc;
}

extern "C" int ln_build_splat(const LnTable* t, const float* positions_raw, const float* sigmas_host, int n, int* idx,
                              float* w, int flags, const LnCsr* csr, void* workspace, size_t workspace_bytes, float* clear_values,
                              long long clear_values_elems, void* stream) {
    return ln_build_points(t, positions_raw, sigmas_host, n, idx, w, flags, nullptr, 0, nullptr, csr, workspace, workspace_bytes,
                           clear_values, clear_values_elems, stream, "ln_build_splat");
}

extern "C" int ln_distribute(const LnTable* t, const float* positions_raw, const float* sigmas_host, const float* vals, int n,
                             int val_dim, int* idx, float* w, float* distributed, int flags, const LnCsr* csr, void* workspace,
                             size_t workspace_bytes, float* clear_values, long long clear_values_elems, void* stream) {
    LN_REQUIRE(vals && distributed && idx && w, LN_ERR_ARG, "ln_distribute: null buffer");
    LN_REQUIRE(val_dim >= 1, LN_ERR_ARG, "ln_distribute: val_dim=%d", val_dim);
    return ln_build_points(t, positions_raw, sigmas_host, n, idx, w, flags | LN_BUILD_WRITE_IDX, vals, val_dim, distributed, csr, workspace,
                           workspace_bytes, clear_values, clear_values_elems, stream, "ln_distribute");
}

// ------------------------------------------------------------------------------------------
// ln_rehash: re-insert the existing vertices (rows 0 .. nr_filled-1, keys[]) into the slot range [0, t->capacity)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rehash_clear(LnTable t) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < t.capacity; i += stride) {
        t.slot_keys[i] = LN_EMPTY_KEY;
        t.slot_tok[i] = LN_EMPTY_TOK;
        t.slot_cnt[i] = 0;
        t.entries[i] = -1;
    }
}

template <int D>
__global__ void __launch_bounds__(256) k_rehash_rows(LnTable t) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= *t.nr_filled || r >= t.capacity) return;
    int key[D];
#pragma unroll
    for (int i = 0; i < D; ++i) key[i] = t.keys[(size_t)r * D + i];
    const uint64_t pk = KeyPack<D>::pack(key, t.key_format);  // (it was packable when it was inserted)
    const LnProbe pr = LnProbe::of_key<D>(key, t, ln_bucket_slots(t.capacity));
    for (int probes = 0; probes < t.capacity; ++probes) {
        const int h = pr.slot(probes);
        if (atomicCAS(&t.slot_keys[h], (unsigned long long)LN_EMPTY_KEY, (unsigned long long)pk) == LN_EMPTY_KEY) {
            t.entries[h] = r;
            t.slot_tok[h] = 0u;  // an existing vertex: never "first occurrence" of a later build (k_mark_first tests entries[h] >= 0)
            return;
        }
    }
    atomicOr(t.status, LN_STATUS_TABLE_FULL);
}

extern "C" int ln_rehash(const LnTable* t, void* stream) {
    int rc = ln_check_table(t, "ln_rehash");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    int blocks = ln_div_up(t->capacity, 256);
    LN_LAUNCH("k_rehash_clear", k_rehash_clear, dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, st, *t);
    LN_DISPATCH_D(t->pos_dim, { LN_LAUNCH("k_rehash_rows", k_rehash_rows<D>, dim3(blocks), dim3(256), 0, st, *t); });
    return ln_check_launch("ln_rehash");
}

extern "C" int ln_canonicalize(const LnTable* t, int* idx, long long tokens, const LnCsr* csr, void* workspace, size_t workspace_bytes,
                               void* stream) {
    int rc = ln_check_table(t, "ln_canonicalize");
    if (rc) return rc;
    LN_REQUIRE(tokens >= 0, LN_ERR_ARG, "ln_canonicalize: tokens=%lld", tokens);
    if (tokens == 0) return LN_OK;
    BuildWs ws;
    rc = ln_carve_ws(tokens, t->capacity, workspace, workspace_bytes, ws);
    if (rc) return rc;
    LN_DISPATCH_D(t->pos_dim, { rc = ln_canonicalize_impl<D>(*t, idx, tokens, csr, ws, (hipStream_t)stream); });
    return rc;
}

extern "C" int ln_coarsen(const LnTable* fine, int fine_rows_upper, const LnTable* coarse, const LnCsr* csr, void* workspace,
                          size_t workspace_bytes, void* stream) {
    int rc = ln_check_table(fine, "ln_coarsen(fine)");
    if (rc) return rc;
    rc = ln_check_table(coarse, "ln_coarsen(coarse)");
    if (rc) return rc;
    rc = ln_check_csr(csr, "ln_coarsen");
    if (rc) return rc;
    LN_REQUIRE(fine->pos_dim == coarse->pos_dim, LN_ERR_ARG, "ln_coarsen: pos_dim mismatch");
    if (fine_rows_upper <= 0) return LN_OK;
    const long long tokens = (long long)fine_rows_upper * (2 * (fine->pos_dim + 1) + 1);
    BuildWs ws;
    rc = ln_carve_ws(tokens, coarse->capacity, workspace, workspace_bytes, ws);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH_D(fine->pos_dim, {
        LN_LAUNCH("k_insert_coarse", k_insert_coarse<D>, dim3(ln_div_up(fine_rows_upper, 256)), dim3(256), 0, st, *fine, fine_rows_upper,
                  *coarse, ws.tok_slot, ws.tok_pos);
        rc = ln_rank_and_finalize<D>(*coarse, ws.tok_slot, ws.tok_pos, (int*)nullptr, tokens, ws, *csr, st);
    });
    return rc;
}

// ------------------------------------------------------------------------------------------
// neighbour traversal (LatticeGPU.cuh:1479-1684): one thread per (query vertex, filter slot)
// ------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
    k_neighbours(LnTable tq, int query_rows_upper, LnTable tn, float scale, int dilation, int flip, int* __restrict__ nbr) {
    // whole vertices per workgroup; same table on both sides and space-ordered: the vertices of kd region x on XCD x (see k_reduce_and_neighbours)
    constexpr int E = 2 * (D + 1) + 1, ROWS = 256 / E;
    const LnSlotMap smap = ln_load_slot_map(tn.slot_map);  // (first: its scalar loads travel with those of the partition)
    const int tile = ln_partition_tile((int)blockIdx.x, (int)gridDim.x, tq.entries == tn.entries ? tq.row_regions : nullptr, ROWS);
    if ((int)threadIdx.x < ROWS * E)
        ln_neighbours_body<D>((long long)tile * (ROWS * E) + threadIdx.x, tq, query_rows_upper, tn, scale, dilation, flip, nbr, smap);
}

extern "C" int ln_neighbours(const LnTable* query, int query_rows_upper, const LnTable* neigh, int lvl_query, int lvl_neigh,
                             int dilation, int flip, int* nbr, void* stream) {
    int rc = ln_check_table(query, "ln_neighbours(query)");
    if (rc) return rc;
    rc = ln_check_table(neigh, "ln_neighbours(neigh)");
    if (rc) return rc;
    LN_REQUIRE(query->pos_dim == neigh->pos_dim, LN_ERR_ARG, "ln_neighbours: pos_dim mismatch");
    const int diff = lvl_query - lvl_neigh;
    LN_REQUIRE(diff >= -1 && diff <= 1, LN_ERR_ARG,
               "ln_neighbours: lattices must be at most one level apart (query lvl %d, neighbours lvl %d)", lvl_query,
               lvl_neigh);  // Lattice.cu:439
    LN_REQUIRE(dilation >= 1, LN_ERR_ARG, "ln_neighbours: dilation=%d", dilation);
    LN_REQUIRE(nbr != nullptr || query_rows_upper == 0, LN_ERR_ARG, "ln_neighbours: null output");
    if (query_rows_upper <= 0) return LN_OK;
    const float scale = diff == 0 ? 1.0f : (diff > 0 ? 2.0f : 0.5f);  // pow(2, lvl_diff), LatticeGPU.cuh:1488
    const int E = 2 * (query->pos_dim + 1) + 1;
    LN_DISPATCH_D(query->pos_dim, {
        LN_LAUNCH("k_neighbours", k_neighbours<D>, dim3(ln_div_up(query_rows_upper, 256 / E)), dim3(256), 0, (hipStream_t)stream, *query,
                           query_rows_upper, *neigh, scale, dilation, flip, nbr);
    });
    return ln_check_launch("ln_neighbours");
}

// ------------------------------------------------------------------------------------------
// simplex retrieval for slice_no_precomputation (LatticeGPU.cuh:2598-2750, index part)
// ------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
    k_retrieve_points(LnTable t, const float* __restrict__ pos_raw, LnScale<D> sc, int n, int* __restrict__ idx,
                      float* __restrict__ w) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float pr[D];
#pragma unroll
    for (int i = 0; i < D; ++i) pr[i] = pos_raw[(size_t)p * D + i];
    LnSimplex<D> s;
    ln_simplex<D>(pr, sc, s);
    ln_simplex_of_cloud<D>(s, p, t.batch_points, t.batch_key_step);
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        int key[D];
        ln_vertex_key<D>(s, r, key);
        const int row = ln_retrieve<D>(t, key);
        idx[(size_t)p * (D + 1) + r] = row;  // -1 when absent (Lattice.cu:812-815 pre-fill)
        w[(size_t)p * (D + 1) + r] = row >= 0 ? s.bary[r] : -1.0f;
    }
}

int ln_retrieve_points(const LnTable* t, const float* positions_raw, const float* sigmas_host, int n, int* idx, float* w,
                       void* stream) {
    int rc = ln_check_table(t, "ln_slice_no_precomputation");
    if (rc) return rc;
    if (n <= 0) return LN_OK;
    LN_DISPATCH_D(t->pos_dim, {
        LnScale<D> sc = ln_make_scale<D>(sigmas_host);
        LN_LAUNCH("k_retrieve_points", k_retrieve_points<D>, dim3(ln_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, *t, positions_raw,
                           sc, n, idx, w);
    });
    return ln_check_launch("ln_slice_no_precomputation(retrieve)");
}
