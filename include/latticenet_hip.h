/* latticenet_hip.h — C ABI of the MI355X (gfx950) permutohedral-lattice backend.
 *
 * Drop-in boundary for ONE path of AIS-Bonn/lattice_net: everything `latticenet.Lattice`
 * (src/PyBridge.cxx:41-113) forwards to `LatticeGPU` launch wrappers
 * (include/lattice_net/kernels/LatticeGPU.cuh:42-412).  Each entry point below names the
 * reference interface it replaces.  Plain C: device pointers, sizes, a stream handle
 * (hipStream_t passed as void*), int status.  No torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - all tensors are contiguous row-major; positions/values/weights fp32, indices int32;
 *   - return value 0 = launched OK; <0 = argument / launch error, text via
 *     ln_last_error_string(); kernels never block the host;
 *   - asynchronous device-side conditions (table full, key out of packable range) are
 *     recorded in LnTable.status and must be read back by the caller (ln_status_string).
 *
 * Vertex numbering: deterministic.  Builds that start from a cleared table number the vertices bucket by bucket
 * (runs of consecutive hash slots; inside a bucket by first occurrence) — the reference's own numbering is thread-arrival
 * order and not reproducible; with
 * LN_BUILD_CANONICAL_ROWS / ln_canonicalize, and on every incremental build, rows are numbered by first
 * occurrence in (point, remainder) order, i.e. what a serial run of HashTableGPU::insert
 * (HashTableGPU.cuh:425-484) produces.
 */
#ifndef LATTICENET_HIP_H
#define LATTICENET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LN_OK 0
#define LN_ERR_ARG (-1)
#define LN_ERR_UNSUPPORTED (-2)
#define LN_ERR_LAUNCH (-3)
#define LN_ERR_WORKSPACE (-4)

/* bits of *LnTable.status (device int32) */
#define LN_STATUS_TABLE_FULL 1      /* insert probed every slot (reference would spin forever, HashTableGPU.cuh:443) */
#define LN_STATUS_KEY_RANGE 2       /* a lattice key did not fit the packed 64-bit slot format */
#define LN_STATUS_BUCKET_OVERFLOW 4 /* bucketed build: one LDS-staged bucket filled up; nothing of this build is valid,
                                       re-run it with LN_BUILD_ATOMIC_PATH (whose inserts spill past a full bucket) */

#define LN_MAX_POS_DIM 6
#define LN_SLOT_MAP_INTS 32 /* ints behind LnTable.slot_map */
#define LN_KEYS_RAW 0     /* LnTable.key_format */
#define LN_KEYS_LATTICE 1
#define LN_NOT_VISITED (-2)          /* neighbour-list code: traversal never looks at this slot */

/* Device-side open-addressing table.  Replaces HashTableGPU (HashTableGPU.cuh:12-30) and its
 * host owner HashTable (src/HashTable.cu:21-47).  `keys`, `entries`, `nr_filled` have exactly
 * the reference's meaning and layout; `slot_keys`/`slot_tok`/`status` are additions:
 *   slot_keys[h]  the lattice key stored in slot h, packed into 64 bits (all-ones = empty), so
 *                 claim + key publication is ONE 64-bit CAS (no spin lock, no fence);
 *   slot_tok[h]   smallest insertion token that touched slot h during the build that created
 *                 it (0xFFFFFFFF while empty) — defines the canonical row order;
 *   slot_cnt[h]   how many tokens of the current build landed on slot h;
 *   status        sticky error bits, see LN_STATUS_*. */
typedef struct LnTable {
    int capacity;
    int pos_dim;
    unsigned long long* slot_keys; /* [capacity] */
    unsigned int* slot_tok;        /* [capacity] */
    int* slot_cnt;                 /* [capacity] tokens of the current build per slot (scratch) */
    int* entries;                  /* [capacity]   slot -> row, -1 empty  (m_entries) */
    int* keys;                     /* [capacity,d] row  -> key            (m_keys)    */
    int* nr_filled;                /* [1]                                 (m_nr_filled) */
    int* status;                   /* [1] */
    int* host_counters;            /* NULL, or 8-byte aligned pinned, device-visible host memory: every build ends with ONE 64-bit store
                                      there: nr_filled | status << 32 | (host_seq & 0xFFFFFF) << 40.  A host that passes a fresh
                                      host_seq per build can spin on that word and read the vertex count as soon as the build's last
                                      kernel has got there, without a copy or an event */
    int host_seq;                  /* sequence number of this build (24 bits are reported) */
    int key_format;                /* how slot_keys packs a key (fixed for the life of the table's contents): 0 = raw, any integer
                                      tuple (needed by ln_coarsen's target, which receives halved fine keys); 1 = lattice points
                                      only — remainder + quotients, 6-7x the coordinate range for pos_dim 5-6 (every table
                                      built from positions).  See KeyPack in csrc/ln_common.h */
    int row_limit;                 /* 0 = none.  > 0 (static-rows mode of the host, whose [rows, *] tensors are this tall): a build
                                      leaves vertices that would get a row >= row_limit un-inserted — idx = -1, no entries[] /
                                      keys[] row — so that no consumer can index past the host's tensors; nr_filled still
                                      counts them, which is how the host notices (nr_filled > row_limit) */
    const int* slot_map;           /* NULL (slots hashed over the whole table, as the reference), or LN_SLOT_MAP_INTS device ints that
                                      order the SLOTS — and with them the rows — of this table by space:
                                        [0..6]   planes of a 3-level kd partition of KEY space into 8 leaves, heap order (node i has the
                                                 children 2i+1: key below the plane, 2i+2: key >= plane; level l compares key[l % pos_dim]);
                                        [7]      buckets per leaf = ln_table_bucket_count(capacity) / 8;
                                        [8..16]  first slot of each leaf's run of slots, then the end of the last run (<= capacity);
                                        [17..24] slots per bucket inside each leaf (run length = [7] x this).
                                      A key starts probing inside its leaf's run (the hash picks the slot), rows are numbered bucket by
                                      bucket, so the vertices of a leaf own one contiguous row range and the gathers of the path (9
                                      neighbour rows per vertex, d+1 rows per point) stay inside one XCD's L2.  The map must stay what it
                                      was when the table's contents were inserted (retrieval uses the same function): change it only in
                                      front of a build that starts with a clear.  A map whose leaves do not fit the cloud overfills
                                      buckets (LN_STATUS_BUCKET_OVERFLOW -> rebuild on the atomic path, whose inserts spill); a fitting
                                      one comes from the host's calibration (Lattice.balanced_region_planes).  csrc/ln_common.h */
    int bucket_slots_max;          /* with a slot map: the largest of its slots-per-bucket entries (sizes the LDS of the bucket pass) */
    int batch_points;              /* 0, or > 0: the positions handed to a build / retrieval of this table are a BATCH of independent clouds
                                      of batch_points points each (cloud c = point index / batch_points).  Cloud c's lattice keys are then
                                      shifted by c * batch_key_step on the first coordinate — a multiple of pos_dim + 1, i.e. a translation
                                      of the lattice onto itself — so that the clouds' vertex sets are disjoint in ONE table and every kernel
                                      of the path (neighbour lists, convolutions, slice, the scatters) runs over the batch in one launch with
                                      per-cloud results (the multi-cloud launch form for clouds too small to fill the chip) */
    int batch_key_step;            /* (pos_dim + 1) x the quotient distance between clouds; halved per coarser level so that the
                                      level-crossing neighbour search (keys x 2^(lvl difference)) stays inside a cloud */
    int* row_regions;              /* NULL, or LN_XCD_GROUPS + 1 device ints a bucketed build over a space-ordered table fills: the first row
                                      of each of the 8 top-level kd regions and the row count — the argument of ln_conv_row_partition */
} LnTable;

/* Adjacency "group -> the tokens that touch it" in CSR form, cut into segments of at most 16
 * entries (the unit of work of ln_csr_reduce_rows).  A group is a hash slot for the CSR that
 * ln_build_splat / ln_distribute / ln_coarsen emit as a by-product (row of a group = entries[slot]),
 * or a row for ln_csr_build (row of a group = the group). */
#define LN_XCD_GROUPS 8 /* segment / point / vertex lists are kept per XCD group (see csrc/ln_common.h, LnProbe) */
typedef struct LnCsr {
    int* grp_start; /* [groups_upper + 1] */
    int* csr_tok;   /* [tokens]           tokens grouped by group */
    int* seg_desc;  /* [LN_XCD_GROUPS * seg_region * 4], 16-byte aligned: one descriptor {group, first CSR entry, entries from
                       there to the end of the group, offset of the first entry inside the group} per segment — everything a
                       reduce needs about a segment in ONE 16-byte load; region g holds the segments of XCD group g */
    int* seg_count; /* [LN_XCD_GROUPS + 2] device-side number of segments per region, then the number of regions in use
                       (1: everything in region 0 — ln_csr_build, the atomic build path; LN_XCD_GROUPS: bucketed build with
                       planes), then the descriptor format: 0 = the first descriptor word is the group; 1 (bucketed build) = it is the
                       ROW of the group, so a reduce needs no group -> row lookup */
    long long seg_region; /* entries per region (>= ln_csr_max_segments: one region may hold every segment) */
    const int* planes;    /* NULL (everything in region 0), or 7 device ints: split planes of a 3-level kd partition of KEY space
                             into LN_XCD_GROUPS compact regions — key[0] >= planes[0] picks the half, key[1] against
                             planes[1 + half] the quarter, key[2] against planes[3 + quarter] the region.  A bucketed
                             build then files every vertex's segments under its region, and the scatter kernels let
                             XCD r walk region r: the d+1 gathers of a point row meet in ONE L2.  These planes only steer
                             work placement (any values are correct).  When the table itself is space-ordered
                             (LnTable.slot_map) the region of a vertex is that of its bucket and this array is only a flag. */
    int dense;            /* host-side hints for the segment reduces (any value is correct).  Bit 0 = dense cloud, about 16 or more
                             tokens per vertex — most vertices then own several segments, and the reduce combines partial sums across
                             the waves of a workgroup before it resorts to atomics (C5: 104 -> 84 us; costs the sparse C3 scan 10 %).
                             Bit 1 = deterministic: one lane group walks a whole row in CSR order and stores it — no atomics, no
                             combining; with LN_BUILD_SORTED_CSR the sums are identical run to run (slower on hot vertices) */
} LnCsr;

const char* ln_last_error_string(void);
const char* ln_version(void);
/* short hash of THIS header as it was when the library was built (the binding compares it with the header it was written
 * against and refuses a stale library) */
const char* ln_abi_hash(void);

/* Live per-kernel timing (used by bench.py for the roofline block).  ln_profile_begin arms timing of every launch of the
 * kernels named in `kernel_names` (one name of ln_kernel_names, or several separated by commas), up to max_samples launches in
 * total: each such launch carries a (start, stop) event pair bound to the dispatch itself (hipExtLaunchKernelGGL), i.e. the
 * kernel's own duration as rocprofv3 reports it.  ln_profile_end waits for the recorded pairs, returns the summed duration and
 * the number of launches, and disarms.  Launches inside a stream capture cannot be timed this way (do not arm across one). */
const char* ln_kernel_names(void);
int ln_profile_begin(const char* kernel_names, int max_samples);
int ln_profile_end(double* total_ms, int* launches);
/* The same with a breakdown: kernel_names "*" arms every launch of the library; ln_profile_end_table ends the profiling like
 * ln_profile_end and writes one line "name launches total_ms\n" per launch name that occurred (in order of first occurrence,
 * NUL-terminated, truncated to out_bytes) — bench.py's per-operator roofline table. */
int ln_profile_end_table(char* out, int out_bytes);

/* HashTable::clear (src/HashTable.cu:49-57): entries=-1, keys=0, nr_filled=0 (+ our slots/status)
 * in one launch.  `values` (may be NULL) is zero-filled too: values_elems floats. */
int ln_table_clear(const LnTable* t, float* values, long long values_elems, void* stream);

/* The freshly allocated structure buffers of a table in one launch (the reference's HashTable::init issues one fill per tensor,
 * src/HashTable.cu:21-47): `arena` holds `words` 32-bit words; words [minus_begin, minus_end) are set to -1 (the entries), every
 * other word to 0 (keys, slot counters, the device counters, a placeholder values row — carved from the arena by the host). */
int ln_arena_init(int* arena, long long words, long long minus_begin, long long minus_end, void* stream);

/* Buckets (runs of consecutive slots, one workgroup of the bucketed build each) of a table that hashes into `capacity` slots: what a
 * host needs to lay out LnTable.slot_map. */
int ln_table_bucket_count(int capacity);
/* How many builds / scans the CALLING THREAD keeps in flight on this GPU (1 = one at a time, the default).  A speed hint for the builds
 * it issues afterwards: with several in flight the bucket pass of a build over small buckets runs on 512-thread workgroups, which finish
 * a bucket later but pack beside the other scans' kernels (ln_table.hip).  No effect on results.  (The reference has no counterpart: its
 * build is one stream's work, Lattice.cu:196-241.) */
int ln_build_concurrency(int scans_in_flight);

/* Scratch needed by ln_build_splat / ln_distribute / ln_coarsen for `tokens` insertions. */
size_t ln_build_workspace_bytes(long long tokens, int capacity);

/* kernel_splat (LatticeGPU.cuh:707-842) behind Lattice::splat_standalone / just_create_verts
 * (src/Lattice.cu:196-290), with `positions_raw / sigmas` (Lattice.cu:226) fused in.
 * Inserts the d+1 simplex vertices of every point; when flags has LN_BUILD_WRITE_IDX writes
 * idx[n*(d+1)] (row ids) and w[n*(d+1)] (barycentric weights).  Both must be pre-sized; rows
 * that cannot be inserted keep -1 (Lattice.cu:212-215 semantics are produced here, no pre-fill
 * needed).  idx/w may be NULL without that flag.
 * `csr` (required; groups = hash slots, groups_upper = capacity, sized with ln_csr_max_segments)
 * receives the slot -> tokens adjacency of this build: the build needs it to find each vertex's
 * first occurrence, and the caller reuses it for every scatter onto the vertices
 * (ln_csr_reduce_rows with grp_row = t->entries).
 * flags: LN_BUILD_WRITE_IDX = write idx / w; LN_BUILD_CLEAR_FIRST = run ln_table_clear(t, clear_values,
 * clear_values_elems) first, in the same call (begin_splat + splat in one host round trip).
 * Two build paths produce the same table, rows, idx and CSR semantics:
 *   bucketed (taken when LN_BUILD_CLEAR_FIRST is set and LN_BUILD_ATOMIC_PATH is not): tokens are partitioned by
 *     hash bucket and each bucket is resolved by one workgroup in LDS, no per-token global atomics.  A bucket
 *     of ~512 slots that fills up (tables loaded beyond ~0.85) sets LN_STATUS_BUCKET_OVERFLOW.
 *   atomic (any table state, any load < 1): one 64-bit CAS per new vertex + one returning add per token. */
#define LN_BUILD_WRITE_IDX 1
#define LN_BUILD_CLEAR_FIRST 2
#define LN_BUILD_ATOMIC_PATH 4
#define LN_BUILD_CANONICAL_ROWS 8 /* bucketed path: run ln_canonicalize behind the build (the atomic path numbers canonically anyway) */
#define LN_BUILD_SORTED_CSR 16    /* rewrite every token list of `csr` in ascending token order behind the build: the order tokens arrive in
                                     differs from run to run, and with it the last bits of every fp32 sum over them.  With LnCsr.dense & 2
                                     in the reduces: run-to-run identical splat values / slice and gather gradients */
int ln_build_splat(const LnTable* t, const float* positions_raw, const float* sigmas_host, int n, int* idx, float* w,
                   int flags, const LnCsr* csr, void* workspace, size_t workspace_bytes, float* clear_values,
                   long long clear_values_elems, void* stream);

/* Vertex numbering.  The reference numbers vertices in thread-arrival order (atomicAdd(m_nr_filled), HashTableGPU.cuh:454) — it
 * differs from run to run and nothing downstream depends on it.  The bucketed build numbers them bucket by bucket (inside a
 * bucket by the rank of the vertex's smallest token: deterministic, and the whole build is two launches).  ln_canonicalize relabels
 * the rows of a table that ONE bucketed build has just produced — entries[], keys[] and, when given, the idx[tokens] that
 * build wrote and the row ids in the segment descriptors of `csr` — into first-occurrence order over
 * (point, remainder), i.e. the numbering a serial run of HashTableGPU::insert (HashTableGPU.cuh:425-484) produces and the golden vectors hold.  It must run before anything that
 * stores row ids elsewhere (accumulated values, neighbour lists).  workspace: ln_build_workspace_bytes(tokens, capacity).
 * LN_BUILD_CANONICAL_ROWS makes ln_build_splat / ln_distribute do this themselves.
 * `csr` is the LnCsr that build filled: a bucketed build writes ROW ids into its segment descriptors, so it has to be relabelled
 * with the table.  Passing NULL is only valid when that CSR is never used again (the caller discards it): a scatter through the
 * stale descriptors would land on the old rows. */
int ln_canonicalize(const LnTable* t, int* idx, long long tokens, const LnCsr* csr, void* workspace, size_t workspace_bytes, void* stream);

/* A host may let a build hash into fewer slots than the table owns (LnTable.capacity = the slots in use: clearing and emitting
 * the slot range then costs what the cloud needs, not what the cfg's hash_table_capacity reserves — slot positions are not
 * reference-visible, only row ids are).  ln_rehash re-inserts the existing vertices (rows 0 .. nr_filled-1 of keys[]) into the
 * slot range [0, t->capacity) of the same buffers — call it with the larger capacity before an incremental build that may
 * outgrow the smaller range.  Rows, keys and counters are untouched. */
int ln_rehash(const LnTable* t, void* stream);

/* splatCacheNaive (LatticeGPU.cuh:926-973): table_values[idx] += vals * w. */
int ln_splat_accumulate(float* table_values, const float* vals, const int* idx, const float* w, int n, int pos_dim,
                        int val_dim, void* stream);

/* Scatter without per-element global atomics.  ln_csr_build transposes splat indices idx[tokens]
 * (row per token, <0 = skip) into an LnCsr whose groups are rows (groups_upper = rows_upper).
 * ln_csr_reduce_rows then ADDS, for every group g with row r = grp_row ? grp_row[g] : g,
 *     dst[r, j] += sum_{t in g} src[(t / src_div) * src_stride + j] * w[t],   j < val_dim
 * (dst zero-initialised by the caller): one lane group per segment, plain stores for groups that
 * fit one segment, global atomics only to combine the segments of longer ones.
 * With src_div = d+1, src_stride = V it replaces splatCacheNaive (LatticeGPU.cuh:926-973) and
 * slice_backwards_..._no_homogeneous (LatticeGPU.cuh:3540-3623); with src_div = 1,
 * src_stride = V+1 it replaces gather_backwards_with_precomputation (LatticeGPU.cuh:3761-3817). */
long long ln_csr_max_segments(long long tokens, int groups_upper);
size_t ln_csr_workspace_bytes(long long tokens, int groups_upper);
int ln_csr_build(const int* idx, long long tokens, int groups_upper, const LnCsr* csr, void* workspace, size_t workspace_bytes,
                 void* stream);
int ln_csr_reduce_rows(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, const float* w, int val_dim,
                       int src_div, int src_stride, float* dst, void* stream);
/* Rewrites every token list of a CSR (ln_csr_build's, or a build's: groups_upper = what it was built over) in ascending token order:
 * what LN_BUILD_SORTED_CSR does behind a build.  workspace: `tokens` ints (tokens = what the CSR was sized for). */
int ln_csr_sort(const LnCsr* csr, int groups_upper, void* workspace, size_t workspace_bytes, long long tokens, void* stream);

/* The two launches that follow a splat build and do not depend on each other, as ONE launch: ln_csr_reduce_rows(csr,
 * grp_row, .., dst) (splatCacheNaive) in the first workgroups, ln_neighbours(table, query_rows_upper, table, same level,
 * dilation 1, no flip, nbr) in the rest. */
int ln_splat_accumulate_and_neighbours(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, const float* w,
                                       int val_dim, int src_div, int src_stride, float* dst, const LnTable* table, int query_rows_upper,
                                       int* nbr, void* stream);

/* "Next" row: the vertex-wise aggregations PointNetModule runs on the distributed
 * rows — torch_scatter.scatter_max with argmax (lattice_modules.py:688) and scatter_add of ones
 * (lattice_modules.py:692) — on the same adjacency.
 * ln_csr_segment_max: out_max[r, c] = max over the tokens t of row r of src[t, c], out_arg[r, c] = the
 * token attaining it (smallest token on ties); rows without tokens get 0 / -1.
 * packed_ws: rows*channels*8 bytes of scratch.  ln_csr_group_sizes: counts[r] = tokens of row r. */
int ln_csr_segment_max(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, int channels, int rows,
                       void* packed_ws, float* out_max, int* out_arg, void* stream);
int ln_csr_group_sizes(const LnCsr* csr, const int* grp_row, int groups_upper, int rows, int* counts, void* stream);

/* distribute kernel (LatticeGPU.cuh:534-650) behind Lattice::distribute (Lattice.cu:351-410):
 * ln_build_splat + the dense rows [pos_scaled(d) | val(V) | bary] -> distributed[n*(d+1), d+V+1].
 * flags: LN_BUILD_CLEAR_FIRST / LN_BUILD_ATOMIC_PATH as for ln_build_splat (idx / w are always written). */
int ln_distribute(const LnTable* t, const float* positions_raw, const float* sigmas_host, const float* vals, int n,
                  int val_dim, int* idx, float* w, float* distributed, int flags, const LnCsr* csr, void* workspace,
                  size_t workspace_bytes, float* clear_values, long long clear_values_elems, void* stream);

/* coarsen kernel (LatticeGPU.cuh:2314-2514) behind Lattice::create_coarse_verts (Lattice.cu:670-703).
 * fine_rows_upper bounds the launch; the kernel also honours *fine->nr_filled. */
int ln_coarsen(const LnTable* fine, int fine_rows_upper, const LnTable* coarse, const LnCsr* csr, void* workspace,
               size_t workspace_bytes, void* stream);

/* Neighbour traversal shared by im2row / im2rowindices / row2im (LatticeGPU.cuh:1479-1684,
 * 1844-1915, 2187-2284): nbr[query_rows_upper, E] (E = 2(d+1)+1) rows of the neighbour table;
 * -1 = visited-but-absent, LN_NOT_VISITED = never looked at.  Slot layout as the reference:
 * axis a: np -> 2a+(flip), nm -> 2a+(1-flip); centre -> E-1. */
int ln_neighbours(const LnTable* query, int query_rows_upper, const LnTable* neigh, int lvl_query, int lvl_neigh,
                  int dilation, int flip, int* nbr, void* stream);

/* im2row (LatticeGPU.cuh:1464-1688) from a neighbour list: out[m, E*V]. */
int ln_im2row(const int* nbr, const float* values_neigh, int m, int filter_extent, int val_dim, float* out, void* stream);
/* im2rowindices (LatticeGPU.cuh:1690-1920): out[m, E*V] int32, reference fill rules. */
int ln_im2rowindices(const int* nbr, int m, int filter_extent, int val_dim, int* out, void* stream);
/* row2im (LatticeGPU.cuh:2067-2305): out[m, V] = adjoint gather of rowified[m_rows, E*V];
 * `nbr` is the un-flipped list of the OUTPUT lattice against the lattice indexing `rowified`. */
int ln_row2im(const int* nbr, const float* rowified, int m, int filter_extent, int val_dim, float* out, void* stream);

/* Lattice::convolve_im2row_standalone (Lattice.cu:424-474) without materialising the rowified
 * tensor: out[m, F] = sum_e values_neigh[nbr[m,e'], :] @ B_e   (fp32 MFMA), with
 *   flags & LN_CONV_FLIP_NEIGHBOURS   : e' = e^1 for the neighbour slots (what the reference's
 *        flip_neighbours traversal produces, LatticeGPU.cuh:1622-1626) — lets the backward pass reuse
 *        the forward neighbour list;
 *   flags & LN_CONV_TRANSPOSED_FILTER : `filter` is the [E*F, V] bank of the convolution being
 *        differentiated and B_e is its per-slot transpose, i.e. the filter_bank_backwards of
 *        lattice_funcs.py:307-311 without building it; otherwise `filter` is [E*V, F] and
 *        B_e = filter[e*V:(e+1)*V, :].
 * filter_extent = 1 with nbr = 0..m-1 is a per-vertex linear layer (the 1x1 blocks of lattice_modules.py:806-832):
 * out = values @ W^T for W = `filter` [F, V] under LN_CONV_TRANSPOSED_FILTER. */
#define LN_CONV_FLIP_NEIGHBOURS 1
#define LN_CONV_TRANSPOSED_FILTER 2
/* ln_conv_forward_ws only: the first ln_conv_bank_workspace_bytes(...) bytes of `workspace` still hold what an earlier call with the
 * SAME filter contents, sizes (m, filter_extent, val_dim, nr_filters) and flags left there (the filter split into bf16 parts in the
 * layout of the kernel those sizes select): the split launch is skipped.  For hosts that convolve repeatedly with unchanged
 * weights (inference; several clouds per optimizer step) and keep one workspace per layer. */
#define LN_CONV_BANK_READY 4
int ln_conv_forward(const int* nbr, const float* values_neigh, const float* filter, int m, int filter_extent, int val_dim,
                    int nr_filters, int flags, float* out, void* stream);
/* ln_conv_forward with scratch.  When the lattice has few vertices (coarse levels of a U-Net: fewer vertex tiles than CUs) the
 * contraction is split over the filter slots and the partial sums are added by a second launch; that needs
 * ln_conv_forward_workspace_bytes(m, filter_extent, val_dim, nr_filters) bytes of 16-byte aligned scratch (256 when no
 * split applies).  Without enough scratch it runs unsplit, exactly as ln_conv_forward. */
size_t ln_conv_bank_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters); /* the split-bank part (0: no split bank for these sizes) */
size_t ln_conv_forward_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters);
int ln_conv_forward_ws(const int* nbr, const float* values_neigh, const float* filter, int m, int filter_extent, int val_dim,
                       int nr_filters, int flags, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* grad_filter = rowified^T @ grad_out (lattice_funcs.py:302) without the rowified tensor.
 * workspace: ln_conv_grad_filter_workspace_bytes(). */
size_t ln_conv_grad_filter_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters);
int ln_conv_grad_filter(const int* nbr, const float* values_neigh, const float* grad_out, int m, int filter_extent,
                        int val_dim, int nr_filters, float* grad_filter, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Work placement hint for the vertex-tiled convolution kernels (ln_conv_forward*, ln_conv_backward) of the CALLING THREAD: `row_starts`
 * = LnTable.row_regions of the space-ordered table whose rows the following calls convolve over (device memory, 9 ints, read by the
 * kernels at run time), or NULL to reset.  XCD x then works on the row tiles of kd region x, whose neighbour rows share its L2.
 * Purely a speed matter: any contents give correct results. */
int ln_conv_row_partition(const int* row_starts);

/* slice_with_precomputation (LatticeGPU.cuh:2552-2595). */
int ln_slice_forward(const float* values, const int* idx, const float* w, int n, int pos_dim, int val_dim, float* out,
                     void* stream);
/* ln_slice_forward that also zero-fills `grad_accumulator` (grad_accumulator_elems floats): the [rows, val_dim] buffer the
 * backward pass of this slice scatters into (ln_csr_reduce_rows / ln_slice_backward need it zeroed), saving that fill launch. */
int ln_slice_forward_prepare_backward(const float* values, const int* idx, const float* w, int n, int pos_dim, int val_dim, float* out,
                                      float* grad_accumulator, long long grad_accumulator_elems, void* stream);
/* slice_with_precomputation for the indices a build of `t` wrote, with the points taken in the order of that build's slot CSR (`csr`,
 * as ln_build_splat / ln_distribute filled it; groups = hash slots) instead of input order: over a space-ordered table
 * (LnTable.slot_map) the d+1 value rows of the points a workgroup slices then sit in one kd region, i.e. in one XCD's L2.  Same
 * arithmetic per point, bit-identical output rows.  grad_accumulator (may be NULL) as ln_slice_forward_prepare_backward.  Widths the
 * ordered kernel does not cover run ln_slice_forward. */
int ln_slice_forward_ordered(const LnTable* t, const LnCsr* csr, const float* values, const int* idx, const float* w, int n, int val_dim,
                             float* out, float* grad_accumulator, long long grad_accumulator_elems, void* stream);

/* slice_no_precomputation (LatticeGPU.cuh:2598-2750): also writes idx/w (-1 where absent). */
int ln_slice_no_precomputation(const LnTable* t, const float* values, const float* positions_raw, const float* sigmas_host,
                               int n, int val_dim, float* out, int* idx, float* w, void* stream);
/* slice_backwards_with_precomputation_no_homogeneous (LatticeGPU.cuh:3540-3623);
 * grad_values[m_rows, V] must be zeroed by the caller (Lattice.cu:1079). */
int ln_slice_backward(const float* grad_sliced, const int* idx, const float* w, int n, int pos_dim, int val_dim,
                      float* grad_values, void* stream);

/* gather_with_precomputation (LatticeGPU.cuh:2886-2929): out[n, (d+1)(V+1)] (fully written). */
int ln_gather_forward(const float* values, const int* idx, const float* w, int n, int pos_dim, int val_dim, float* out,
                      void* stream);
/* gather_backwards_with_precomputation (LatticeGPU.cuh:3761-3817); grad_values zeroed by caller. */
int ln_gather_backward(const float* grad_gathered, const int* idx, const float* w, int n, int pos_dim, int val_dim,
                       float* grad_values, void* stream);

/* slice_classify_with_precomputation (LatticeGPU.cuh:3387-3464): logits[n, C] (fully written). */
int ln_slice_classify_forward(const float* values, const float* delta_w, const float* lin_w, const float* lin_b,
                              const int* idx, const float* w, int n, int pos_dim, int val_dim, int nr_classes,
                              float* logits, void* stream);
/* slice_classify_backwards_with_precomputation (LatticeGPU.cuh:3628-3756).  g_delta_w, g_lin_w, g_lin_b are
 * caller-allocated, zeroed (lattice_funcs.py:554-557) and accumulated into.  The gradient wrt the lattice values
 * is a scatter of grad_sliced[n, V] (= dL/d sliced features, written here) with weights w_eff[n*(d+1)] (= w +
 * delta_w, written here): pass g_values = NULL and run ln_csr_reduce_rows(csr, grp_row, .., grad_sliced, w_eff, V,
 * d+1, V, g_values) on the adjacency of `idx` (no atomics), or pass g_values (zeroed [M, V]) to have it scattered
 * here with global atomics.  workspace: ln_slice_classify_backward_workspace_bytes. */
size_t ln_slice_classify_backward_workspace_bytes(int n, int pos_dim, int val_dim, int nr_classes);
int ln_slice_classify_backward(const float* grad_logits, const float* values, const float* delta_w, const float* lin_w,
                               const int* idx, const float* w, int n, int pos_dim, int val_dim, int nr_classes,
                               float* g_values, float* g_delta_w, float* g_lin_w, float* g_lin_b, float* grad_sliced,
                               float* w_eff, void* workspace, size_t workspace_bytes, void* stream);

/* Both gradients of out = ln_conv_forward(nbr_q, values_neigh, filter[E*val_dim, nr_filters]) in one call:
 * grad_filter as ln_conv_grad_filter(nbr_q, values_neigh, grad_out, mq, ..), grad_values[mn, val_dim] as
 * ln_conv_forward(nbr_n, grad_out, filter, mn, E, nr_filters, val_dim, FLIP | TRANSPOSED_FILTER) where nbr_n is the
 * neighbour list with the query / neighbour roles swapped.  For the small-filter shapes the slab sum of the filter
 * gradient runs inside the value-gradient launch.  workspace: ln_conv_grad_filter_workspace_bytes(mq, ..). */
int ln_conv_backward(const int* nbr_q, const int* nbr_n, const float* values_neigh, const float* grad_out, const float* filter, int mq, int mn,
                     int filter_extent, int val_dim, int nr_filters, float* grad_values, float* grad_filter, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Half-precision feature path of the scatters and the slice (C5: "features fp16, accumulate fp32"): the segment reduce with
 * fp16 source rows (fp32 weights, fp32 accumulation into an fp32 dst) — plain and fused with the neighbour traversal — and the
 * slice of fp16 lattice values into fp16 rows (fp32 arithmetic, val_dim % 4 == 0). */
int ln_csr_reduce_rows_f16(const LnCsr* csr, const int* grp_row, long long max_segments, const void* src_f16, const float* w, int val_dim,
                           int src_div, int src_stride, float* dst, void* stream);
int ln_splat_accumulate_and_neighbours_f16(const LnCsr* csr, const int* grp_row, long long max_segments, const void* src_f16, const float* w,
                                           int val_dim, int src_div, int src_stride, float* dst, const LnTable* table, int query_rows_upper,
                                           int* nbr, void* stream);
int ln_slice_forward_f16(const void* values_f16, const int* idx, const float* w, int n, int pos_dim, int val_dim, void* out_f16, void* stream);
/* ... that also zero-fills the fp32 accumulator of this slice's backward pass (as ln_slice_forward_prepare_backward). */
int ln_slice_forward_f16_prepare_backward(const void* values_f16, const int* idx, const float* w, int n, int pos_dim, int val_dim, void* out_f16,
                                          float* grad_accumulator, long long grad_accumulator_elems, void* stream);

/* Half-precision feature path of the convolution (BASELINE.json config 5 / SURVEY.md 8d C5: features fp16, accumulate
 * fp32).  Same arguments and flags as ln_conv_forward / ln_conv_grad_filter; values, filter bank, grad_out and out are
 * IEEE fp16 (`_Float16`), accumulation is fp32 (v_mfma_f32_16x16x16_f16 for val_dim in {16,32,64,96,128,256} and
 * nr_filters % 16 == 0; any other shape on a scalar kernel), the filter gradient is returned in fp32. */
int ln_conv_forward_f16(const int* nbr, const void* values_neigh, const void* filter, int m, int filter_extent, int val_dim,
                        int nr_filters, int flags, void* out, void* stream);
size_t ln_conv_grad_filter_f16_workspace_bytes(int m, int filter_extent, int val_dim, int nr_filters);
int ln_conv_grad_filter_f16(const int* nbr, const void* values_neigh, const void* grad_out, int m, int filter_extent, int val_dim,
                            int nr_filters, float* grad_filter, void* workspace, size_t workspace_bytes, void* stream);

/* "Next" row: the per-token MLP of PointNetModule (lattice_modules.py:636-676) on the distributed rows —
 * y[rows, cout] = act(x[rows, cin] @ w[cout, cin]^T + b), act = LeakyReLU(slope) (slope < 0: identity).  Streaming
 * kernels for rows ~ 10^5..10^6 and cin, cout <= 128, where a BLAS GEMM with K = 4..32 is far off the memory roofline
 * (float4 lanes when the channel count is a multiple of 4, scalar lanes otherwise).  backward: grad_x may be NULL (inputs
 * that need no gradient), grad_b may be NULL; `y` is the forward output (the activation mask is its sign);
 * cin * cout <= 10240. */
int ln_linear_act_forward(const float* x, const float* w, const float* b, long long rows, int cin, int cout, float slope, float* y,
                          void* stream);
size_t ln_linear_act_backward_workspace_bytes(int cin, int cout);
int ln_linear_act_backward(const float* x, const float* w, const float* y, const float* grad_y, long long rows, int cin, int cout,
                           float slope, float* grad_x, float* grad_w, float* grad_b, void* workspace, size_t workspace_bytes, void* stream);

/* "Next" row: GroupNorm (+ optional fused ReLU) of the LNN blocks on the native [m, channels]
 * value layout (lattice_modules.py:585-616 runs torch.nn.GroupNorm on a transposed [1, C, M] view).  Statistics
 * per group over (all rows) x (channels of the group), biased variance, as torch.nn.GroupNorm.
 * forward:  y = act(x * a[c] + b[c]),  a = gamma*rstd[g], b = beta - mean[g]*a;  also writes mean_rstd[2*groups]
 *           (means then rstds) and scale_shift[2*channels] (a then b) for the backward pass.
 * backward: grad_x (and grad_gamma / grad_beta when non-NULL) from x, grad_y and the two saved vectors.
 * channels % 4 == 0, channels <= 1024, 16-byte aligned tensors; workspace = ln_group_norm_workspace_bytes(channels).
 * next_workspace (may be NULL): a caller that alternates two workspaces on one stream passes the other one here; the
 * call then trusts `workspace` to be all-zero (no fill launch) and zero-fills next_workspace_bytes of `next_workspace` (what
 * its last user dirtied) before it returns the GPU. */
size_t ln_group_norm_workspace_bytes(int channels);
int ln_group_norm_forward(const float* x, const float* gamma, const float* beta, int m, int channels, int groups, float eps, int relu,
                          float* y, float* mean_rstd, float* scale_shift, void* workspace, size_t workspace_bytes, void* next_workspace,
                          size_t next_workspace_bytes, void* stream);
int ln_group_norm_backward(const float* x, const float* grad_y, const float* gamma, const float* mean_rstd, const float* scale_shift, int m,
                           int channels, int groups, int relu, float* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                           size_t workspace_bytes, void* next_workspace, size_t next_workspace_bytes, void* stream);
/* The same with a device-side row count (static-rows mode: the [m, channels] tensors are taller than the lattice they hold):
 * rows_device == NULL, or one device int — the statistics run over min(m, *rows_device) rows, rows beyond are written as zeros
 * (y, grad_x) and contribute to nothing.  Pass LnTable.nr_filled of the lattice the values belong to. */
int ln_group_norm_forward_rows(const float* x, const float* gamma, const float* beta, int m, int channels, int groups, float eps, int relu,
                               float* y, float* mean_rstd, float* scale_shift, void* workspace, size_t workspace_bytes, void* next_workspace,
                               size_t next_workspace_bytes, const int* rows_device, void* stream);
int ln_group_norm_backward_rows(const float* x, const float* grad_y, const float* gamma, const float* mean_rstd, const float* scale_shift, int m,
                                int channels, int groups, int relu, float* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                                size_t workspace_bytes, void* next_workspace, size_t next_workspace_bytes, const int* rows_device, void* stream);

/* ---- max-centring of the gathered simplex rows in the DeformSlice head ------------------------------------------
 * Replaces the torch broadcasting at lattice_modules.py:525-529 (`rowified -= gamma * max_vals + beta`, max over the
 * d+1 vertices of a simplex) and its autograd backward, whose two reductions over the N points dominate it.
 * x, out, grad_out, grad_x: [n, k, c] (k = d+1 <= 8 vertex rows of c <= 64 values per point); gamma, beta: [c].
 * Forward also emits max_vals [n, c] and arg_max [n, c] (u8, first maximum) for the backward pass.
 * Backward: grad_x = grad_out - [k == arg_max] * gamma * sum_k grad_out;  grad_gamma_beta [2, c]: row 0 =
 * -sum_n s*max, row 1 = -sum_n s with s = sum_k grad_out (summed in a fixed order: deterministic). */
int ln_max_centre_forward(const float* x, const float* gamma, const float* beta, long long n, int k, int c, float* out,
                          float* max_vals, unsigned char* arg_max, void* stream);
size_t ln_max_centre_backward_workspace_bytes(long long n, int k, int c);
int ln_max_centre_backward(const float* grad_out, const float* max_vals, const unsigned char* arg_max, const float* gamma,
                           long long n, int k, int c, float* grad_x, float* grad_gamma_beta, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Both gradients of a per-row linear layer y = x w^T (x [rows, cin], w [cout, cin]: the 1 x 1 layers of the blocks, run as lattice
 * convolutions over the identity neighbour list `ident` [rows, 1] = 0 .. rows-1): grad_w [cout, cin] and grad_x [rows, cin] (may be
 * NULL).  One call so that the slab sum of the weight gradient rides in the bank split of the input gradient's convolution. */
size_t ln_linear_backward_workspace_bytes(int rows, int cin, int cout);
int ln_linear_backward(const int* ident, const float* x, const float* grad_y, const float* w, int rows, int cin, int cout, float* grad_x,
                       float* grad_w, void* workspace, size_t workspace_bytes, void* stream);

/* ---- launch diet of the glue around the lattice operators in a training step ----------------------------------------------
 * Weight normalisation of the reference's weight_norm_wrapper with v_dim=None (latticenet_py/lattice/utils.py:72-158; LinearWN,
 * ConvLatticeIm2RowWN, CoarsenLatticeWN, FinefyLatticeWN):  w = v * g / ||v||_F  for v [rows, cols] and one magnitude per row
 * (g_dim 0, g [rows]) or per column (g_dim 1, g [cols]); at most 1024 magnitudes, rows * cols <= 2^24.  One workgroup each way,
 * sums in a fixed order.  `norm`: one float written by the forward call and read by the backward call.
 *   grad_g[j] = sum_k grad_w[j,k] v[j,k] / n      grad_v = grad_w * g / n - v * (sum_j g[j] sum_k grad_w[j,k] v[j,k]) / n^3 */
int ln_weight_norm_forward(const float* v, const float* g, int rows, int cols, int g_dim, float* w, float* norm, void* stream);
int ln_weight_norm_backward(const float* v, const float* g, const float* grad_w, const float* norm, int rows, int cols, int g_dim,
                            float* grad_v, float* grad_g, void* stream);

/* DistributeLatticeModule's per-token tail (lattice_modules.py:72-94) in one pass: distributed / out [tokens, width] rows whose
 * first pos_dim columns are positions; position_sums [m, pos_dim] and counts [m] per vertex (ln_csr_reduce_rows of the positions
 * with unit weights, ln_csr_group_sizes);  out[t, :pos_dim] = d[t, :pos_dim] - sums[idx[t]] / max(counts[idx[t]], 1), the other
 * columns copied; rows of tokens with idx[t] <= 0 (no vertex, or vertex 0 = the "invalid" bucket) are zero. */
int ln_distribute_centre(const float* distributed, const int* splat_idx, const float* position_sums, const int* counts, long long tokens,
                         int width, int pos_dim, float* out, void* stream);

/* The vertex-side reduction of PointNetModule (lattice_modules.py:688-712: scatter_max, scatter_add of ones, index_select of the
 * winning tokens' barycentric weights, cat, masked_fill(nr_points < 4), keep mask of vertex 0) over the token adjacency `csr`:
 *   out [rows, 2 * channels] = [max over the row's tokens of src[t, :] | bary[argmax token * bary_stride]]
 *   out_arg [rows, channels] = winning token, -1 where the row is dropped (fewer than min_points tokens, row 0) or has no token.
 * Backward wrt src, token-major (every element of grad_src [tokens, channels] is written: no fill, no scatter):
 *   grad_src[t, c] = grad_out[idx[t] * grad_stride + c] if out_arg[idx[t], c] == t else 0. */
size_t ln_pointnet_reduce_workspace_bytes(int rows, int channels);
int ln_pointnet_reduce_forward(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, int channels,
                               const float* bary, int bary_stride, int rows, int min_points, void* workspace, size_t workspace_bytes,
                               float* out, int* out_arg, void* stream);
int ln_pointnet_reduce_backward(const float* grad_out, int grad_stride, const int* out_arg, const int* splat_idx, long long tokens,
                                int channels, float* grad_src, void* stream);

/* Mean negative log-likelihood of the training loop (ln_train.py:130, torch.nn.NLLLoss(ignore_index=...)): log_probs [n, classes]
 * float, target [n] int64; ignore_index: a label value that is skipped (pass a value no label takes, e.g. LLONG_MIN, for none).
 * loss_count [2]: {loss = -sum lp[i, y_i] / max(count, 1), max(count, 1)} (count = labels not ignored); labels outside
 * [0, classes) are clamped, as in the gather formulation it replaces.  Backward writes every element of grad_log_probs [n, classes]:
 * -grad_loss / count at (i, y_i) of the labels that count, 0 elsewhere.  Sums in a fixed order (deterministic). */
size_t ln_nll_workspace_bytes(void);
int ln_nll_forward(const float* log_probs, const long long* target, long long n, int classes, long long ignore_index, void* workspace,
                   size_t workspace_bytes, float* loss_count, void* stream);
int ln_nll_backward(const long long* target, const float* grad_loss, const float* loss_count, long long n, int classes,
                    long long ignore_index, float* grad_log_probs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LATTICENET_HIP_H */
