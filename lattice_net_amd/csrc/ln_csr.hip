// Vertex -> contributions adjacency (CSR) and the scatter it enables without per-element atomics.
//
// The reference scatters point rows onto lattice vertices with one global fp32 atomicAdd per
// (point, simplex vertex, channel): splatCacheNaive (LatticeGPU.cuh:937-971), slice backward
// (LatticeGPU.cuh:3574-3613), gather backward (LatticeGPU.cuh:3778-3814).  On MI355X those
// N*(d+1)*V device-scope float atomics are executed at the memory side and dominate the whole
// path (291 us per launch at N=120k, V=32).  Here the splat indices are transposed ONCE per
// lattice build into CSR form (row_start[M+1], csr_tok[T]: the tokens p*(d+1)+r that touch each
// vertex), costing one int atomic per token, and every scatter becomes a gather-reduce.
//
// Vertex degrees are heavily skewed (a LiDAR scan puts thousands of points on the vertices next to
// the sensor), so the unit of work is a SEGMENT: at most LN_SEG consecutive CSR entries of one row.
// A lane group reduces one segment in registers; rows that fit one segment are written with a plain
// store, longer rows combine their segments with global atomicAdd (few, and only on hot rows).
#include "ln_csr.h"
#include "ln_neighbours.h"

#define LN_SCAN_BLOCK 1024
#define LN_CSR_LDS_SCAN_BLOCKS 512  // up to this many row blocks (512k rows) the top-level scan is redone in LDS by every workgroup
#define LN_SEG LN_CSR_SEG

__global__ void __launch_bounds__(256)
    k_csr_count(const int* __restrict__ idx, long long tokens, int rows_upper, int* __restrict__ cnt, int* __restrict__ pos) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tokens) return;
    const int row = idx[t];
    pos[t] = (row >= 0 && row < rows_upper) ? atomicAdd(&cnt[row], 1) : -1;
}

__device__ __forceinline__ int ln_wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// exclusive scans inside blocks of 1024 rows of (a) the token counts and (b) the segment counts
__global__ void __launch_bounds__(LN_SCAN_BLOCK)
    k_csr_scan_local(const int* __restrict__ cnt, int rows_upper, int* __restrict__ local_tok, int* __restrict__ local_seg,
                     int* __restrict__ block_tot) {
    __shared__ int s_tok[16];
    __shared__ int s_seg[16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = blockIdx.x * LN_SCAN_BLOCK + tid;
    const int c = (i < rows_upper) ? cnt[i] : 0;
    const int sg = (c + LN_SEG - 1) / LN_SEG;
    const int ic = ln_wave_incl_scan(c, lane);
    const int is = ln_wave_incl_scan(sg, lane);
    if (lane == 63) {
        s_tok[wave] = ic;
        s_seg[wave] = is;
    }
    __syncthreads();
    int off_t = 0, off_s = 0;
    for (int k = 0; k < wave; ++k) {
        off_t += s_tok[k];
        off_s += s_seg[k];
    }
    if (i < rows_upper) {
        local_tok[i] = off_t + ic - c;
        local_seg[i] = off_s + is - sg;
    }
    if (tid == LN_SCAN_BLOCK - 1) {
        block_tot[2 * blockIdx.x] = off_t + ic;
        block_tot[2 * blockIdx.x + 1] = off_s + is;
    }
}

// Exclusive scan of `n` ints of `src` (stride `stride`) into LDS array `dst[0..n]` (dst[n] = total),
// done redundantly by every workgroup that needs the top level of the two-level scan: n is the number
// of 1024-row blocks (98 for a 100k-slot table), so this is far cheaper than one more kernel launch.
__device__ __forceinline__ void ln_block_excl_scan_to_lds(const int* __restrict__ src, int stride, int n, int* dst, int* s_part) {
    const int tid = threadIdx.x;
    const int per = (n + 255) / 256;
    const int b = tid * per;
    int sum = 0;
    for (int k = 0; k < per; ++k)
        if (b + k < n) sum += src[(size_t)(b + k) * stride];
    int total;
    int run = ln_block_excl_scan_256(sum, s_part, &total);
    if (tid == 0) dst[n] = total;
    for (int k = 0; k < per; ++k)
        if (b + k < n) {
            dst[b + k] = run;
            run += src[(size_t)(b + k) * stride];
        }
    __syncthreads();
}

// Tables beyond LN_CSR_LDS_SCAN_BLOCKS row blocks (8M+ slots): one workgroup turns the per-block totals into exclusive
// offsets in place (block_tot[2b], block_tot[2b+1]; the grand totals go to block_tot[2nb], block_tot[2nb+1]) and
// k_csr_fill<true> reads them from global memory instead of re-scanning them into LDS in every workgroup.
__global__ void __launch_bounds__(256) k_csr_scan_top(int* __restrict__ block_tot, int nb) {
    __shared__ int s_part[8];
    const int tid = threadIdx.x;
    const int per = (nb + 255) / 256;
    const int b = tid * per;
    for (int which = 0; which < 2; ++which) {
        int sum = 0;
        for (int k = 0; k < per; ++k)
            if (b + k < nb) sum += block_tot[2 * (size_t)(b + k) + which];
        int total;
        int run = ln_block_excl_scan_256(sum, s_part, &total);
        for (int k = 0; k < per; ++k)
            if (b + k < nb) {
                const int v = block_tot[2 * (size_t)(b + k) + which];
                block_tot[2 * (size_t)(b + k) + which] = run;
                run += v;
            }
        if (tid == 0) block_tot[2 * (size_t)nb + which] = total;
        __syncthreads();
    }
}

template <bool GLOBAL_OFFSETS>
__global__ void __launch_bounds__(256)
    k_csr_fill(const int* __restrict__ idx, const int* __restrict__ pos, long long tokens, int rows_upper,
               const int* __restrict__ local_tok, const int* __restrict__ local_seg, const int* __restrict__ block_tot, int nb,
               const int* __restrict__ grp_cnt, int* __restrict__ row_start, int* __restrict__ csr_tok, int4* __restrict__ seg_desc,
               int* __restrict__ seg_count) {
    extern __shared__ int s_fill[];  // off_tok[nb+1] | off_seg[nb+1] | part[256]
    const int* off_tok;
    const int* off_seg;
    int seg_stride = 1;
    if constexpr (GLOBAL_OFFSETS) {
        off_tok = block_tot;
        off_seg = block_tot + 1;
        seg_stride = 2;
    } else {
        int* lt = s_fill;
        int* ls = s_fill + (nb + 1);
        int* part = s_fill + 2 * (nb + 1);
        ln_block_excl_scan_to_lds(block_tot, 2, nb, lt, part);
        ln_block_excl_scan_to_lds(block_tot + 1, 2, nb, ls, part);
        off_tok = lt;
        off_seg = ls;
    }
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {  // one region holds everything
        seg_count[0] = off_seg[(size_t)nb * seg_stride];
        for (int g = 1; g < LN_XCD_GROUPS; ++g) seg_count[g] = 0;
        seg_count[LN_XCD_GROUPS] = 1;
        seg_count[LN_XCD_GROUPS + 1] = 0;  // descriptors name groups, weights are read per token
    }
    if (t <= rows_upper) row_start[t] = (t < rows_upper) ? local_tok[t] + off_tok[(size_t)(t / LN_SCAN_BLOCK) * seg_stride] : off_tok[(size_t)nb * seg_stride];
    if (t < tokens) {
        const int p = pos[t];
        if (p >= 0) {
            const int row = idx[t];
            const int blk = row / LN_SCAN_BLOCK;
            const int rbeg = local_tok[row] + off_tok[(size_t)blk * seg_stride];
            csr_tok[rbeg + p] = int(t);
            if (p % LN_SEG == 0) {  // this token opens a segment
                const int sid = local_seg[row] + off_seg[(size_t)blk * seg_stride] + p / LN_SEG;
                seg_desc[sid] = make_int4(row, rbeg + p, grp_cnt[row] - p, p);
            }
        }
    }
}

static size_t ln_align256c(size_t x) { return (x + 255) & ~size_t(255); }

extern "C" long long ln_csr_max_segments(long long tokens, int groups_upper) {
    if (tokens < 0) tokens = 0;
    const long long g = tokens < groups_upper ? tokens : groups_upper;
    return g + tokens / LN_SEG + 1;
}

size_t ln_csr_scan_workspace_bytes(int groups_upper) {
    if (groups_upper < 1) groups_upper = 1;
    const size_t nb = (size_t)ln_div_up(groups_upper, LN_SCAN_BLOCK);
    return ln_align256c((size_t)groups_upper * 4) * 2 + ln_align256c((nb + 1) * 8);
}

int ln_csr_from_counts(const int* tok_grp, const int* tok_pos, long long tokens, const int* grp_cnt, int groups_upper,
                       const LnCsr& csr, void* workspace, size_t workspace_bytes, hipStream_t st) {
    LN_REQUIRE(workspace && workspace_bytes >= ln_csr_scan_workspace_bytes(groups_upper), LN_ERR_WORKSPACE, "csr scan workspace too small");
    const int nb = ln_div_up(groups_upper, LN_SCAN_BLOCK);
    char* p = static_cast<char*>(workspace);
    int* local_tok = reinterpret_cast<int*>(p);
    p += ln_align256c((size_t)groups_upper * 4);
    int* local_seg = reinterpret_cast<int*>(p);
    p += ln_align256c((size_t)groups_upper * 4);
    int* block_tot = reinterpret_cast<int*>(p);
    LN_LAUNCH("k_csr_scan_local", k_csr_scan_local, dim3(nb), dim3(LN_SCAN_BLOCK), 0, st, grp_cnt, groups_upper, local_tok, local_seg, block_tot);
    const long long work = (tokens > groups_upper + 1) ? tokens : (long long)groups_upper + 1;
    if (nb <= LN_CSR_LDS_SCAN_BLOCKS) {
        // every workgroup re-scans the (few) block totals into LDS: cheaper than one more launch
        const size_t lds = (size_t)(2 * (nb + 1) + 256) * sizeof(int);
        LN_LAUNCH("k_csr_fill", k_csr_fill<false>, dim3(ln_div_up(work, 256)), dim3(256), lds, st, tok_grp, tok_pos, tokens, groups_upper, local_tok,
                  local_seg, block_tot, nb, grp_cnt, csr.grp_start, csr.csr_tok, reinterpret_cast<int4*>(csr.seg_desc), csr.seg_count);
    } else {
        LN_LAUNCH("k_csr_scan_top", k_csr_scan_top, dim3(1), dim3(256), 0, st, block_tot, nb);
        LN_LAUNCH("k_csr_fill", k_csr_fill<true>, dim3(ln_div_up(work, 256)), dim3(256), 0, st, tok_grp, tok_pos, tokens, groups_upper, local_tok,
                  local_seg, block_tot, nb, grp_cnt, csr.grp_start, csr.csr_tok, reinterpret_cast<int4*>(csr.seg_desc), csr.seg_count);
    }
    return ln_check_launch("ln_csr_from_counts");
}

extern "C" size_t ln_csr_workspace_bytes(long long tokens, int groups_upper) {
    if (tokens < 1) tokens = 1;
    if (groups_upper < 1) groups_upper = 1;
    return ln_align256c((size_t)groups_upper * 4) + ln_align256c((size_t)tokens * 4) + ln_csr_scan_workspace_bytes(groups_upper);
}

extern "C" int ln_csr_build(const int* idx, long long tokens, int groups_upper, const LnCsr* csr, void* workspace,
                            size_t workspace_bytes, void* stream) {
    LN_REQUIRE(tokens >= 0 && groups_upper >= 1, LN_ERR_ARG, "ln_csr_build: bad sizes");
    LN_REQUIRE(csr && csr->grp_start && csr->seg_count && (tokens == 0 || (idx && csr->csr_tok && csr->seg_desc)), LN_ERR_ARG,
               "ln_csr_build: null buffer");
    LN_REQUIRE(workspace && workspace_bytes >= ln_csr_workspace_bytes(tokens, groups_upper), LN_ERR_WORKSPACE,
               "ln_csr_build: workspace too small");
    LN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, LN_ERR_WORKSPACE, "ln_csr_build: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* p = static_cast<char*>(workspace);
    int* cnt = reinterpret_cast<int*>(p);
    p += ln_align256c((size_t)groups_upper * 4);
    int* pos = reinterpret_cast<int*>(p);
    p += ln_align256c((size_t)(tokens < 1 ? 1 : tokens) * 4);
    if (ln_zero_async(cnt, (size_t)groups_upper * 4, st) != LN_OK) return ln_check_launch("ln_csr_build(memset)");
    if (tokens > 0)
        LN_LAUNCH("k_csr_count", k_csr_count, dim3(ln_div_up(tokens, 256)), dim3(256), 0, st, idx, tokens, groups_upper, cnt, pos);
    return ln_csr_from_counts(idx, pos, tokens, cnt, groups_upper, *csr, p, ln_csr_scan_workspace_bytes(groups_upper), st);
}

// ------------------------------------------------------------------------------------------
// Deterministic token order (ln_csr_sort_groups): the position of a token inside its group's list comes from an atomic counter
// (LDS in the bucket pass, global on the atomic path), i.e. from arrival order — a group's tokens are the same in every run, their
// order is not, and fp32 sums over them differ in the last bits.  One wave per group rewrites the list in ascending token order:
// rank of a token = tokens of the group that are smaller (all pairs, 64 at a time through shuffles), written through a scratch copy.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_csr_sort_groups(const int* __restrict__ grp_start, int groups, const int* __restrict__ csr_tok, int* __restrict__ sorted) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= groups) return;
    const int beg = grp_start[g], end = grp_start[g + 1];
    for (int i0 = beg; i0 < end; i0 += 64) {  // (wave-uniform trip counts)
        const int mine = i0 + lane < end ? csr_tok[i0 + lane] : 0x7FFFFFFF;
        int rank = 0;
        for (int j0 = beg; j0 < end; j0 += 64) {
            const int other = j0 + lane < end ? csr_tok[j0 + lane] : 0x7FFFFFFF;
            const int n = min(64, end - j0);
            for (int k = 0; k < n; ++k) {
                const int o = __shfl(other, k, 64);
                // ties (the -1 fillers behind an overflowed bucket) keep their relative order
                rank += (o < mine || (o == mine && j0 + k < i0 + lane)) ? 1 : 0;
            }
        }
        if (i0 + lane < end) sorted[beg + rank] = mine;
    }
}
__global__ void __launch_bounds__(256) k_csr_copy_tokens(const int* __restrict__ src, const int* __restrict__ total, int* __restrict__ dst) {
    const int n = *total;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}

// groups = the group count the CSR was built over; scratch: as many ints as the CSR has tokens
int ln_csr_sort_groups(const LnCsr& csr, int groups, int* scratch, hipStream_t st) {
    if (groups <= 0) return LN_OK;
    LN_LAUNCH("k_csr_sort_groups", k_csr_sort_groups, dim3(ln_div_up(groups, 4)), dim3(256), 0, st, csr.grp_start, groups, csr.csr_tok, scratch);
    LN_LAUNCH("k_csr_copy_tokens", k_csr_copy_tokens, dim3(1024), dim3(256), 0, st, scratch, csr.grp_start + groups, csr.csr_tok);
    return ln_check_launch("ln_csr_sort_groups");
}

extern "C" int ln_csr_sort(const LnCsr* csr, int groups_upper, void* workspace, size_t workspace_bytes, long long tokens, void* stream) {
    LN_REQUIRE(csr && csr->grp_start && csr->csr_tok && groups_upper >= 0 && tokens >= 0, LN_ERR_ARG, "ln_csr_sort: bad arguments");
    LN_REQUIRE(tokens == 0 || (workspace && workspace_bytes >= (size_t)tokens * sizeof(int)), LN_ERR_WORKSPACE, "ln_csr_sort: workspace too small");
    if (tokens == 0) return LN_OK;
    return ln_csr_sort_groups(*csr, groups_upper, static_cast<int*>(workspace), (hipStream_t)stream);
}

// dst[row, j] += sum over the row's tokens t of src[(t / src_div) * src_stride + j] * w[t]   (j < V)
// One lane group per segment; LN_SEG/U batches of U independent gathers per lane.  Segments of one
// group have consecutive ids, so neighbouring lane groups of a wave often hold partial sums of the
// same (hot) vertex: they are combined with a segmented shuffle reduction and only the head of each
// run writes — a plain store when the run covers the whole group, one atomicAdd otherwise.
struct LnReduceArgs {
    const int* grp_start;
    const int* csr_tok;
    const int4* seg_desc;
    const int* seg_count;
    const int* grp_row;
    const void* src;  // float rows, or _Float16 rows for the HALF instantiations (accumulation and dst stay fp32)
    const float* w;
    int chunks, lanes_per_seg, src_div, src_stride;
    float* dst;
    long long seg_region;
    int dbg_plain;  // experiment: plain stores instead of atomics (wrong sums, timing only)
    int deterministic;  // LnCsr.dense & 2: one lane group per ROW, tokens in CSR order, no atomics (run-to-run identical sums)
};

// one block of segments (256 threads).  Per segment: descriptor -> row -> batches of 4 tokens {4 ids} -> {4 weights, 4 row chunks}.
// Measured on MI355X (C3, 17.5 us): the kernel is insensitive to the length of this dependency chain — prefetching the next
// batch's ids with the current gathers (6 trips instead of 10): 18.4 us; all 16 ids + 8 gathers in flight (4 trips, 128
// VGPRs): 19.3 us; plain stores instead of the hot-vertex atomics: no change; 44 MB instead of 79 MB of L2 misses (kd
// regions, LnCsr.planes): no change; 71 instead of 104 SGPRs (8 instead of 6 workgroups per CU admitted): no change.
// WG: also combine runs across the wave boundaries of the workgroup (dense clouds: LnCsr.dense; two barriers per block and ten more
// registers, which cost the sparse C3 scan 10 % — hence a variant of its own, chosen by the host)
// L8 = 8: eight lanes per segment and eight chunks per row as COMPILE-TIME constants (32 fp32 channels on float4 lanes: the headline chain) —
// the lane / group arithmetic becomes shifts, the chunk loop and the shuffle rounds unroll; 0: both from the arguments.
template <int VEC, bool HALF, bool WG, int L8 = 0>
__device__ __forceinline__ void ln_reduce_chunk(const LnSegOfThread& so, const LnReduceArgs& a) {
    const int* __restrict__ csr_tok = a.csr_tok;
    const int4* __restrict__ seg_desc = a.seg_desc;
    const int* __restrict__ grp_row = a.grp_row;
    const float* __restrict__ src = static_cast<const float*>(a.src);
    const _Float16* __restrict__ src16 = static_cast<const _Float16*>(a.src);
    const float* __restrict__ w = a.w;
    float* __restrict__ dst = a.dst;
    const int chunks = L8 ? L8 : a.chunks, lanes_per_seg = L8 ? L8 : a.lanes_per_seg, src_div = a.src_div, src_stride = a.src_stride;
    constexpr int U = 4;
    const int lc = so.lane_in_seg;
    const int lane = threadIdx.x & 63;
    const int grp_in_wave = lane / lanes_per_seg;
    const int groups_per_wave = 64 / lanes_per_seg;
    const bool active = so.active;
    const bool rows_in_desc = a.seg_count[LN_XCD_GROUPS + 1] & 1;  // (uniform: a scalar load)
    int grp = -1, beg = 0, rbeg = 0, rend = 0, cnt = 0, row = -1;
    if (active) {
        const int4 d = seg_desc[so.sid];  // {group, first entry, entries to the end of the group, offset inside the group}
        grp = d.x;
        beg = d.y;
        cnt = min(LN_SEG, d.z);
        rbeg = beg - d.w;
        rend = beg + d.z;
        if (a.deterministic) {  // the lane group of a row's FIRST segment walks the whole row, in CSR order; the others stand by
            cnt = d.w == 0 ? d.z : 0;
            if (d.w != 0) grp = -1 - grp;  // (never equal to a neighbour's group: no combining)
        }
    }
    const int end = beg + cnt;
    // the descriptor names the row itself (bucketed build), a hash slot (row = entries[slot]) or a row-group (ln_csr_build)
    if (active) {
        const int g0 = grp >= 0 ? grp : -1 - grp;
        row = (rows_in_desc || !grp_row) ? g0 : grp_row[g0];
    }
    const int V = chunks * VEC;
    const int nchunk_iter = (chunks + lanes_per_seg - 1) / lanes_per_seg;  // wave-uniform trip count (shuffles inside)
    const bool pow2 = (src_div & (src_div - 1)) == 0;
    const int shift = __ffs(src_div) - 1;
    for (int it = 0; it < nchunk_iter; ++it) {
        const int c = lc + it * lanes_per_seg;
        const bool cok = active && row >= 0 && c < chunks && (!a.deterministic || cnt > 0);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        if (cok) {
            for (int e0 = beg; e0 < end; e0 += U) {
                // No branch per entry (round 6): the entries of the last, partly filled batch re-read the segment's last entry with
                // weight 0 (a row that is summed anyway), and an id of -1 (left behind by an overflowed build) reads row 0 with weight 0 —
                // the 4 ids, then the 4 weights and 4 row chunks go out back to back instead of each behind its own exec-mask test.
                int tk[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ok[u] = e0 + u < end;
                    tk[u] = csr_tok[min(e0 + u, end - 1)];
                }
                float wt[U];
                float x[U][VEC];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool valid = ok[u] && tk[u] >= 0;
                    const int t = max(tk[u], 0);
                    const float wl = w[t];
                    wt[u] = valid ? wl : 0.f;
                    {
                        const int srow = pow2 ? (t >> shift) : (t / src_div);
                        const size_t off = (size_t)srow * src_stride + c * VEC;
                        if constexpr (HALF) {
                            if constexpr (VEC == 8) {  // 16-byte words: a 64-channel fp16 row on 8 lanes, 8 segments per wave
                                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                                const h8 v8 = *reinterpret_cast<const h8*>(src16 + off);
#pragma unroll
                                for (int k = 0; k < 8; ++k) x[u][k] = (float)v8[k];
                            } else if constexpr (VEC == 4) {
                                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                                const h4 v4 = *reinterpret_cast<const h4*>(src16 + off);
                                x[u][0] = (float)v4[0]; x[u][1] = (float)v4[1]; x[u][2] = (float)v4[2]; x[u][3] = (float)v4[3];
                            } else {
                                x[u][0] = (float)src16[off];
                            }
                        } else if constexpr (VEC == 4) {
                            const float4 v4 = *reinterpret_cast<const float4*>(src + off);
                            x[u][0] = v4.x; x[u][1] = v4.y; x[u][2] = v4.z; x[u][3] = v4.w;
                        } else {
                            x[u][0] = src[off];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(x[u][k], wt[u], acc[k]);
            }
        }
        // segmented suffix reduction over the lane groups of this wave (runs of equal group id are contiguous)
        int run_end = end;
        for (int off = 1; off < groups_per_wave; off <<= 1) {
            const int delta = off * lanes_per_seg;
            const int o_grp = __shfl_down(grp, delta, 64);
            const int o_end = __shfl_down(run_end, delta, 64);
            float o_acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o_acc[k] = __shfl_down(acc[k], delta, 64);
            if (grp_in_wave + off < groups_per_wave && o_grp == grp && grp >= 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += o_acc[k];
                run_end = max(run_end, o_end);
            }
        }
        const int prev_grp = __shfl_up(grp, lanes_per_seg, 64);
        const bool head = (grp_in_wave == 0) || (prev_grp != grp);
        // ... and across the wave boundaries of the workgroup: a run that reaches the end of its wave takes in the FIRST run of the
        // next wave(s) when it is the same group.  On dense clouds (C5: 27 tokens per vertex) most rows have several segments and a
        // run cut by a wave boundary costs lanes x VEC float atomics per piece: plain stores instead of the atomics took the C5 splat
        // from 104 to 71 us (tools/probes/reduce_locality_probe.py), three of four boundaries are inside a workgroup.
        bool absorbed = false;
        if constexpr (WG) {
            __shared__ float s_first_acc[4][64 * VEC];  // the first run of every wave: [lane in segment][VEC]
            __shared__ int s_first_key[4], s_first_end[4], s_first_whole[4], s_last_key[4];
            const int wave = threadIdx.x >> 6;
            const int key = (active && row >= 0) ? grp : -1;
            const int first_key = __shfl(key, 0, 64);
            const int last_key = __shfl(key, (groups_per_wave - 1) * lanes_per_seg, 64);
            __syncthreads();  // (the arrays may still be read from the previous block of segments / channel chunk)
            if (grp_in_wave == 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) s_first_acc[wave][lc * VEC + k] = cok ? acc[k] : 0.f;
                if (lc == 0) {
                    s_first_key[wave] = key;
                    s_first_end[wave] = run_end;
                    s_first_whole[wave] = (first_key >= 0 && first_key == last_key) ? 1 : 0;
                    s_last_key[wave] = last_key;
                }
            }
            __syncthreads();
            absorbed = head && grp_in_wave == 0 && wave > 0 && key >= 0 && s_last_key[wave - 1] == key;
            if (head && key >= 0 && key == last_key && !absorbed) {
                for (int w2 = wave + 1; w2 < 4 && s_first_key[w2] == key; ++w2) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += s_first_acc[w2][lc * VEC + k];
                    run_end = max(run_end, s_first_end[w2]);
                    if (!s_first_whole[w2]) break;
                }
            }
        }
        if (cok && head && !absorbed) {
            float* d = dst + (size_t)row * V + c * VEC;
            if ((beg == rbeg && run_end == rend) || a.dbg_plain) {  // this run is the whole group: no other writer
                if constexpr (VEC == 8) {
                    *reinterpret_cast<float4*>(d) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    *reinterpret_cast<float4*>(d + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
                } else if constexpr (VEC == 4) {
                    *reinterpret_cast<float4*>(d) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else {
                    d[0] = acc[0];
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) __hip_atomic_fetch_add(d + k, acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// body of the segment reduce for workgroup `block_x` of `nblocks`
template <int VEC, bool HALF, bool WG, int L8 = 0>
__device__ __forceinline__ void ln_reduce_body(int block_x, int nblocks, const LnReduceArgs& a) {
    for (LnSegWalk wk(block_x, nblocks, L8 ? L8 : a.lanes_per_seg, a.seg_count, a.seg_region); wk.more(); wk.next())
        ln_reduce_chunk<VEC, HALF, WG, L8>(wk.here(), a);
}

template <int VEC, bool HALF, bool WG, int L8 = 0>
__global__ void __launch_bounds__(256) k_csr_reduce_segments(LnReduceArgs a) { ln_reduce_body<VEC, HALF, WG, L8>(blockIdx.x, gridDim.x, a); }

// Horizontal fusion of the two launches that follow a splat build and do not depend on each other: workgroups
// [0, reduce_blocks) accumulate the point features onto the vertices (segment reduce), the rest run the same-level
// neighbour traversal.  One launch instead of two, and the traversal's short, latency-bound workgroups fill the CUs
// the reduce's tail leaves idle.
template <int VEC, int D, bool HALF, bool WG, int L8 = 0>
__global__ void __launch_bounds__(256)
    k_reduce_and_neighbours(LnReduceArgs a, int reduce_blocks, LnTable t, int query_rows_upper, int* __restrict__ nbr) {
    if ((int)blockIdx.x < reduce_blocks) {
        ln_reduce_body<VEC, HALF, WG, L8>(blockIdx.x, reduce_blocks, a);
    } else {
        // whole vertices per workgroup, and over a space-ordered table the vertices of kd region x on XCD x: the 9 lookups of a vertex
        // land in the slot run of its own region, which then sits in ONE L2 instead of in all eight
        constexpr int E = 2 * (D + 1) + 1, ROWS = 256 / E;
        const LnSlotMap smap = ln_load_slot_map(t.slot_map);  // (first: its scalar loads travel with those of the partition)
        const int tile = ln_partition_tile((int)blockIdx.x - reduce_blocks, (int)gridDim.x - reduce_blocks, t.row_regions, ROWS);
        if ((int)threadIdx.x < ROWS * E)
            ln_neighbours_body<D>((long long)tile * (ROWS * E) + threadIdx.x, t, query_rows_upper, t, 1.0f, 1, 0, nbr, smap);
    }
}

// LN_REDUCE_L8=0: the general kernels for the 8-lane, 8-chunk shape too (A/B; read once)
static bool ln_reduce_l8() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LN_REDUCE_L8");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static int ln_reduce_args(const char* who, const LnCsr* csr, const int* grp_row, long long max_segments, const void* src, bool half,
                          const float* w, int val_dim, int src_div, int src_stride, float* dst, LnReduceArgs& a, int& vec, long long& work) {
    LN_REQUIRE(max_segments >= 0 && val_dim >= 1 && src_div >= 1 && src_stride >= val_dim, LN_ERR_ARG, "%s: bad sizes", who);
    LN_REQUIRE(max_segments == 0 || (csr && csr->grp_start && csr->csr_tok && csr->seg_desc && csr->seg_count && src && w && dst),
               LN_ERR_ARG, "%s: null buffer", who);
    const bool vec4 = (val_dim % 4 == 0) && (src_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & (half ? 7 : 15)) == 0) &&
                      ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    // fp16 rows of >= 64 channels: 16-byte words (8 halfs) per lane
    const bool vec8 = vec4 && half && val_dim % 8 == 0 && val_dim >= 64 && src_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    vec = vec8 ? 8 : (vec4 ? 4 : 1);
    const int chunks = val_dim / vec;
    int lanes = 1;
    while (lanes < chunks && lanes < 64) lanes <<= 1;
    work = max_segments * lanes;
    if (max_segments > 0) a = LnReduceArgs{csr->grp_start, csr->csr_tok, reinterpret_cast<const int4*>(csr->seg_desc), csr->seg_count, grp_row, src, w, chunks, lanes,
                                           src_div, src_stride, dst, csr->seg_region, (ln_debug_mask() & 128) ? 1 : 0, (csr->dense & 2) ? 1 : 0};
    return LN_OK;
}

static int ln_csr_reduce_rows_impl(const char* who, const LnCsr* csr, const int* grp_row, long long max_segments, const void* src, bool half,
                                   const float* w, int val_dim, int src_div, int src_stride, float* dst, void* stream) {
    LnReduceArgs a;
    int vec;
    long long work;
    int rc = ln_reduce_args(who, csr, grp_row, max_segments, src, half, w, val_dim, src_div, src_stride, dst, a, vec, work);
    if (rc) return rc;
    if (max_segments == 0) return LN_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ln_seg_grid(max_segments, a.lanes_per_seg)), block(256);
    // dense cloud, or wide rows (16+ lanes per segment: a wave then holds four segments or fewer, so every second to fourth segment of a
    // row is cut by a wave boundary — at 96 fp32 channels on the C3 cloud 50.2 -> 42.4 us): combine across the waves of a workgroup
    const bool wg = ((csr->dense & 1) != 0 || a.lanes_per_seg >= 16) && vec >= 4 && !(csr->dense & 2);
#define LN_REDUCE_LAUNCH(VV, HH)                                                                                                     \
    {                                                                                                                                \
        if (wg && VV == 8 && HH && a.lanes_per_seg == 8 && a.chunks == 8 && ln_reduce_l8())                                          \
            LN_LAUNCH("k_csr_reduce_segments", (k_csr_reduce_segments<8, true, true, 8>), grid, block, 0, st, a);                    \
        else if (wg)                                                                                                                 \
            LN_LAUNCH("k_csr_reduce_segments", (k_csr_reduce_segments<VV, HH, (VV >= 4)>), grid, block, 0, st, a);                   \
        else if (VV == 4 && !HH && a.lanes_per_seg == 8 && a.chunks == 8 && ln_reduce_l8())                                          \
            LN_LAUNCH("k_csr_reduce_segments", (k_csr_reduce_segments<4, false, false, 8>), grid, block, 0, st, a);                  \
        else                                                                                                                         \
            LN_LAUNCH("k_csr_reduce_segments", (k_csr_reduce_segments<VV, HH, false>), grid, block, 0, st, a);                       \
    }
    if (half) {
        if (vec == 8) LN_REDUCE_LAUNCH(8, true)
        else if (vec == 4) LN_REDUCE_LAUNCH(4, true)
        else LN_REDUCE_LAUNCH(1, true)
    } else {
        if (vec == 4) LN_REDUCE_LAUNCH(4, false)
        else LN_REDUCE_LAUNCH(1, false)
    }
#undef LN_REDUCE_LAUNCH
    return ln_check_launch(who);
}

extern "C" int ln_csr_reduce_rows(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, const float* w,
                                  int val_dim, int src_div, int src_stride, float* dst, void* stream) {
    return ln_csr_reduce_rows_impl("ln_csr_reduce_rows", csr, grp_row, max_segments, src, false, w, val_dim, src_div, src_stride, dst, stream);
}

extern "C" int ln_csr_reduce_rows_f16(const LnCsr* csr, const int* grp_row, long long max_segments, const void* src_f16, const float* w,
                                      int val_dim, int src_div, int src_stride, float* dst, void* stream) {
    return ln_csr_reduce_rows_impl("ln_csr_reduce_rows_f16", csr, grp_row, max_segments, src_f16, true, w, val_dim, src_div, src_stride, dst,
                                   stream);
}

// ln_csr_reduce_rows + ln_neighbours(table, rows_upper, table, same level, dilation 1, no flip) in ONE launch
static int ln_splat_tail_impl(const char* who, const LnCsr* csr, const int* grp_row, long long max_segments, const void* src, bool half,
                              const float* w, int val_dim, int src_div, int src_stride, float* dst, const LnTable* table,
                              int query_rows_upper, int* nbr, void* stream) {
    LnReduceArgs a;
    int vec;
    long long work;
    int rc = ln_reduce_args(who, csr, grp_row, max_segments, src, half, w, val_dim, src_div, src_stride, dst, a, vec, work);
    if (rc) return rc;
    LN_REQUIRE(table && table->slot_keys && table->entries && table->keys && table->nr_filled && table->capacity > 0, LN_ERR_ARG,
               "%s: bad table", who);
    LN_REQUIRE(max_segments > 0 && query_rows_upper > 0 && nbr, LN_ERR_ARG, "%s: nothing to do / null output", who);
    const int d = table->pos_dim;
    LN_REQUIRE(d >= 1 && d <= LN_MAX_POS_DIM, LN_ERR_UNSUPPORTED, "%s: pos_dim %d unsupported", who, d);
    hipStream_t st = (hipStream_t)stream;
    const int reduce_blocks = ln_seg_grid(max_segments, a.lanes_per_seg);  // a multiple of LN_XCD_GROUPS
    const int nbr_blocks = ln_div_up(query_rows_upper, 256 / (2 * (d + 1) + 1));  // whole vertices per workgroup (k_reduce_and_neighbours)
    const dim3 grid(reduce_blocks + nbr_blocks), block(256);
#define LN_FUSED_LAUNCH(VV, DD, HH)                                                                                                  \
    {                                                                                                                                \
        if (wg && VV == 8 && HH && a.lanes_per_seg == 8 && a.chunks == 8 && ln_reduce_l8())                                          \
            LN_LAUNCH("k_reduce_and_neighbours", (k_reduce_and_neighbours<8, DD, true, true, 8>), grid, block, 0, st, a, reduce_blocks, \
                      *table, query_rows_upper, nbr);                                                                                \
        else if (wg)                                                                                                                 \
            LN_LAUNCH("k_reduce_and_neighbours", (k_reduce_and_neighbours<VV, DD, HH, (VV >= 4)>), grid, block, 0, st, a, reduce_blocks, \
                      *table, query_rows_upper, nbr);                                                                                \
        else if (VV == 4 && !HH && a.lanes_per_seg == 8 && a.chunks == 8 && ln_reduce_l8())                                          \
            LN_LAUNCH("k_reduce_and_neighbours", (k_reduce_and_neighbours<4, DD, false, false, 8>), grid, block, 0, st, a, reduce_blocks, \
                      *table, query_rows_upper, nbr);                                                                                \
        else                                                                                                                         \
            LN_LAUNCH("k_reduce_and_neighbours", (k_reduce_and_neighbours<VV, DD, HH, false>), grid, block, 0, st, a, reduce_blocks,  \
                      *table, query_rows_upper, nbr);                                                                                \
    }
#define LN_FUSED_CASE(DD)                                                                                                            \
    case DD:                                                                                                                         \
        if (half && vec == 8) LN_FUSED_LAUNCH(8, DD, true)                                                                           \
        else if (half && vec == 4) LN_FUSED_LAUNCH(4, DD, true)                                                                      \
        else if (half) LN_FUSED_LAUNCH(1, DD, true)                                                                                  \
        else if (vec == 4) LN_FUSED_LAUNCH(4, DD, false)                                                                             \
        else LN_FUSED_LAUNCH(1, DD, false)                                                                                           \
        break;
    const bool wg = ((csr->dense & 1) != 0 || a.lanes_per_seg >= 16) && vec >= 4 && !(csr->dense & 2);
    switch (d) { LN_FUSED_CASE(1) LN_FUSED_CASE(2) LN_FUSED_CASE(3) LN_FUSED_CASE(4) LN_FUSED_CASE(5) LN_FUSED_CASE(6) }
#undef LN_FUSED_LAUNCH
#undef LN_FUSED_CASE
    return ln_check_launch(who);
}

extern "C" int ln_splat_accumulate_and_neighbours(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src,
                                                  const float* w, int val_dim, int src_div, int src_stride, float* dst, const LnTable* table,
                                                  int query_rows_upper, int* nbr, void* stream) {
    return ln_splat_tail_impl("ln_splat_accumulate_and_neighbours", csr, grp_row, max_segments, src, false, w, val_dim, src_div, src_stride, dst,
                              table, query_rows_upper, nbr, stream);
}

extern "C" int ln_splat_accumulate_and_neighbours_f16(const LnCsr* csr, const int* grp_row, long long max_segments, const void* src_f16,
                                                      const float* w, int val_dim, int src_div, int src_stride, float* dst,
                                                      const LnTable* table, int query_rows_upper, int* nbr, void* stream) {
    return ln_splat_tail_impl("ln_splat_accumulate_and_neighbours_f16", csr, grp_row, max_segments, src_f16, true, w, val_dim, src_div, src_stride,
                              dst, table, query_rows_upper, nbr, stream);
}
// ------------------------------------------------------------------------------------------
// segment max with argmax (the PointNet aggregation of the reference's first stage:
// torch_scatter.scatter_max over splat indices, lattice_modules.py:688) and vertex degrees
// (torch_scatter.scatter_add of ones, lattice_modules.py:692) on the same CSR adjacency.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int ln_float_to_ordered(float f) {
    f = (f == 0.0f) ? 0.0f : f;  // -0 and +0 compare equal: give them one code so ties go to the smallest token
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ln_ordered_to_float(unsigned int o) {
    const unsigned int u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

// one lane per (segment, channel): running max over <=16 tokens; value and token are packed into 64 bits
// (ordered value in the high word, ~token in the low word -> ties go to the smallest token) so that the
// segments of a hot vertex combine with one 64-bit atomicMax.
// COMB (lanes per segment a power of two <= 64, so that no segment straddles a wave): the segments of one row that sit side by side in a
// wave, and across the waves of the workgroup, are combined before anything is written — as in ln_reduce_chunk.  The 64-bit atomics were
// half of the kernel on the C3 cloud (43 -> 20 us with plain stores: 69 % of its tokens sit on vertices with several segments).
template <int VEC, bool COMB>
__global__ void __launch_bounds__(256)
    k_csr_segment_max(const int* __restrict__ csr_tok, const int4* __restrict__ seg_desc, const int* __restrict__ seg_count, long long seg_region,
                      const int* __restrict__ grp_row, const float* __restrict__ src, int channels,
                      unsigned long long* __restrict__ packed, int* __restrict__ counts) {
  // a lane owns VEC consecutive channels of its segment's rows (VEC = 4: one 16-byte load per token and lane, four tokens in flight)
  const int lanes = channels / VEC;
  for (LnSegWalk wk(blockIdx.x, gridDim.x, lanes, seg_count, seg_region); wk.more(); wk.next()) {
    const LnSegOfThread so = wk.here();
    const long long sid = so.sid;
    const int lc = so.lane_in_seg;
    const int c = lc * VEC;
    int row = -1, beg = 0, end = 0, rbeg = 0, rend = 0;
    if (so.active) {
        const int4 sd = seg_desc[sid];
        const int grp = sd.x;
        row = ((seg_count[LN_XCD_GROUPS + 1] & 1) || !grp_row) ? grp : grp_row[grp];
        beg = sd.y;
        end = beg + min(LN_SEG, sd.z);
        rbeg = beg - sd.w;
        rend = beg + sd.z;
    }
    if constexpr (!COMB) {
        if (row < 0) continue;
    }
    unsigned long long best[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) best[k] = 0ull;
    constexpr int U = 4;
    if (row >= 0)
    for (int e0 = beg; e0 < end; e0 += U) {
        int tk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) tk[u] = (e0 + u < end) ? csr_tok[e0 + u] : -1;
        float x[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (tk[u] >= 0) {
                if constexpr (VEC == 4) {
                    const float4 v4 = *reinterpret_cast<const float4*>(src + (size_t)tk[u] * channels + c);
                    x[u][0] = v4.x; x[u][1] = v4.y; x[u][2] = v4.z; x[u][3] = v4.w;
                } else {
                    x[u][0] = src[(size_t)tk[u] * channels + c];
                }
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (tk[u] >= 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const unsigned long long p = ((unsigned long long)ln_float_to_ordered(x[u][k]) << 32) |
                                                 (unsigned long long)(0xFFFFFFFFu - (unsigned int)tk[u]);
                    best[k] = p > best[k] ? p : best[k];
                }
            }
    }
    int run_end = end;
    bool writes = row >= 0;  // this lane group writes its run
    if constexpr (COMB) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int groups_per_wave = 64 / lanes, g = lane / lanes;
        // segmented suffix maximum over the lane groups of the wave (the segments of a row are neighbours)
        for (int off = 1; off < groups_per_wave; off <<= 1) {
            const int delta = off * lanes;
            const int o_row = __shfl_down(row, delta, 64);
            const int o_end = __shfl_down(run_end, delta, 64);
            unsigned long long o_best[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o_best[k] = __shfl_down(best[k], delta, 64);
            if (g + off < groups_per_wave && o_row == row && row >= 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) best[k] = o_best[k] > best[k] ? o_best[k] : best[k];
                run_end = max(run_end, o_end);
            }
        }
        const int prev_row = __shfl_up(row, lanes, 64);
        const bool head = (g == 0) || (prev_row != row);
        // ... and over the wave boundaries of the workgroup
        __shared__ unsigned long long s_first_best[4][64 * VEC];
        __shared__ int s_first_row[4], s_first_end[4], s_first_whole[4], s_last_row[4];
        const int first_row = __shfl(row, 0, 64);
        const int last_row = __shfl(row, (groups_per_wave - 1) * lanes, 64);
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) s_first_best[wave][lc * VEC + k] = best[k];
            if (lc == 0) {
                s_first_row[wave] = row;
                s_first_end[wave] = run_end;
                s_first_whole[wave] = (first_row >= 0 && first_row == last_row) ? 1 : 0;
                s_last_row[wave] = last_row;
            }
        }
        __syncthreads();
        const bool absorbed = head && g == 0 && wave > 0 && row >= 0 && s_last_row[wave - 1] == row;
        if (head && row >= 0 && row == last_row && !absorbed) {
            for (int w2 = wave + 1; w2 < 4 && s_first_row[w2] == row; ++w2) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const unsigned long long o = s_first_best[w2][lc * VEC + k];
                    best[k] = o > best[k] ? o : best[k];
                }
                run_end = max(run_end, s_first_end[w2]);
                if (!s_first_whole[w2]) break;
            }
        }
        writes = row >= 0 && head && !absorbed;
    }
    if (!writes) continue;
    const bool whole_row = beg == rbeg && run_end == rend;  // no other writer
    unsigned long long* d = packed + (size_t)row * channels + c;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
#ifdef LN_SEGMAX_PLAIN  // timing ablation (wrong results): what the 64-bit atomics cost
        d[k] = best[k];
#else
        if (whole_row)
            d[k] = best[k];
        else
            atomicMax(d + k, best[k]);
#endif
    }
    if (counts && c == 0) {  // vertex degree on the side (the fused PointNet reduction needs it; integer adds: order-free)
        if (whole_row)
            counts[row] = run_end - beg;
        else
            atomicAdd(counts + row, run_end - beg);
    }
  }
}

__global__ void __launch_bounds__(256)
    k_csr_segment_max_decode(const unsigned long long* __restrict__ packed, long long work, float* __restrict__ out_max,
                             int* __restrict__ out_arg) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const unsigned long long p = packed[g];
    if (p == 0ull) {  // vertex without any token
        out_max[g] = 0.f;
        out_arg[g] = -1;
    } else {
        out_max[g] = ln_ordered_to_float((unsigned int)(p >> 32));
        out_arg[g] = int(0xFFFFFFFFu - (unsigned int)(p & 0xFFFFFFFFull));
    }
}

static void ln_launch_segment_max(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, int channels,
                                  unsigned long long* packed, int* counts, hipStream_t st) {
    const bool vec4 = channels % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    const int lanes = vec4 ? channels / 4 : channels;
    const bool comb = lanes <= 64 && (lanes & (lanes - 1)) == 0;  // no segment straddles a wave: runs can be combined before they are written
#define LN_SEGMAX(VV, CC)                                                                                                            \
    LN_LAUNCH("k_csr_segment_max", (k_csr_segment_max<VV, CC>), dim3(ln_seg_grid(max_segments, lanes)), dim3(256), 0, st, csr->csr_tok, \
              reinterpret_cast<const int4*>(csr->seg_desc), csr->seg_count, csr->seg_region, grp_row, src, channels, packed, counts)
    if (vec4 && comb) LN_SEGMAX(4, true);
    else if (vec4) LN_SEGMAX(4, false);
    else if (comb) LN_SEGMAX(1, true);
    else LN_SEGMAX(1, false);
#undef LN_SEGMAX
}

extern "C" int ln_csr_segment_max(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, int channels,
                                  int rows, void* packed_ws, float* out_max, int* out_arg, void* stream) {
    LN_REQUIRE(max_segments >= 0 && channels >= 1 && rows >= 0, LN_ERR_ARG, "ln_csr_segment_max: bad sizes");
    LN_REQUIRE(rows == 0 || (csr && csr->grp_start && csr->csr_tok && csr->seg_desc && csr->seg_count && src && packed_ws &&
                             out_max && out_arg),
               LN_ERR_ARG, "ln_csr_segment_max: null buffer");
    if (rows == 0) return LN_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long work = (long long)rows * channels;
    if (ln_zero_async(packed_ws, (size_t)work * sizeof(unsigned long long), st) != LN_OK)
        return ln_check_launch("ln_csr_segment_max(memset)");
    if (max_segments > 0)
        ln_launch_segment_max(csr, grp_row, max_segments, src, channels, static_cast<unsigned long long*>(packed_ws), nullptr, st);
    LN_LAUNCH("k_csr_segment_max_decode", k_csr_segment_max_decode, dim3(ln_div_up(work, 256)), dim3(256), 0, st,
              static_cast<const unsigned long long*>(packed_ws), work, out_max, out_arg);
    return ln_check_launch("ln_csr_segment_max");
}

// ---- the whole vertex-side reduction of PointNetModule (lattice_modules.py:688-712) in three launches ------------------------
//   out[row, :C]  = max over the row's tokens of src[t, :C]            out[row, C:] = bary[argmax token]
//   rows with fewer than `min_points` tokens and row 0 (the "invalid" bucket) are zero; arg = -1 there and where no token is
// (the reference: scatter_max, scatter_add of ones, index_select, cat, masked_fill, a multiplication by a keep mask).
__global__ void __launch_bounds__(256)
    k_pointnet_reduce_decode(const unsigned long long* __restrict__ packed, const int* __restrict__ counts, const float* __restrict__ bary,
                             int bary_stride, long long work, int channels, int min_points, float* __restrict__ out, int* __restrict__ out_arg) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= work) return;
    const long long row = g / channels;
    const int c = int(g - row * channels);
    const unsigned long long p = packed[g];
    float mx = 0.f, bw = 0.f;
    int arg = -1;
    if (p != 0ull && row != 0 && counts[row] >= min_points) {
        mx = ln_ordered_to_float((unsigned int)(p >> 32));
        arg = int(0xFFFFFFFFu - (unsigned int)(p & 0xFFFFFFFFull));
        bw = bary[(size_t)arg * bary_stride];
    }
    out[row * 2 * channels + c] = mx;
    out[row * 2 * channels + channels + c] = bw;
    out_arg[g] = arg;
}

// gradient of the maxima wrt the per-token rows, token-major (every element written: no zero fill, no scatter):
//   grad_src[t, c] = grad_out[row, c] if arg[row, c] == t else 0,   row = idx[t]
// A thread owns VEC consecutive channels of one token (VEC = 4: 16-byte loads of the winners' ids and a 16-byte store).
template <int VEC>
__global__ void __launch_bounds__(256)
    k_pointnet_reduce_backward(const float* __restrict__ grad_out, int grad_stride, const int* __restrict__ arg, const int* __restrict__ idx,
                               long long work, int channels, float* __restrict__ grad_src) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over tokens * channels / VEC
    if (g >= work) return;
    const int per_row = channels / VEC;
    const long long t = g / per_row;
    const int c = int(g - t * per_row) * VEC;
    const int row = idx[t];
    float x[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) x[k] = 0.f;
    if (row > 0) {
        if constexpr (VEC == 4) {
            const int4 a = *reinterpret_cast<const int4*>(arg + (size_t)row * channels + c);
            const int tt = (int)t;
            if (a.x == tt || a.y == tt || a.z == tt || a.w == tt) {
                const float* gr = grad_out + (size_t)row * grad_stride + c;
                if (a.x == tt) x[0] = gr[0];
                if (a.y == tt) x[1] = gr[1];
                if (a.z == tt) x[2] = gr[2];
                if (a.w == tt) x[3] = gr[3];
            }
        } else {
            if (arg[(size_t)row * channels + c] == (int)t) x[0] = grad_out[(size_t)row * grad_stride + c];
        }
    }
    if constexpr (VEC == 4)
        *reinterpret_cast<float4*>(grad_src + g * 4) = make_float4(x[0], x[1], x[2], x[3]);
    else
        grad_src[g] = x[0];
}

extern "C" size_t ln_pointnet_reduce_workspace_bytes(int rows, int channels) {
    if (rows < 0 || channels < 1) return 256;
    return (size_t)rows * channels * sizeof(unsigned long long) + (size_t)rows * sizeof(int) + 256;
}

extern "C" int ln_pointnet_reduce_forward(const LnCsr* csr, const int* grp_row, long long max_segments, const float* src, int channels,
                                          const float* bary, int bary_stride, int rows, int min_points, void* workspace,
                                          size_t workspace_bytes, float* out, int* out_arg, void* stream) {
    LN_REQUIRE(max_segments >= 0 && channels >= 1 && rows >= 0 && bary_stride >= 1, LN_ERR_ARG, "ln_pointnet_reduce_forward: bad sizes");
    if (rows == 0) return LN_OK;
    LN_REQUIRE(csr && csr->grp_start && csr->csr_tok && csr->seg_desc && csr->seg_count && src && bary && workspace && out && out_arg,
               LN_ERR_ARG, "ln_pointnet_reduce_forward: null buffer");
    LN_REQUIRE(workspace_bytes >= ln_pointnet_reduce_workspace_bytes(rows, channels), LN_ERR_WORKSPACE,
               "ln_pointnet_reduce_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long long work = (long long)rows * channels;
    unsigned long long* packed = static_cast<unsigned long long*>(workspace);
    int* counts = reinterpret_cast<int*>(packed + work);
    if (ln_zero_async(workspace, (size_t)work * sizeof(unsigned long long) + (size_t)rows * sizeof(int), st) != LN_OK)
        return ln_check_launch("ln_pointnet_reduce_forward(memset)");
    if (max_segments > 0)
        ln_launch_segment_max(csr, grp_row, max_segments, src, channels, packed, counts, st);
    LN_LAUNCH("k_pointnet_reduce_decode", k_pointnet_reduce_decode, dim3(ln_div_up(work, 256)), dim3(256), 0, st, packed, counts, bary, bary_stride,
              work, channels, min_points, out, out_arg);
    return ln_check_launch("ln_pointnet_reduce_forward");
}

extern "C" int ln_pointnet_reduce_backward(const float* grad_out, int grad_stride, const int* arg, const int* splat_idx, long long tokens,
                                           int channels, float* grad_src, void* stream) {
    LN_REQUIRE(tokens >= 0 && channels >= 1 && grad_stride >= channels, LN_ERR_ARG, "ln_pointnet_reduce_backward: bad sizes");
    if (tokens == 0) return LN_OK;
    LN_REQUIRE(grad_out && arg && splat_idx && grad_src, LN_ERR_ARG, "ln_pointnet_reduce_backward: null buffer");
    const bool vec4 = channels % 4 == 0 && ((reinterpret_cast<uintptr_t>(arg) | reinterpret_cast<uintptr_t>(grad_src)) & 15) == 0;
    const long long work = tokens * channels / (vec4 ? 4 : 1);
    if (vec4)
        LN_LAUNCH("k_pointnet_reduce_backward", k_pointnet_reduce_backward<4>, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream,
                  grad_out, grad_stride, arg, splat_idx, work, channels, grad_src);
    else
        LN_LAUNCH("k_pointnet_reduce_backward", k_pointnet_reduce_backward<1>, dim3(ln_div_up(work, 256)), dim3(256), 0, (hipStream_t)stream,
                  grad_out, grad_stride, arg, splat_idx, work, channels, grad_src);
    return ln_check_launch("ln_pointnet_reduce_backward");
}

// degree of every row: number of tokens whose group maps to it
__global__ void __launch_bounds__(256)
    k_csr_group_sizes(const int* __restrict__ grp_start, const int* __restrict__ grp_row, int groups_upper, int* __restrict__ counts) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups_upper) return;
    const int c = grp_start[g + 1] - grp_start[g];
    if (c == 0) return;
    const int row = grp_row ? grp_row[g] : g;
    if (row >= 0) counts[row] = c;  // one group per row: no conflicts
}

extern "C" int ln_csr_group_sizes(const LnCsr* csr, const int* grp_row, int groups_upper, int rows, int* counts, void* stream) {
    LN_REQUIRE(groups_upper >= 1 && rows >= 0, LN_ERR_ARG, "ln_csr_group_sizes: bad sizes");
    LN_REQUIRE(rows == 0 || (csr && csr->grp_start && counts), LN_ERR_ARG, "ln_csr_group_sizes: null buffer");
    if (rows == 0) return LN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (ln_zero_async(counts, (size_t)rows * sizeof(int), st) != LN_OK) return ln_check_launch("ln_csr_group_sizes(memset)");
    LN_LAUNCH("k_csr_group_sizes", k_csr_group_sizes, dim3(ln_div_up(groups_upper, 256)), dim3(256), 0, st, csr->grp_start, grp_row, groups_upper,
              counts);
    return ln_check_launch("ln_csr_group_sizes");
}
