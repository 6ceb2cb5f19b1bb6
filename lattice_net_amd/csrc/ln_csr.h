// Internal interface between the hash build (ln_table.hip) and the CSR construction (ln_csr.hip).
#pragma once
#include "ln_common.h"

#ifndef LN_CSR_SEG
#define LN_CSR_SEG 16  // max CSR entries per segment (unit of work of the segment reduce)
#endif

size_t ln_csr_scan_workspace_bytes(int groups_upper);

// Builds the CSR of `tokens` tokens over `groups_upper` groups from already-known per-token
// (group, position-in-group) and per-group counts: scan -> fill.
int ln_csr_from_counts(const int* tok_grp, const int* tok_pos, long long tokens, const int* grp_cnt, int groups_upper,
                       const LnCsr& csr, void* workspace, size_t workspace_bytes, hipStream_t st);

// Rewrites every group's token list in ascending token order (deterministic sums; see ln_csr.hip).  scratch: one int per CSR token.
int ln_csr_sort_groups(const LnCsr& csr, int groups, int* scratch, hipStream_t st);

#if defined(__HIPCC__)
// Which segments a workgroup works on.  Segment lists come in two layouts (LnCsr.seg_count[LN_XCD_GROUPS] says which):
//   1 region : segments 0 .. seg_count[0]-1; workgroup b takes the blocks b, b + nblocks, ... of them;
//   G regions: region g (XCD group g of the table, LnProbe) starts at g * seg_region and holds seg_count[g] segments;
//              workgroup b — dispatched to XCD b % G — takes the blocks b / G, b / G + nblocks / G, ... of region b % G, so
//              that each XCD walks the tokens of its own group of lattice cells.
// A block = 256 / lanes_per_seg consecutive segments.  The grid (ln_seg_grid, a multiple of G) covers max_segments once;
// an uneven split over the regions just makes the workgroups of the heavier regions loop.
struct LnSegOfThread {
    long long sid;  // index into seg_grp / seg_beg
    int lane_in_seg;
    bool active;
};
struct LnSegWalk {
    int region, j, stride, cnt, lanes;
    long long base;
    __device__ __forceinline__ LnSegWalk(int block_x, int nblocks, int lanes_per_seg, const int* __restrict__ seg_count, long long seg_region) {
        const int nreg = seg_count[LN_XCD_GROUPS];
        region = 0;
        j = block_x;
        stride = nblocks;
        if (nreg > 1) {
            region = block_x % LN_XCD_GROUPS;
            j = block_x / LN_XCD_GROUPS;
            stride = nblocks / LN_XCD_GROUPS;
        }
        cnt = seg_count[region];
        lanes = lanes_per_seg;
        base = (long long)region * seg_region;
    }
    __device__ __forceinline__ bool more() const { return (long long)j * 256 < (long long)cnt * lanes; }  // workgroup-uniform
    __device__ __forceinline__ LnSegOfThread here() const {
        LnSegOfThread r;
        const long long gt = (long long)j * 256 + threadIdx.x;
        const long long s = gt / lanes;
        r.lane_in_seg = int(gt - s * lanes);
        r.active = s < cnt;
        r.sid = base + s;
        return r;
    }
    __device__ __forceinline__ void next() { j += stride; }
};
#endif
static inline int ln_seg_grid(long long max_segments, int lanes_per_seg) {
    return LN_XCD_GROUPS * ln_div_up(max_segments * lanes_per_seg, 256 * LN_XCD_GROUPS);
}
