// Internal interface between the hash build (ln_table.hip) and the CSR construction (ln_csr.hip).
#pragma once
#include "ln_common.h"

#define LN_CSR_SEG 16  // max CSR entries per segment (unit of work of the segment reduce)

size_t ln_csr_scan_workspace_bytes(int groups_upper);

// Builds the CSR of `tokens` tokens over `groups_upper` groups from already-known per-token
// (group, position-in-group) and per-group counts: scan -> fill.
int ln_csr_from_counts(const int* tok_grp, const int* tok_pos, long long tokens, const int* grp_cnt, int groups_upper,
                       const LnCsr& csr, void* workspace, size_t workspace_bytes, hipStream_t st);
