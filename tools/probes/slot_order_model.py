#!/usr/bin/env python3
"""CPU model (NumPy, oracle keys) of candidate slot functions for the bucketed build: bucket load balance and how many distinct
128-byte value rows each XCD would fetch for the 9-neighbour gather of the convolution and the d+1 row gather of the slice.
Round 6, verdict item 1(a): decides which locality-preserving slot function is worth building.  Not a product path."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import lattice_oracle as O
from lattice_net_amd import synthetic

def vertices(pos, sigma):
    p = O.scale_positions(pos, np.full((pos.shape[1],), sigma, np.float32))
    rem0, rank, bary = O.simplex(p)
    keys = O.simplex_keys(rem0, rank)  # [N, d+1, d]
    n, dp1, d = keys.shape
    flat = keys.reshape(-1, d).astype(np.int64)
    uk, inv, cnt = np.unique(flat, axis=0, return_inverse=True, return_counts=True)
    return uk, inv.reshape(n, dp1), cnt

def kd_planes(keys, wts, levels):
    """balanced kd tree over key space, axis cycling; returns leaf id per vertex (levels bits) — calibrated on THIS set."""
    d = keys.shape[1]
    leaf = np.zeros(len(keys), np.int64)
    tree = {}
    def rec(sel, lvl, node):
        if lvl == levels: return
        ax = lvl % d
        k = keys[sel, ax]; w = wts[sel]
        o = np.argsort(k, kind="stable"); cs = np.cumsum(w[o])
        if len(o) == 0:
            plane = 0
        else:
            plane = k[o][np.searchsorted(cs, cs[-1] / 2)]
        tree[node] = (ax, plane)
        hi = keys[sel, ax] >= plane
        leaf[sel[hi]] |= 1 << (levels - 1 - lvl)
        rec(sel[~hi], lvl + 1, 2 * node + 1); rec(sel[hi], lvl + 1, 2 * node + 2)
    rec(np.arange(len(keys)), 0, 0)
    return leaf, tree

def apply_tree(keys, tree, levels):
    d = keys.shape[1]
    leaf = np.zeros(len(keys), np.int64); node = np.zeros(len(keys), np.int64)
    for lvl in range(levels):
        ax = np.array([tree[n_][0] for n_ in node]); pl = np.array([tree[n_][1] for n_ in node])
        hi = keys[np.arange(len(keys)), ax] >= pl
        leaf |= hi.astype(np.int64) << (levels - 1 - lvl)
        node = 2 * node + 1 + hi
    return leaf

def hash_u32(keys):
    k = np.zeros(len(keys), np.uint64)
    for i in range(keys.shape[1]):
        k = (k + (keys[:, i].astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
        k = (k * np.uint64(2531011)) & np.uint64(0xFFFFFFFF)
    return k

def stir(k):
    k = k ^ (k >> np.uint64(15)); k = (k * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF); k ^= k >> np.uint64(13); return k

def neighbours(uk):
    d = uk.shape[1]
    full = np.concatenate([uk, -uk.sum(1, keepdims=True)], 1)
    look = {tuple(k): i for i, k in enumerate(uk)}
    nbr = np.full((len(uk), 2 * (d + 1) + 1), -1, np.int64)
    for a in range(d + 1):
        for s, col in ((1, 2 * a), (-1, 2 * a + 1)):
            nk = full + s; nk[:, a] = full[:, a] - s * d
            nbr[:, col] = [look.get(tuple(k[:d]), -1) for k in nk]
    nbr[:, -1] = np.arange(len(uk))
    return nbr

def report(name, bucket, uk, cnt, nbr, inv, nbk, sb):
    m = len(uk)
    vload = np.bincount(bucket, minlength=nbk); tload = np.bincount(bucket, weights=cnt, minlength=nbk)
    # rows: bucket-major (order inside a bucket irrelevant at 128-byte granularity for V=32: one row = one line)
    order = np.argsort(bucket, kind="stable"); row = np.empty(m, np.int64); row[order] = np.arange(m)
    # conv: workgroup of 64 rows -> XCD = wg % 8 (today's mapping) vs region mapping (XCD = bucket * 8 // nbk)
    nb_rows = np.where(nbr >= 0, row[np.maximum(nbr, 0)], -1)
    res = {}
    for mapping in ("wg%8", "region"):
        xcd_of_row = (np.arange(m) // 64) % 8 if mapping == "wg%8" else (bucket[order] * 8 // nbk)
        tot = 0
        for x in range(8):
            q = order[xcd_of_row == x]  # vertices whose OUTPUT row this XCD computes
            lines = np.unique(nb_rows[q][nb_rows[q] >= 0])
            tot += len(lines)
        res[mapping] = tot / m
    # slice: points in input order (wg of 32 points -> XCD wg%8) vs points by region of first vertex
    prow = row[inv]  # [N, d+1]
    n = len(prow)
    tot_in = 0; tot_reg = 0
    xin = (np.arange(n) // 32) % 8; xreg = bucket[inv[:, 0]] * 8 // nbk
    for x in range(8):
        tot_in += len(np.unique(prow[xin == x])); tot_reg += len(np.unique(prow[xreg == x]))
    print(f"{name:28s} vertices/bucket max {vload.max():4d} mean {vload.mean():6.1f} (slots {sb})  tokens/bucket max {int(tload.max()):6d} mean {tload.mean():7.1f}"
          f" | conv rows fetched / M: wg%8 {res['wg%8']:.2f} region {res['region']:.2f} | slice rows / M: input order {tot_in / m:.2f} region {tot_reg / m:.2f}")

def main():
    n, sigma = 120000, 0.9
    nbk, cap = 256, 100000
    sb = (cap + nbk - 1) // nbk
    cal = synthetic.lidar_cloud(n, 7777)
    ukc, invc, cntc = vertices(cal, sigma)
    for seed in (0, 1):
        pos = synthetic.lidar_cloud(n, seed)
        uk, inv, cnt = vertices(pos, sigma)
        nbr = neighbours(uk)
        print(f"seed {seed}: M = {len(uk)}, tokens/vertex mean {cnt.mean():.1f} max {cnt.max()}")
        h = hash_u32(uk)
        report("hash (today)", (h % np.uint64(cap)).astype(np.int64) // sb, uk, cnt, nbr, inv, nbk, sb)
        for levels, what in ((3, "kd8 x hash32"), (5, "kd32 x hash8"), (8, "kd256")):
            _, tree = kd_planes(ukc, cntc.astype(float), levels)   # calibrated on ANOTHER cloud, token weighted
            leaf = apply_tree(uk, tree, levels)
            sub = nbk >> levels
            b = leaf * sub + (stir(h) % np.uint64(max(sub, 1))).astype(np.int64)
            report(what + " (tok-weighted)", b, uk, cnt, nbr, inv, nbk, sb)
            _, tree = kd_planes(ukc, np.ones(len(ukc)), levels)    # vertex weighted
            leaf = apply_tree(uk, tree, levels)
            b = leaf * sub + (stir(h) % np.uint64(max(sub, 1))).astype(np.int64)
            report(what + " (vtx-weighted)", b, uk, cnt, nbr, inv, nbk, sb)
        for shift in (2, 3):
            cell = uk >> shift
            ch = stir(hash_u32(cell))
            report(f"cellhash>>{shift}", (ch % np.uint64(nbk)).astype(np.int64), uk, cnt, nbr, inv, nbk, sb)
            _, tree = kd_planes(ukc, np.ones(len(ukc)), 3)
            leaf = apply_tree(uk, tree, 3)
            report(f"kd8 x cellhash>>{shift}", leaf * 32 + (ch % np.uint64(32)).astype(np.int64), uk, cnt, nbr, inv, nbk, sb)

if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def main2():
    """mixed weights (tokens + mean tokens per vertex), levels 3..6, and the contiguous-eighth workgroup map of the convolutions"""
    n, sigma, nbk, cap = 120000, 0.9, 256, 100000
    sb = (cap + nbk - 1) // nbk
    cal = synthetic.lidar_cloud(n, 7777)
    ukc, invc, cntc = vertices(cal, sigma)
    pos = synthetic.lidar_cloud(n, 0)
    uk, inv, cnt = vertices(pos, sigma)
    nbr = neighbours(uk)
    h = hash_u32(uk)
    m = len(uk)
    for levels in (3, 4, 5, 6):
        for wname, wts in (("tokens", cntc.astype(float)), ("mixed", cntc + cntc.mean()), ("vertices", np.ones(len(ukc)))):
            _, tree = kd_planes(ukc, wts, levels)
            leaf = apply_tree(uk, tree, levels)
            sub = nbk >> levels
            b = leaf * sub + (stir(h) % np.uint64(sub)).astype(np.int64)
            vload = np.bincount(b, minlength=nbk); tload = np.bincount(b, weights=cnt, minlength=nbk)
            order = np.argsort(b, kind="stable"); row = np.empty(m, np.int64); row[order] = np.arange(m)
            nb_rows = np.where(nbr >= 0, row[np.maximum(nbr, 0)], -1)
            xc = np.arange(m) * 8 // m  # XCD of output row (contiguous eighths)
            tot = 0
            for x in range(8):
                q = order[xc == x]
                tot += len(np.unique(nb_rows[q][nb_rows[q] >= 0]))
            prow = row[inv]
            # slice in CSR order: point handled where its r = 0 token sits; XCD = contiguous eighth of the CSR (tokens in bucket order)
            tok_bucket = b[inv[:, 0]]
            pord = np.argsort(tok_bucket, kind="stable"); xs = np.empty(len(pord), np.int64); xs[pord] = np.arange(len(pord)) * 8 // len(pord)
            ts = sum(len(np.unique(prow[xs == x])) for x in range(8))
            print(f"levels {levels} weights {wname:8s}: vertices/bucket max {vload.max():4d} (slots {sb}) tokens/bucket max {int(tload.max()):6d} (mean {tload.mean():.0f})"
                  f" | conv rows fetched / M (contiguous eighths) {tot / m:.2f} | slice rows / M (CSR order) {ts / m:.2f}")

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mixed":
    main2()
