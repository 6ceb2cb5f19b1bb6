#!/usr/bin/env python3
"""CPU probe (NumPy oracle): would kd leaves of key space work as hash BUCKETS (one bucket workgroup owning a spatial cell, so that
the splat accumulate could be fused into the bucket pass with every point row read ~1.4x instead of 4x)?  Planes are medians of
the vertex keys of one cloud, applied to other clouds.  Result (120 k-point LiDAR-like scans, 256 leaves): vertices per leaf
109 ... 265 (fine), leaves per point 1.44 (fine), TOKENS per leaf up to 29 k against a mean of 1.9 k — the sensor's near field
puts hundreds of points on single vertices, so vertex-balanced cells are 15x token-imbalanced and token-balanced cells are
10x vertex-imbalanced.  Hash buckets balance both; the idea was dropped."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lattice_oracle as O
from lattice_net_amd import synthetic
def keys_of(seed, n=120000, sigma=0.9):
    pos = O.scale_positions(synthetic.lidar_cloud(n, seed), np.full((3,), sigma, np.float32))
    rem0, rank, bary = O.simplex(pos)
    k = O.simplex_keys(rem0, rank).reshape(n * 4, 3).astype(np.int64)
    return k
def build_planes(uk, levels):
    planes = {}
    def rec(idx, node, lvl):
        if lvl == levels: return
        d = lvl % 3
        med = np.median(uk[idx, d])
        thr = int(np.floor(med))
        planes[node] = thr
        left = idx[uk[idx, d] <= thr]; right = idx[uk[idx, d] > thr]
        rec(left, 2 * node, lvl + 1); rec(right, 2 * node + 1, lvl + 1)
    rec(np.arange(uk.shape[0]), 1, 0)
    return planes
def leaf_of(k, planes, levels):
    node = np.ones(k.shape[0], np.int64)
    for lvl in range(levels):
        d = lvl % 3
        thr = np.array([planes.get(int(x), 0) for x in np.unique(node)])
        m = dict(zip(np.unique(node).tolist(), thr.tolist()))
        t = np.vectorize(m.get)(node)
        node = 2 * node + (k[:, d] > t)
    return node - (1 << levels)
for levels in (8, 9):
    kA = keys_of(101)
    uA = np.unique(kA, axis=0)
    planes = build_planes(uA, levels)
    for seed in (0, 1, 7):
        kB = keys_of(seed)
        uB, inv = np.unique(kB, axis=0, return_inverse=True)
        lf_rows = leaf_of(uB, planes, levels)
        rows_per = np.bincount(lf_rows, minlength=1 << levels)
        lf_tok = lf_rows[inv.reshape(-1)]
        tok_per = np.bincount(lf_tok, minlength=1 << levels)
        pl = lf_tok.reshape(-1, 4)
        srt = np.sort(pl, axis=1)
        distinct = 1 + (srt[:, 1:] != srt[:, :-1]).sum(1)
        print(f"levels {levels} seed {seed}: rows {uB.shape[0]} per leaf mean {rows_per.mean():.0f} max {rows_per.max()} min {rows_per.min()}; tokens per leaf max {tok_per.max()} mean {tok_per.mean():.0f}; leaves per point mean {distinct.mean():.3f}  hist {np.bincount(distinct)[1:]}")
