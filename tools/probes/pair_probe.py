#!/usr/bin/env python3
"""Which kernel is the victim?  One stream loops the fused convolution backward of scan A (the form chosen by LN_BWD_T /
LN_DEBUG_MASK in the environment), a second stream loops a kernel of scan B on fixed inputs and compares every result with its
first one: the segment reduce of the slice backward (k_csr_reduce_segments), the convolution forward, a torch elementwise kernel.
argv: rounds.  (wrong gradient rows when the one- / two-sub-tile bf16x3 backward runs beside other kernels.)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import lattice_net_amd as L
from lattice_net_amd.synthetic import lidar_cloud

dev = torch.device("cuda", 0)
n, v = 120000, 32
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(3)
W = torch.from_numpy((rng.standard_normal((9 * v, v)) / 17).astype(np.float32)).to(dev)


def scan(seed):
    pos = torch.from_numpy(lidar_cloud(n, seed)).to(dev)
    lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
    vals = torch.randn((n, v), device=dev)
    lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    lat.neighbours(lat, 1, False)
    return dict(lat=lat, pos=pos, idx=idx, w=w, m=m, G=torch.randn((m, v), device=dev), P=torch.randn((n, v), device=dev))


A, B = scan(1), scan(2)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


import os
KIND = os.environ.get("PAIR_AGGRESSOR", "fused")  # fused: the fused backward; perslot: the per-slot bf16x3 convolution (V = F = 64)
if KIND == "perslot":
    W64 = torch.from_numpy((rng.standard_normal((9 * 64, 64)) / 24).astype(np.float32)).to(dev)
    A["lat"].set_values(torch.randn((A["m"], 64), device=dev))


def aggressor():
    if KIND == "perslot":
        A["lat"].convolve_im2row_standalone(W64, 1, A["lat"], False)
    else:
        A["lat"].convolve_im2row_backward(A["G"], W, 1, None, None)


def victim_reduce():
    gv = torch.zeros((B["m"], v), device=dev)
    B["lat"]._scatter_rows(B["P"], B["idx"], B["w"], gv, v, 4, v)
    return gv


def victim_conv():
    return B["lat"].convolve_im2row_standalone(W, 1, B["lat"], False).values()[:B["m"]]


x_el = torch.randn((B["m"], v), device=dev)


def victim_elementwise():
    return x_el * 1.0001 + 0.5


only = sys.argv[2] if len(sys.argv) > 2 else ""
for name, victim in (("segment reduce", victim_reduce), ("convolution forward", victim_conv), ("torch elementwise", victim_elementwise)):
    if only and only not in name:
        continue
    for with_aggressor in (False, True):
        ref = victim().clone()
        scale = float(ref.abs().max())
        torch.cuda.synchronize()
        bad = torch.zeros((), device=dev, dtype=torch.int64)
        worst = torch.zeros((), device=dev)
        for _ in range(rounds):
            if with_aggressor:
                with torch.cuda.stream(sA):
                    aggressor()
            with torch.cuda.stream(sB):
                d = (victim() - ref).abs().max() / scale
                bad += (d > 1e-4).long()
                worst = torch.maximum(worst, d)
        torch.cuda.synchronize()
        print(f"{name:22s} {'beside the fused backward' if with_aggressor else 'alone':26s}: {int(bad)} bad of {rounds}, worst rel {float(worst):.2e}", flush=True)
