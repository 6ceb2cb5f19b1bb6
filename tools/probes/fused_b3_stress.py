#!/usr/bin/env python3
"""Stress of the fused convolution backward under concurrency: K captured C3-like steps replayed concurrently, the filter gradient of
every replay compared with the eager one (1e-5).  argv: n (points), K (scans), rounds."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import lattice_net_amd as L
from lattice_net_amd.synthetic import lidar_cloud

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 100
v = f = 32
cap = 100000 if n > 20000 else 30000
rng = np.random.default_rng(9)
W = torch.from_numpy((rng.standard_normal((9 * v, f)) / np.sqrt(9 * v)).astype(np.float32)).to(dev).requires_grad_(True)
scans = []
for k in range(K):
    sc = {"st": {}, "pos": torch.from_numpy(lidar_cloud(n, 40 + k)).to(dev), "vals": torch.randn((n, v), device=dev), "G": torch.randn((n, f), device=dev),
          "lat": L.Lattice(sigmas=[0.9] * 3, capacity=cap, device=dev)}

    def step(sc=sc):
        W.grad = None
        lv, _, idx, w = L.SplatLattice.apply(sc["lat"], sc["pos"], sc["vals"])
        lv = lv[:sc["lat"].nr_lattice_vertices()].requires_grad_(True)
        cv, cw = L.ConvIm2RowLattice.apply(lv, sc["lat"], W, 1)
        out = L.SliceLattice.apply(cv, cw.lattice, sc["pos"], idx, w)
        out.backward(sc["G"])
        sc["st"].update(idx=idx, gw=W.grad, gv=lv.grad)

    step()
    torch.cuda.synchronize()
    sc["gw_ref"] = sc["st"]["gw"].detach().clone()
    sc["cap"] = L.CapturedStep(step, [sc["lat"]], region_indices=lambda sc=sc: sc["st"]["idx"], stream=torch.cuda.Stream(), before_capture=sc["st"].clear)
    scans.append(sc)
torch.cuda.synchronize()
bad = 0
for rnd in range(rounds):
    for sc in scans:
        sc["cap"].launch()
    torch.cuda.synchronize()
    for k, sc in enumerate(scans):
        rel = float((sc["st"]["gw"] - sc["gw_ref"]).abs().max() / sc["gw_ref"].abs().max())
        if rel > 1e-5:
            bad += 1
            print("round", rnd, "scan", k, "rel", rel, flush=True)
print(f"n={n} K={K}: {bad} bad of {rounds * K} replays")
