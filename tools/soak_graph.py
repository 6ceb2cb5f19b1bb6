#!/usr/bin/env python3
"""Soak of the captured step: ONE capture, then many different clouds written into the captured input buffers; every replay is
checked against its bounds and every k-th one against an eager pass on a second lattice (1e-5).  python tools/soak_graph.py [clouds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
clouds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, v, f, sigma, cap = 120000, 32, 32, 0.9, 100000
rng = np.random.default_rng(0)
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
W = (torch.rand((9 * v, f), device=dev) - 0.5).requires_grad_(True)
lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
ref_lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
st = {}


def run(lattice, store):
    W.grad = None
    lv, _, idx, w = L.SplatLattice.apply(lattice, pos, vals)
    lv = lv[:lattice.nr_lattice_vertices()].requires_grad_(True)
    cv, cw = L.ConvIm2RowLattice.apply(lv, lattice, W, 1)
    out = L.SliceLattice.apply(cv, cw.lattice, pos, idx, w)
    out.backward(G)
    store.update(out=out, gw=W.grad, idx=idx)


cap_step = L.CapturedStep(lambda: run(lat, st), [lat], row_slack=0.10, region_indices=lambda: st["idx"], before_capture=st.clear)
worst = 0.0
over = 0
t0 = time.time()
for k in range(1, clouds + 1):
    pos.copy_(torch.from_numpy(synthetic.lidar_cloud(n, k)))
    cap_step.launch()
    torch.cuda.synchronize()
    try:
        m = cap_step.check()[0]
    except L.LatticeNetHipError as exc:  # a cloud with more vertices than the calibrated bound: reported, never silent
        over += 1
        print(f"cloud {k}: {exc}")
        continue
    if k % 10 == 0:
        ref = {}
        out_g, gw_g = st["out"].detach().clone(), st["gw"].detach().clone()
        run(ref_lat, ref)
        e1 = float((out_g - ref["out"]).abs().max() / ref["out"].abs().max())
        e2 = float((gw_g - ref["gw"]).abs().max() / ref["gw"].abs().max())
        worst = max(worst, e1, e2)
        assert ref_lat.nr_lattice_vertices() == m
print(f"{clouds} clouds through one captured step in {time.time() - t0:.1f} s: {over} beyond the row bound (reported), worst relative deviation from eager {worst:.2e}")
assert worst < 1e-5
