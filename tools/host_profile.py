#!/usr/bin/env python3
"""cProfile of the host side of one bench step (where does the CPU time go?)."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402

dev = torch.device("cuda", 0)
n, v, f = int(os.environ.get("HP_N", "120000")), 32, 32
rng = np.random.default_rng(0)
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
W = (torch.rand((9 * v, f), device=dev) - 0.5).requires_grad_(True)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
torch.autograd.set_multithreading_enabled(False)


def step():
    W.grad = None
    lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    lv = lv[:m].requires_grad_(True)
    cv, cwrap = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
    out = L.SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
    out.backward(G)


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
