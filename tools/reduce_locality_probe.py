#!/usr/bin/env python3
"""What bounds the segment reduce?  The same launch (480 k tokens onto 46.6 k rows, V = 32) over three adjacencies:
  real     the splat indices of the benchmark cloud (i.i.d. point order: every token gathers a random 128-byte row);
  local    token t -> row (t / 4) % M: consecutive segments read consecutive point rows, the 4 tokens of a point sit in one segment;
  uniform  random rows, uniform (no hot vertices).
Time per launch from hipEvents over 200 launches.  python tools/reduce_locality_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402

dev = torch.device("cuda", 0)
n, v, sigma, cap = 120000, 32, 0.9, 100000
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.randn((n, v), device=dev)
lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
m = lat.nr_lattice_vertices()
t = torch.arange(4 * n, device=dev, dtype=torch.int64)
cases = {"real": idx.clone(),
         "local": ((t // 4) % m).to(torch.int32),
         "uniform": torch.randint(0, m, (4 * n,), device=dev, dtype=torch.int32)}
for name, ix in cases.items():
    dst = torch.zeros((m, v), device=dev)
    for _ in range(5):
        dst.zero_()
        lat._scatter_rows(vals, ix, w, dst, v, 4, v)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        dst.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lat._scatter_rows(vals, ix, w, dst, v, 4, v)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ref = torch.zeros((m, v), device=dev, dtype=torch.float64)
    ref.index_add_(0, ix.long(), (vals.repeat_interleave(4, 0) * w[:, None]).double())
    err = float((dst.double() - ref).abs().max() / ref.abs().max())
    print(f"{name:8s} median {np.median(ts):6.1f} us  p10 {np.percentile(ts, 10):6.1f}   (max rel err {err:.1e})", flush=True)
