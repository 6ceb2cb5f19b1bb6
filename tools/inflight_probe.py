#!/usr/bin/env python3
"""Throughput with B independent clouds in flight: B (lattice, hipGraph, stream) sets, steps issued round-robin."""
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
n, v, f, sigma, cap = 120000, 32, 32, 0.9, 100000
W = (torch.rand((9 * v, f), device=dev) - 0.5).requires_grad_(True)


class Set:
    def __init__(self, k):
        rng = np.random.default_rng(k)
        self.pos = torch.from_numpy(synthetic.lidar_cloud(n, k)).to(dev)
        self.vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
        self.G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
        self.lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
        self.lat.set_static_rows(49408)
        self.stream = torch.cuda.Stream()
        self.out = None

    def step(self):
        W.grad = None
        lv, wrap, idx, w = L.SplatLattice.apply(self.lat, self.pos, self.vals)
        m = self.lat.nr_lattice_vertices()
        lv = lv[:m].requires_grad_(True)
        cv, cwrap = L.ConvIm2RowLattice.apply(lv, self.lat, W, 1)
        out = L.SliceLattice.apply(cv, cwrap.lattice, self.pos, idx, w)
        out.backward(self.G)
        self.out = out

    def capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.step()
            self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.out = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.step()
        self.grad = W.grad


sets = [Set(k) for k in range(4)]
for s in sets:
    s.capture()
torch.cuda.synchronize()
for B in (1, 2, 3, 4):
    for rep in range(2):
        torch.cuda.synchronize()
        K = 400
        t0 = time.perf_counter()
        for i in range(K):
            s = sets[i % B]
            with torch.cuda.stream(s.stream):
                s.graph.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
    print(f"B={B}: {dt * 1e6:7.1f} us per cloud  -> {n / dt / 1e6:7.1f} Mpoints/s")
for s in sets:
    print(s.lat.static_build_report())
