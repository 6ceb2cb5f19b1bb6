#!/usr/bin/env python3
"""Does an independent lattice build overlap with the rest of a step when both sit in one hipGraph (fork / join)?
Replays: (a) the full step alone, (b) a build alone, (c) full step with a second lattice's build forked onto a side stream."""
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
rng = np.random.default_rng(0)
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
pos2 = torch.from_numpy(synthetic.lidar_cloud(n, 1)).to(dev)
vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
W = (torch.rand((9 * v, f), device=dev) - 0.5).requires_grad_(True)
lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
lat2 = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
lat.set_static_rows(49408)
lat2.set_static_rows(49408)


def full():
    W.grad = None
    lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    lv = lv[:m].requires_grad_(True)
    cv, cwrap = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
    out = L.SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
    out.backward(G)


def build2():
    lat2.begin_splat()
    lat2.just_create_verts(pos2, True)


side2 = torch.cuda.Stream()


def both():
    cur = torch.cuda.current_stream()
    side2.wait_stream(cur)
    with torch.cuda.stream(side2):
        build2()
    full()
    cur.wait_stream(side2)


GRAPHS = {}


def bench(fn, name):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us / replay")
    GRAPHS[name] = g


bench(full, "full step")
bench(build2, "build only")
bench(both, "full step || second build")

# two graphs on two streams, launched back to back
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ga, gb = GRAPHS["build only"], GRAPHS["full step"]
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(200):
        with torch.cuda.stream(s1):
            ga.replay()
        with torch.cuda.stream(s2):
            gb.replay()
    torch.cuda.synchronize()
    print(f"two graphs on two streams    {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us / pair")
# eager launches of the build on a second stream while the full-step graph replays
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    with torch.cuda.stream(s2):
        gb.replay()
    with torch.cuda.stream(s1):
        build2()
torch.cuda.synchronize()
print(f"graph(full) || eager build   {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us / pair")
