#!/usr/bin/env python3
"""Kernel time of the fused slice + classifier head (forward, backward) at the SemanticKITTI LNN shape.
Usage: python tools/bench_slice_classify.py [points] [val_dim] [classes]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L
from lattice_net_amd import synthetic
lib = L.load_library()
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
v = int(sys.argv[2]) if len(sys.argv) > 2 else 96
c = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
lat.begin_splat()
idx, w = lat.just_create_verts(pos, True)
m = lat.nr_lattice_vertices()
vals = torch.randn((m, v), device=dev, requires_grad=True)
dw = (torch.randn((n, 4), device=dev) * 0.01).requires_grad_(True)
lw = torch.randn((c, v), device=dev, requires_grad=True)
lb = torch.zeros((c,), device=dev, requires_grad=True)
g = torch.randn((n, c), device=dev)
def step():
    for t in (vals, dw, lw, lb):
        t.grad = None
    L.SliceClassifyLattice.apply(vals, lat, pos, dw, lw, lb, c, idx, w).backward(g)
for _ in range(3):
    step()
for k in ("k_slice_classify_forward", "k_slice_classify_backward", "k_sc_reduce_slabs", "k_csr_reduce_segments"):
    lib.ln_profile_begin(k.encode(), 64)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ms, cnt = C.c_double(0.0), C.c_int(0)
    lib.ln_profile_end(C.byref(ms), C.byref(cnt))
    print(f"{k:28s} {ms.value / max(cnt.value, 1) * 1e3:7.1f} us x{cnt.value // 5}")
