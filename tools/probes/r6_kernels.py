#!/usr/bin/env python3
"""Per-launch times (ln_profile_begin("*"): event pairs bound to each dispatch) of the C3 chain's kernels under both slot orders, with the
fused post-build launch taken apart (segment reduce and neighbour traversal as launches of their own).  Round 6 probe; not a product path."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lattice_net_amd as L
from lattice_net_amd import lattice as LT, synthetic, _lib

dev = torch.device("cuda", 0)
lib = L.load_library()
torch.autograd.set_multithreading_enabled(False)
n, v, f, sigma, cap = 120000, 32, 32, 0.9, int(os.environ.get('CAP', '100000'))
rng = np.random.default_rng(0)
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
cal = torch.from_numpy(synthetic.lidar_cloud(n, 77)).to(dev)
vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
W = ((torch.rand((9 * v, f), device=dev) * 2 - 1) * 0.4).requires_grad_(True)
reps = int(os.environ.get("REPS", "60"))

def table(fn, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    assert lib.ln_profile_begin(b"*", 64 * reps) == 0
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16)
    lib.ln_profile_end_table(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt) / reps, float(ms) / int(cnt) * 1e3)
    return out

for order in (sys.argv[1:] or ["hash", "space"]):
    LT.set_slot_order(order)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
    _, _, idx, _ = L.SplatLattice.apply(lat, cal, vals)
    lat.nr_lattice_vertices()
    lat.calibrate_regions(idx, vertex_weight=float(os.environ.get("VW", "0")))

    def step():
        W.grad = None
        lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
        m = lat.nr_lattice_vertices()
        lv = lv[:m].requires_grad_(True)
        cv, cw = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
        out = L.SliceLattice.apply(cv, cw.lattice, pos, idx, w)
        out.backward(G)

    t = table(step)
    print(f"== {order}: chain   " + "  ".join(f"{k} {c:.0f}x{us:.2f}" for k, (c, us) in t.items()), "| sum", round(sum(c * us for c, us in t.values()), 1))
    lat.prefetch_neighbours = False

    def parts():
        lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
        m = lat.nr_lattice_vertices()
        lat.m_hash_table._storage.nbr_cache.clear()
        lat.neighbours(lat, 1, False)

    t = table(parts)
    print(f"== {order}: apart   " + "  ".join(f"{k} {c:.0f}x{us:.2f}" for k, (c, us) in t.items()))
    lat.prefetch_neighbours = True
