import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import lattice_net_amd as L
from lattice_net_amd import synthetic
dev = torch.device("cuda", 0)
n, v, f = int(os.environ.get("HP_N", "2500")), 32, 32
rng = np.random.default_rng(0)
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
W = (torch.rand((9 * v, f), device=dev) - 0.5).requires_grad_(True)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
torch.autograd.set_multithreading_enabled(False)
T = [0.0] * 7
def step(rec):
    t0 = time.perf_counter()
    W.grad = None
    lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
    t1 = time.perf_counter()
    m = lat.nr_lattice_vertices()
    t2 = time.perf_counter()
    lv = lv[:m].requires_grad_(True)
    t3 = time.perf_counter()
    cv, cwrap = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
    t4 = time.perf_counter()
    out = L.SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
    t5 = time.perf_counter()
    out.backward(G)
    t6 = time.perf_counter()
    if rec:
        for i, (a, b) in enumerate([(t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6)]):
            T[i] += b - a
for _ in range(50): step(False)
torch.cuda.synchronize()
K = 500
t = time.perf_counter()
for _ in range(K): step(True)
torch.cuda.synchronize()
tot = (time.perf_counter() - t) / K * 1e6
names = ["splat.apply", "nr_vertices(sync)", "slice+requires_grad", "conv.apply", "slice.apply", "backward"]
print(f"n={n} total {tot:.1f} us/step")
for nm, x in zip(names, T): print(f"  {nm:22s} {x / K * 1e6:7.1f} us")
