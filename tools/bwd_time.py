import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import lattice_net_amd as L
from lattice_net_amd import synthetic
dev = torch.device("cuda", 0)
n, v = 120000, 32
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.randn((n, v), device=dev)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
m = lat.nr_lattice_vertices()
lvm = lv[:m].contiguous()
lat.set_values(lvm)
W = (torch.rand((9 * v, v), device=dev) - 0.5)
G = torch.randn((m, v), device=dev)
def run():
    return lat.convolve_im2row_backward(G, W, 1, lat, lat)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print("backward call: %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
