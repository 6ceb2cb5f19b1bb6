"""Diagnostic: which (channel, filter) pairs a convolution build actually multiplies (one-hot filter bank, centre slot only)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import lattice_net_amd as L
from lattice_net_amd import synthetic
dev = torch.device("cuda", 0)
pos = torch.from_numpy(synthetic.lidar_cloud(120000, 0)).to(dev)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
lat.begin_splat()
dl, _, _, _ = lat.distribute(pos, torch.zeros((120000, 1), device=dev))
m = dl.nr_lattice_vertices()
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(32, 64), (64, 32)]
for v, f in shapes:
    vals = torch.zeros((m, v), device=dev)
    vals[:] = torch.arange(1, v + 1, device=dev, dtype=torch.float32)[None, :]
    bank = torch.zeros((9, v, f), device=dev)
    for ff in range(f):
        bank[8, ff % v, ff] = 1.0 + ff // v      # out[row][ff] = (1 + ff // v) * vals[row][ff % v]  (slot 8 = the vertex itself)
    dl.set_values(vals)
    y = dl.convolve_im2row_standalone(bank.reshape(9 * v, f), 1, dl, False).values()
    exp = torch.tensor([(1.0 + ff // v) * (ff % v + 1) for ff in range(f)])
    got = y[1000].cpu()
    bad = [(ff, int(got[ff]), int(exp[ff])) for ff in range(f) if got[ff] != exp[ff]]
    print(f"V {v} F {f}: {len(bad)} wrong filters (filter, got, expected):", bad[:48])
