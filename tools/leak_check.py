#!/usr/bin/env python3
"""Device-memory growth over many steps of the headline chain and of the LNN training step (allocated / reserved bytes and
host RSS at the start and at the end).  Usage: python tools/leak_check.py [steps]"""
import os, resource, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lattice_net_amd as L
from lattice_net_amd import ModelParams, synthetic
from lattice_net_amd.losses import nll_loss_gather
from lattice_net_amd.models import LNN
from bench_lnn import PRESETS

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
def snap(tag):
    import gc
    gc.collect()
    torch.cuda.synchronize()
    print(f"{tag:28s} allocated {torch.cuda.memory_allocated() / 2**20:9.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:9.1f} MiB  "
          f"host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:8.1f} MiB", flush=True)

n = 120000
clouds = [torch.from_numpy(synthetic.lidar_cloud(n - 1000 * k, k)).to(dev) for k in range(4)]  # different sizes: buffers get re-made
vals = torch.randn((n, 32), device=dev)
G = torch.randn((n, 32), device=dev)
W = (torch.rand((288, 32), device=dev) - 0.5).requires_grad_(True)
lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
def chain(k):
    pos = clouds[k % 4]
    W.grad = None
    lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals[: pos.shape[0]])
    m = lat.nr_lattice_vertices()
    cv, cw = L.ConvIm2RowLattice.apply(lv[:m].requires_grad_(True), lat, W, 1)
    L.SliceLattice.apply(cv, cw.lattice, pos, idx, w).backward(G[: pos.shape[0]])
for k in range(20):
    chain(k)
snap("chain: after 20 steps")
for k in range(steps):
    chain(k)
snap(f"chain: after {steps + 20} steps")

preset = PRESETS["kitti"]
with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
    f.write(preset["cfg"])
mp = ModelParams.create(f.name)
lattice = L.Lattice.create(f.name, "lattice")
os.unlink(f.name)  # the readers are done with the temporary cfg
net = LNN(preset["classes"], mp)
targets = [torch.from_numpy(np.random.default_rng(k).integers(0, 20, c.shape[0])).to(dev) for k, c in enumerate(clouds)]
opt = None
def train(k):
    global opt
    pos = clouds[k % 4]
    ls, _ = net(lattice, pos, torch.zeros((pos.shape[0], 1), device=dev))
    loss = nll_loss_gather(ls, targets[k % 4])
    if opt is None:
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, fused=True)
    opt.zero_grad()
    loss.backward()
    opt.step()
for k in range(12):
    train(k)
snap("LNN: after 12 steps")
for rep in range(4):
    for k in range(steps // 4):
        train(k)
    snap(f"LNN: after {(rep + 1) * (steps // 4) + 12} steps")
