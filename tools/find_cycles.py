#!/usr/bin/env python3
"""Which reference cycles an LNN training step leaves behind (they keep device tensors alive until Python's cycle collector
runs): collects with DEBUG_SAVEALL after a few steps with the collector off and prints the garbage by type."""
import collections, gc, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lattice_net_amd as L
from lattice_net_amd import ModelParams, synthetic
from lattice_net_amd.losses import nll_loss_gather
from lattice_net_amd.models import LNN
from bench_lnn import PRESETS
dev = torch.device("cuda", 0)
preset = PRESETS["kitti"]
with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
    f.write(preset["cfg"])
mp = ModelParams.create(f.name)
lattice = L.Lattice.create(f.name, "lattice")
os.unlink(f.name)  # the readers are done with the temporary cfg
net = LNN(preset["classes"], mp)
n = 120000
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
target = torch.from_numpy(np.random.default_rng(0).integers(0, 20, n)).to(dev)
vals = torch.zeros((n, 1), device=dev)
opt = None
def train():
    global opt
    ls, _ = net(lattice, pos, vals)
    loss = nll_loss_gather(ls, target)
    if opt is None:
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, fused=True)
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(3):
    train()
gc.collect()
gc.disable()
before = torch.cuda.memory_allocated()
for _ in range(3):
    train()
torch.cuda.synchronize()
print(f"allocated grew by {(torch.cuda.memory_allocated() - before) / 2**20:.1f} MiB over 3 steps with the collector off")
gc.set_debug(gc.DEBUG_SAVEALL)
found = gc.collect()
print("objects in cycles:", found)
types = collections.Counter(type(o).__name__ for o in gc.garbage)
print(types.most_common(15))
for o in gc.garbage:
    if type(o).__name__ in ("function", "cell", "Lattice", "HashTable", "_TableStorage", "LatticeWrapper") or "Backward" in type(o).__name__:
        desc = getattr(o, "__qualname__", None) or type(o).__name__
        print("  ", type(o).__name__, desc)
gc.set_debug(0)
gc.garbage.clear()
