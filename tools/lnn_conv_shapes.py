#!/usr/bin/env python3
"""Prints the (vertices, val_dim, nr_filters) of every lattice convolution of one LNN forward (tools/bench_lnn.py presets)."""
import os, sys, tempfile, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lattice_net_amd as L
from lattice_net_amd import ModelParams, synthetic
from lattice_net_amd.models import LNN
from bench_lnn import PRESETS
name = sys.argv[1] if len(sys.argv) > 1 else "kitti"
preset = PRESETS[name]
with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
    f.write(preset["cfg"])
mp = ModelParams.create(f.name)
lattice = L.Lattice.create(f.name, "lattice")
os.unlink(f.name)
net = LNN(preset["classes"], mp)
n = preset["n"]
gen = {"lidar": synthetic.lidar_cloud, "box": synthetic.box_surface_cloud, "planes": synthetic.planes_cloud}[preset["cloud"]]
dev = torch.device("cuda", 0)
pos = torch.from_numpy(gen(n, 0)).to(dev)
vals = torch.zeros((n, 1), device=dev) if preset["values"] == 1 else torch.rand((n, preset["values"]), device=dev)
lib = L.load_library()
seen = collections.Counter()
orig = L.Lattice.convolve_im2row_standalone
def patched(self, filter_bank, dilation, nb, flip, filter_is_transposed=False):
    q = self
    nbl = nb if nb is not None else self
    m = q.nr_lattice_vertices()
    v = nbl.val_dim()
    f = filter_bank.shape[0] // 9 if filter_is_transposed else filter_bank.shape[1]
    seen[(m, v, int(f), int(lib.ln_conv_forward_workspace_bytes(m, 9, v, int(f))) > 256)] += 1
    return orig(self, filter_bank, dilation, nb, flip, filter_is_transposed)
L.Lattice.convolve_im2row_standalone = patched
with torch.no_grad():
    net(lattice, pos, vals)
for k, c in sorted(seen.items()):
    print(f"vertices {k[0]:7d}  V {k[1]:4d}  F {k[2]:4d}  split {k[3]}  x{c}")
