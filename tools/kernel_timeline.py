#!/usr/bin/env python3
"""Phase timeline of k_point_keys / k_bucket_rows from per-workgroup wall-clock stamps (needs the -DLN_STAMPS build:
LATTICE_NET_LIB=lattice_net_amd/liblatticenet_hip_stamps.so python tools/kernel_timeline.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402

lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda", 0)
n, v, sigma, cap = 120000, 32, 0.9, 100000
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.randn((n, v), device=dev)
L.set_row_order("slot")
lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
lat.prefetch_neighbours = False
stamps = torch.zeros((4096, 24), dtype=torch.int64, device=dev)
for _ in range(5):
    L.SplatLattice.apply(lat, pos, vals)
    lat.nr_lattice_vertices()
torch.cuda.synchronize()
lib.ln_debug_set_stamps.argtypes = [C.c_void_p]
assert lib.ln_debug_set_stamps(stamps.data_ptr()) == 0
acc = {}
reps = 10
for _ in range(reps):
    stamps.zero_()
    L.SplatLattice.apply(lat, pos, vals)
    lat.nr_lattice_vertices()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64) / 100.0  # 100 MHz -> microseconds
    for name, first, last, nwg in (("k_point_keys", 0, 7, None), ("k_bucket_rows", 8, 17, None)):
        rows = s[(s[:, first] > 0)]
        t0 = rows[:, first].min()
        rel = rows[:, first:last + 1] - t0
        acc.setdefault(name, []).append(rel)
lib.ln_debug_set_stamps(None)
labels = {"k_point_keys": ["start", "clear issued+sync", "keys+LDS rank done", "sync", "scan+global atomics", "sync", "LDS staging+sync", "stores issued (end)"],
          "k_bucket_rows": ["start", "init+loads+sync", "place (LDS CAS/add/min)", "scans + publish", "look-back done", "emit slots", "token stores", "(segment ids done, before look-back)", "(scans done, before publish)", "(minima compacted, before the in-bucket rank)"]}
for name, runs in acc.items():
    m = np.mean([r.mean(0) for r in runs], 0)
    mx = np.mean([r.max(0) for r in runs], 0)
    mn = np.mean([r.min(0) for r in runs], 0)
    print(f"{name}: {runs[0].shape[0]} workgroups; microseconds since the first workgroup started")
    prev = 0.0
    for k, lab in enumerate(labels[name]):
        print(f"  {lab:28s} mean {m[k]:7.2f}  (+{m[k] - prev:5.2f})   earliest {mn[k]:7.2f}  latest {mx[k]:7.2f}")
        prev = m[k]

# ---- small-filter convolution (forward): one launch, stamps per workgroup
if hasattr(lib, "ln_debug_set_stamps_conv") and os.environ.get("LN_TIMELINE_CONV"):
    W = (torch.rand((9 * v, v), device=dev) - 0.5)
    lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    lvm = lv[:m].contiguous()
    cst = torch.zeros((4096, 8), dtype=torch.int64, device=dev)
    lib.ln_debug_set_stamps_conv.argtypes = [C.c_void_p]
    runs = []
    for _ in range(reps + 2):
        cst.zero_()
        lib.ln_debug_set_stamps_conv(cst.data_ptr())
        L.ConvIm2RowLattice.apply(lvm, lat, W, 1)
        torch.cuda.synchronize()
        lib.ln_debug_set_stamps_conv(None)
        s = cst.cpu().numpy().astype(np.float64) / 100.0
        rows = s[s[:, 0] > 0]
        runs.append(rows[:, :4] - rows[:, 0].min())
    runs = runs[2:]
    m_ = np.mean([r.mean(0) for r in runs], 0); mx = np.mean([r.max(0) for r in runs], 0); mn = np.mean([r.min(0) for r in runs], 0)
    print(f"k_conv_mfma_full: {runs[0].shape[0]} workgroups")
    prev = 0.0
    for k, lab in enumerate(["start", "ids + gathers issued, bank staged, sync", "MFMA loop done", "stores issued (end)"]):
        print(f"  {lab:44s} mean {m_[k]:7.2f}  (+{m_[k] - prev:5.2f})   earliest {mn[k]:7.2f}  latest {mx[k]:7.2f}")
        prev = m_[k]

    # ---- fused backward of the same convolution (k_conv_backward_fused)
    lvm.requires_grad_(True)
    Wg = W.clone().requires_grad_(True)
    Gc = torch.randn((m, v), device=dev)
    runs = []
    for _ in range(reps + 2):
        out, _ = L.ConvIm2RowLattice.apply(lvm, lat, Wg, 1)
        torch.cuda.synchronize()
        cst.zero_()
        lib.ln_debug_set_stamps_conv(cst.data_ptr())
        out.backward(Gc)
        torch.cuda.synchronize()
        lib.ln_debug_set_stamps_conv(None)
        s = cst.cpu().numpy().astype(np.float64) / 100.0
        rows = s[s[:, 0] > 0]
        runs.append(rows[:, :6] - rows[:, 0].min())
    runs = runs[2:]
    m_ = np.mean([r.mean(0) for r in runs], 0); mx = np.mean([r.max(0) for r in runs], 0); mn = np.mean([r.min(0) for r in runs], 0)
    print(f"backward (k_conv_backward_fused when enabled): {runs[0].shape[0]} workgroups")
    for k, lab in ((0, "start"), (1, "bank + G_0 staged, first barrier"), (4, "slot 0 MFMAs done, barrier 1"), (5, "barrier 8"), (2, "slot loop done"), (3, "end")):
        print(f"  {lab:44s} mean {m_[k]:7.2f}   earliest {mn[k]:7.2f}  latest {mx[k]:7.2f}")
