#!/usr/bin/env python3
"""Probe: how much would the segment reduce gain if the segment list of every kd region were in SPATIAL order (segments of
neighbouring vertices in one workgroup, so that the 4 tokens of a point meet in one CU's vector cache)?  The descriptors a
build emitted are re-ordered on the host (stable sort by the cell of the vertex key, cell = 2^shift lattice units, cells
in Morton order or hashed) and the same launch is timed again.  python tools/segment_order_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic, _lib  # noqa: E402

dev = torch.device("cuda", 0)
n, v, sigma, cap = 120000, 32, 0.9, 100000
G = _lib.LN_XCD_GROUPS
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.randn((n, v), device=dev)
lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
lat.prefetch_neighbours = False
lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
lat.set_region_planes(lat.balanced_region_planes(idx))
lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)  # the build that files segments under the 8 regions
m = lat.nr_lattice_vertices()
st = lat.hash_table()._storage
csr_buf, csr, max_seg, grp_row, _ = lat._csr(idx)
keys = st.keys[:m].cpu().numpy().astype(np.int64)
seg_count = csr_buf[-(G + 2):].cpu().numpy()
print("segments per region", seg_count[:G].tolist(), "format", seg_count[G:].tolist(), "rows", m)
desc0 = csr_buf[: 4 * G * max_seg].clone()


def time_reduce(label):
    dst = torch.zeros((m, v), device=dev)
    for _ in range(5):
        dst.zero_()
        lat._scatter_rows(vals, idx, w, dst, v, 4, v)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        dst.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lat._scatter_rows(vals, idx, w, dst, v, 4, v)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{label:28s} median {np.median(ts):6.1f} us   p10 {np.percentile(ts, 10):6.1f}", flush=True)
    return dst.clone()


def morton(c):
    c = c - c.min(0)
    code = np.zeros(c.shape[0], np.int64)
    for b in range(12):
        for a in range(3):
            code |= ((c[:, a] >> b) & 1) << (3 * b + a)
    return code


ref = time_reduce("as emitted (bucket order)")
for shift, mode in ((2, "morton"), (3, "morton"), (4, "morton"), (3, "hashed"), (0, "morton")):
    d = desc0.cpu().numpy().reshape(G, max_seg, 4).copy()
    for g in range(G):
        c = int(seg_count[g])
        rows = d[g, :c, 0]
        cell = keys[np.maximum(rows, 0)] >> shift
        code = morton(cell)
        if mode == "hashed":
            code = (code * 2654435761) % 4096
        order = np.argsort(code, kind="stable")  # stable: the segments of one row stay adjacent and in order
        d[g, :c] = d[g, :c][order]
    csr_buf[: 4 * G * max_seg] = torch.from_numpy(d.reshape(-1)).to(dev)
    out = time_reduce(f"cells of 2^{shift} units, {mode}")
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
csr_buf[: 4 * G * max_seg] = desc0
