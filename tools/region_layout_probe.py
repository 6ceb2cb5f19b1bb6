#!/usr/bin/env python3
"""What would a region-interleaved row layout buy the row-gather kernels?  Host-side relabelling experiment (no kernel changes).

The rows of a C3 lattice are renumbered so that the unit of 192 rows (one workgroup of the T = 3 convolution kernels) with index u
holds vertices of kd region u % 8 — a workgroup b is dispatched to XCD b % 8, so every XCD then gathers (mostly) rows of its own
region.  Compared with the same arrays in the build's own order, padded to the same height: per-kernel time (dispatch-bound
events) of convolution forward, fused backward, slice forward.  python tools/region_layout_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import _lib, synthetic  # noqa: E402

dev = torch.device("cuda", 0)
lib = L.load_library()
n, v, f, sigma, cap, U = 120000, 32, 32, 0.9, 100000, 192
pos = torch.from_numpy(synthetic.lidar_cloud(n, 0)).to(dev)
vals = torch.randn((n, v), device=dev)
lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
lv, _, idx, w = L.SplatLattice.apply(lat, pos, vals)
m = lat.nr_lattice_vertices()
planes = lat.balanced_region_planes(idx)
keys = lat.m_hash_table.m_keys_tensor[:m].cpu().numpy().astype(np.int64)
half = (keys[:, 0] >= planes[0]).astype(int)
quarter = 2 * half + (keys[:, 1] >= np.where(half == 1, planes[2], planes[1])).astype(int)
region = 2 * quarter + (keys[:, 2] >= np.asarray(planes[3:7])[quarter]).astype(int)
counts = np.bincount(region, minlength=8)
units = int(np.ceil(counts.max() / U))
rows_new = units * 8 * U
print(f"m={m} region vertex counts {counts.tolist()} -> {units} units of {U} rows per region, {rows_new} rows ({rows_new / m - 1:+.1%} padding)")
perm = np.full(m, -1, np.int64)  # old row -> new row
for r in range(8):
    old = np.nonzero(region == r)[0]
    q = np.arange(old.size)
    perm[old] = ((q // U) * 8 + r) * U + (q % U)
perm_t = torch.from_numpy(perm).to(dev)
nbr = lat.neighbours(lat, 1, False)  # [m, 9]
W = (torch.rand((9 * v, f), device=dev) - 0.5) * 0.2
G = torch.randn((rows_new, f), device=dev)
Gp = torch.randn((n, f), device=dev)


def layout(relabel: bool):
    vals_l = torch.zeros((rows_new, v), device=dev)
    nbr_l = torch.full((rows_new, 9), _lib.LN_NOT_VISITED, dtype=torch.int32, device=dev)
    if relabel:
        vals_l[perm_t] = lv[:m]
        nb = nbr.long()
        nbr_l[perm_t] = torch.where(nb >= 0, perm_t[nb.clamp(min=0)], nb).int()
        idx_l = torch.where(idx >= 0, perm_t[idx.long().clamp(min=0)], idx.long()).int()
    else:
        vals_l[:m] = lv[:m]
        nbr_l[:m] = nbr
        idx_l = idx.clone()
    return vals_l.contiguous(), nbr_l.contiguous(), idx_l.contiguous()


def run(tag, relabel):
    vals_l, nbr_l, idx_l = layout(relabel)
    out = torch.empty((rows_new, f), device=dev)
    gv = torch.empty((rows_new, v), device=dev)
    gw = torch.empty((9 * v, f), device=dev)
    ws = torch.empty((int(lib.ln_conv_grad_filter_workspace_bytes(rows_new, 9, v, f)) + 256,), dtype=torch.uint8, device=dev)
    sl = torch.empty((n, f), device=dev)
    st = _lib.stream_ptr(dev)

    def once():
        _lib.check(lib.ln_conv_forward(_lib.ptr(nbr_l), _lib.ptr(vals_l), _lib.ptr(W), rows_new, 9, v, f, 0, _lib.ptr(out), st))
        _lib.check(lib.ln_conv_backward(_lib.ptr(nbr_l), _lib.ptr(nbr_l), _lib.ptr(vals_l), _lib.ptr(G), _lib.ptr(W), rows_new, rows_new, 9, v, f,
                                        _lib.ptr(gv), _lib.ptr(gw), _lib.ptr(ws), ws.numel(), st))
        _lib.check(lib.ln_slice_forward(_lib.ptr(out), _lib.ptr(idx_l), _lib.ptr(w), n, 3, f, _lib.ptr(sl), st))

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    res = {}
    for name in ("k_conv_mfma", "k_conv_backward_fused", "k_slice_forward"):
        lib.ln_profile_begin(name.encode(), 64)
        for _ in range(20):
            once()
        torch.cuda.synchronize()
        ms, cnt = C.c_double(0), C.c_int(0)
        lib.ln_profile_end(C.byref(ms), C.byref(cnt))
        res[name] = ms.value / max(cnt.value, 1) * 1e3
    print(tag, {k: round(x, 2) for k, x in res.items()}, "checksum", float(out.double().abs().sum()), float(sl.double().abs().sum()), flush=True)
    if os.environ.get("PROBE_LOOP"):  # for a counter pass: only this layout, many launches
        for _ in range(50):
            once()
        torch.cuda.synchronize()


which = os.environ.get("PROBE_ONLY", "")
if which in ("", "build"):
    run("build order      ", False)
if which in ("", "region"):
    run("region-interleaved", True)
