#!/usr/bin/env python3
"""Per-slot convolution kernel (k_conv_mfma) at the SemanticKITTI network's widths on the C3 lattice: per-launch time from the
library's own event pairs, and the result against an fp64 gather-matmul of the same operands.

    LATTICE_NET_LIB=lattice_net_amd/liblatticenet_hip_<variant>.so python tools/conv_time.py [--reps 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from ops_roofline import _profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", default="64x64,128x128,64x128,128x64,96x96,32x64,256x256")
    ap.add_argument("--coarse", type=int, default=0, help="time on the level-2 (1) or level-3 (2) lattice of the same cloud instead of the finest")
    ap.add_argument("--row-order", default="", help="canonical: rows numbered in first-occurrence order (ln_canonicalize) instead of slot order")
    a = ap.parse_args()
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    dev = torch.device("cuda", 0)
    lib = L.load_library()
    if a.row_order:
        from lattice_net_amd import lattice as _lat
        _lat.set_row_order(a.row_order)
    pos = torch.from_numpy(synthetic.lidar_cloud(120000, 0)).to(dev)
    lat = L.Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
    lat.begin_splat()
    dl, _, _, _ = lat.distribute(pos, torch.zeros((120000, 1), device=dev))
    for _ in range(a.coarse):
        dl = dl.create_coarse_verts_naive(pos)
    m = dl.nr_lattice_vertices()
    nbr = dl.neighbours(dl, 1, False).long()
    torch.manual_seed(1)
    for shp in a.shapes.split(","):
        v, f = (int(x) for x in shp.split("x"))
        vals = torch.randn((m, v), device=dev)
        bank = torch.randn((9 * v, f), device=dev) * 0.05
        dl.set_values(vals)
        state = {}

        def fwd():
            state["y"] = dl.convolve_im2row_standalone(bank, 1, dl, False).values()
        k = _profile(lib, fwd, a.reps)
        # fp64 check on a sample of rows
        rows = torch.randint(0, m, (2048,), device=dev)
        nb = nbr[rows]                                              # [R, 9]
        g = torch.where((nb >= 0)[..., None], vals.double()[nb.clamp(min=0)], torch.zeros((), device=dev, dtype=torch.double))
        ref = g.reshape(len(rows), 9 * v) @ bank.double()
        err = (state["y"][rows].double() - ref).abs().max().item() / ref.abs().max().item()
        ks = ", ".join(f'{x["kernel"]} {x["launches_per_call"]:g} x {x["avg_us"]:.1f} us' for x in k)
        tot = sum(x["us_per_call"] for x in k)
        print(f"V {v:3d} -> F {f:3d}  m {m}: {tot:7.1f} us/call  [{ks}]  fp32-equivalent {2.0 * m * 9 * v * f / tot / 1e6:6.1f} TFLOP/s  rel err {err:.2e}")


if __name__ == "__main__":
    main()
