#!/usr/bin/env python3
"""Per-kernel timing of the hot path with the library's own HIP-event hooks (ln_profile_*).
Usage: python tools/profile_kernels.py [--cloud lidar|cube] [--n 120000] [--v 32] [--steps 10]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cloud", default="lidar")
    ap.add_argument("--n", type=int, default=120000)
    ap.add_argument("--v", type=int, default=32)
    ap.add_argument("--f", type=int, default=32)
    ap.add_argument("--sigma", type=float, default=0.9)
    ap.add_argument("--capacity", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--scale", type=float, default=1.0, help="cube half-extent")
    ap.add_argument("--regions", action="store_true", help="calibrate kd region planes (XCD-affine segment walk)")
    ap.add_argument("--scan-order", action="store_true",
                    help="sort the points by azimuth, then range (a spinning sensor's order) instead of the i.i.d. order of the generator")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = L.load_library()
    n, v, f = args.n, args.v, args.f
    if args.cloud == "lidar":
        pos_np = synthetic.lidar_cloud(n, 0)
    else:
        pos_np = synthetic.cube_cloud(n, 0, -args.scale, args.scale)
    if args.scan_order:
        az = np.arctan2(pos_np[:, 1], pos_np[:, 0])
        order = np.lexsort((np.hypot(pos_np[:, 0], pos_np[:, 1]), np.round(az / (2 * np.pi / 2048))))  # 2048 azimuth columns
        pos_np = np.ascontiguousarray(pos_np[order])
    rng = np.random.default_rng(0)
    pos = torch.from_numpy(pos_np).to(dev)
    vals = torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)
    G = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)
    W = (torch.rand((9 * v, f), device=dev) - 0.5).requires_grad_(True)
    lat = L.Lattice(sigmas=[args.sigma] * 3, capacity=args.capacity, device=dev)
    st = {}

    def step():
        W.grad = None
        lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
        m = lat.nr_lattice_vertices()
        lv = lv[:m].requires_grad_(True)
        cv, cwrap = L.ConvIm2RowLattice.apply(lv, lat, W, 1)
        out = L.SliceLattice.apply(cv, cwrap.lattice, pos, idx, w)
        out.backward(G)
        st["m"] = m
        st["idx"] = idx

    for _ in range(3):
        step()
    if args.regions:
        lat.set_region_planes(lat.balanced_region_planes(st["idx"]))
        for _ in range(2):
            step()
    torch.cuda.synchronize()
    names = lib.ln_kernel_names().decode().split(",")
    rows = []
    for name in names:
        lib.ln_profile_begin(name.encode(), 16 * args.steps)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms, cnt = C.c_double(0), C.c_int(0)
        lib.ln_profile_end(C.byref(ms), C.byref(cnt))
        if cnt.value:
            rows.append((name, cnt.value / args.steps, ms.value / cnt.value * 1e3, ms.value / args.steps * 1e3))
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps * 1e6
    rows.sort(key=lambda r: -r[3])
    print(f"cloud={args.cloud} n={n} m={st['m']} v={v} f={f}  wall/step={wall:.1f} us  sum(kernels)={sum(r[3] for r in rows):.1f} us")
    print(f"{'kernel':28s} {'launch/step':>11s} {'avg us':>9s} {'us/step':>9s}")
    for r in rows:
        print(f"{r[0]:28s} {r[1]:11.1f} {r[2]:9.2f} {r[3]:9.2f}")


if __name__ == "__main__":
    main()
