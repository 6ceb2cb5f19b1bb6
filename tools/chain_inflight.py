#!/usr/bin/env python3
"""splat -> slice (forward) as one hipGraph per scan, K scans in flight: microseconds per scan and fraction of the HBM peak
against SURVEY 8d's algorithmic bytes.  python tools/chain_inflight.py --in-flight 1,2,3,4,6 [--prefetch 1] [--reps 600]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L  # noqa: E402
from lattice_net_amd import synthetic  # noqa: E402
from lattice_net_amd.capture import CapturedStep, concurrent_streams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--in-flight", default="1,3")
ap.add_argument("--prefetch", type=int, default=0)
ap.add_argument("--reps", type=int, default=600)
ap.add_argument("--regions", type=int, default=1)
ap.add_argument("--n", type=int, default=120000)
args = ap.parse_args()
dev = torch.device("cuda", 0)
n, v, d, sigma, cap = args.n, 32, 3, 0.9, 100000
kmax = max(int(k) for k in args.in_flight.split(","))
chains = []
streams = concurrent_streams(kmax + 1)[1:]  # streams on different hardware queues (capture.concurrent_streams)
for k in range(kmax):
    rng = np.random.default_rng(1000 * k)
    c = {"lat": L.Lattice(sigmas=[sigma] * d, capacity=cap, device=dev), "st": {},
         "pos": torch.from_numpy(synthetic.lidar_cloud(n, k)).to(dev),
         "vals": torch.from_numpy(rng.standard_normal((n, v)).astype(np.float32)).to(dev)}
    c["lat"].prefetch_neighbours = bool(args.prefetch)

    def chain(c=c):
        with torch.no_grad():
            lv2, _, idx2, w2 = L.SplatLattice.apply(c["lat"], c["pos"], c["vals"])
            m2 = c["lat"].nr_lattice_vertices()
            c["st"].update(idx=idx2, m=m2, out=L.SliceLattice.apply(lv2[:m2], c["lat"], c["pos"], idx2, w2))

    chain()
    torch.cuda.synchronize()
    ref = c["st"]["out"].detach().clone()
    c["m"] = c["st"]["m"]
    c["cap"] = CapturedStep(chain, [c["lat"]], row_slack=0.06, regions=bool(args.regions), region_indices=lambda c=c: c["st"]["idx"],
                            stream=streams[k], before_capture=c["st"].clear)
    c["cap"].launch()
    torch.cuda.synchronize()
    err = float((c["st"]["out"] - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
    assert err <= 1e-5, err
    chains.append(c)
m = chains[0]["m"]
nbytes = n * (4.0 * d + 4.0 * v + 8.0 * (d + 1)) + m * (4.0 * d + 4.0 * v) + n * (8.0 * (d + 1) + 4.0 * v) + m * 4.0 * v
for ks in args.in_flight.split(","):
    k = int(ks)
    for i in range(30):
        chains[i % k]["cap"].launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.reps):
        chains[i % k]["cap"].launch()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / args.reps * 1e6
    for c in chains[:k]:
        c["cap"].check()
    print(f"in_flight={k} prefetch={args.prefetch}: {us:7.1f} us/scan  {nbytes / us / 1e3:8.1f} GB/s  frac_of_hbm_peak={nbytes / us / 1e3 / 8000:.4f}", flush=True)
