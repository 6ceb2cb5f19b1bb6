#!/usr/bin/env python3
"""Times the two table-build paths (bucketed LDS build vs. the atomic path) over table sizes: which one the library should pick."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L
from lattice_net_amd import lattice as LM
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (n, sigma, cap) in [(120_000, 0.06, 400_000), (500_000, 0.04, 1_000_000), (500_000, 0.04, 2_000_000), (1_000_000, 0.03, 5_000_000), (200_000, 0.08, 5_000_000),
                        (4_000_000, 0.02, 12_000_000), (4_000_000, 0.02, 13_900_000)]:
    pos = (torch.rand((n, 3), device=dev) - 0.5) * 4.0
    vals = torch.randn((n, 8), device=dev)
    out = []
    for path in ("bucketed", "atomic"):
        LM._FORCE_ATOMIC_BUILD = (path == "atomic")
        lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
        for _ in range(3):
            lat.begin_splat()
            lat.splat_standalone(pos, vals)
            m = lat.nr_lattice_vertices()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lat.begin_splat()
            lat.splat_standalone(pos, vals)
            m = lat.nr_lattice_vertices()
        torch.cuda.synchronize()
        out.append(f"{path} {(time.perf_counter() - t0) * 100:.3f} ms")
    print(f"n={n} cap={cap} m={m} load={m / cap:.2f}: " + ", ".join(out), flush=True)
