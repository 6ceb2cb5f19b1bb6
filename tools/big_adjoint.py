#!/usr/bin/env python3
"""Large-size property check of the backward kernels.  The chain splat -> conv -> coarsen -> finefy -> slice is linear in the
vertex values and in each filter bank separately, so with L = <out, G>:  <W, dL/dW> = L for every bank and <lv, dL/dlv> = L.
Usage: python tools/big_adjoint.py [points]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattice_net_amd as L

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
for (v, f1, fc, sigma, cap) in [(32, 32, 64, 0.06, 6_000_000), (8, 16, 24, 0.03, 14_500_000), (64, 64, 128, 0.12, 1_500_000)]:
    pos = (torch.rand((n, 3), device=dev) - 0.5) * 4.0
    vals = torch.randn((n, v), device=dev)
    G = torch.randn((n, f1), device=dev)
    lat = L.Lattice(sigmas=[sigma] * 3, capacity=cap, device=dev)
    e = lat.get_filter_extent(1)
    W1 = ((torch.rand((e * v, f1), device=dev) - 0.5) * 0.2).requires_grad_(True)
    Wc = ((torch.rand((e * f1, fc), device=dev) - 0.5) * 0.2).requires_grad_(True)
    Wf = ((torch.rand((e * fc, f1), device=dev) - 0.5) * 0.2).requires_grad_(True)
    lv, wrap, idx, w = L.SplatLattice.apply(lat, pos, vals)
    m = lat.nr_lattice_vertices()
    lv = lv[:m].detach().requires_grad_(True)
    c1, s1 = L.ConvIm2RowLattice.apply(lv, lat, W1, 1)
    c2, s2 = L.CoarsenLattice.apply(c1, s1.lattice, Wc)
    mc = s2.lattice.nr_lattice_vertices()
    c3, s3 = L.FinefyLattice.apply(c2, s2.lattice, s1.lattice, Wf)
    out = L.SliceLattice.apply(c3, s3.lattice, pos, idx, w)
    loss = (out.double() * G.double()).sum()
    out.backward(G)
    ref = float(loss)
    line = [f"n={n} v={v} m={m} coarse={mc} L={ref:.6e}"]
    worst = 0.0
    for name, t in (("lv", lv), ("W1", W1), ("Wc", Wc), ("Wf", Wf)):
        got = float((t.detach().double() * t.grad.double()).sum())
        rel = abs(got - ref) / max(abs(ref), 1e-30)
        worst = max(worst, rel)
        line.append(f"{name} {rel:.1e}")
    print("  ".join(line), "OK" if worst < 2e-3 else "MISMATCH", flush=True)
    del lv, c1, c2, c3, out, lat, s1, s2, s3, wrap
    torch.cuda.empty_cache()
