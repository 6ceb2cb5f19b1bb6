#!/usr/bin/env python3
"""Large-size checks on the GPU box: multi-million-point clouds and tables up to 16M slots through both build paths (which
must agree bit for bit), a centre-only identity convolution and the slice of a constant field.  Usage: python tools/big_sizes.py"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import lattice_net_amd as L
from lattice_net_amd import lattice as LM
dev = torch.device("cuda", 0)
torch.manual_seed(0)
L.set_row_order("canonical")  # the two build paths are compared bit for bit: the bucketed one relabels into first-occurrence order
for (n, d, sigma, cap, v) in [(4_000_000, 3, 0.02, 12_000_000, 32), (3_000_000, 5, 0.25, 16_000_000, 8), (8_000_000, 3, 0.05, 6_000_000, 64)]:
    pos = (torch.rand((n, d), device=dev) - 0.5) * 4.0
    vals = torch.randn((n, v), device=dev)
    res = {}
    for path in ("bucketed", "atomic"):
        LM._FORCE_ATOMIC_BUILD = (path == "atomic")
        lat = L.Lattice(sigmas=[sigma] * d, capacity=cap, device=dev)
        lat.begin_splat()
        t0 = time.perf_counter()
        idx, w = lat.splat_standalone(pos, vals)
        m = lat.nr_lattice_vertices()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[path] = (m, idx.clone(), w.clone(), lat.hash_table().m_keys_tensor[:m].clone(), lat.values()[:m].clone())
        print(f"n={n} d={d} cap={cap} v={v} {path}: m={m} load={m/cap:.2f} {dt*1e3:.1f} ms idx range [{int(idx.min())},{int(idx.max())}] wsum err {float((w.view(n,d+1).sum(1)-1).abs().max()):.2e}", flush=True)
        if path == "bucketed":
            # conv with a centre-only identity filter returns the input; slice of a constant field is the constant
            e = lat.get_filter_extent(1)
            vv = min(v, 16)
            W = torch.zeros((e * vv, vv), device=dev)
            W[(e - 1) * vv:, :] = torch.eye(vv, device=dev)
            x = torch.randn((m, vv), device=dev)
            lat.set_values(x)
            out = lat.convolve_im2row_standalone(W, 1, lat, False).values()
            print("   identity conv max err", float((out - x).abs().max()))
            lat.set_values(torch.full((m, 4), 2.5, device=dev))
            s = lat.slice_standalone_with_precomputation(pos, idx, w)
            print("   constant slice max err", float((s - 2.5).abs().max()))
            nb = lat.im2rowindices(lat, e, 1, False) if hasattr(lat, "im2rowindices") else None
    a, b = res["bucketed"], res["atomic"]
    same = a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    rel = float((a[4] - b[4]).abs().max() / a[4].abs().max())
    print(f"   paths agree: {same}, values rel diff {rel:.2e}", flush=True)
    del res, a, b
    torch.cuda.empty_cache()
