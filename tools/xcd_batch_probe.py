#!/usr/bin/env python3
"""Does giving every scan of a batch its own XCD pay?  (GPU box.)

Block-diagonal batch of B = 8 independent C3 scans run through the EXISTING kernels as one big problem, in two layouts:
  A  scan-major: points / vertex rows of scan s contiguous -> every workgroup chunk of a scan lands on XCD (chunk % 8): each
     scan's rows are pulled through all eight L2s (what separate launches per scan do as well);
  B  XCD-interleaved: chunk j of scan s is global chunk 8 j + s (chunk = the rows one workgroup handles), so the workgroups of
     scan s all run on XCD s (block b -> XCD b % 8) and its vertex rows live in ONE 4 MB L2.
Kernels probed: slice forward (32 points per workgroup at V = 32), convolution forward (64 vertex rows per workgroup).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lattice_net_amd import Lattice, _lib, synthetic  # noqa: E402

B = int(os.environ.get("PROBE_B", 8))
N, V, F, E = 120000, 32, 32, 9
dev = torch.device("cuda", 0)
lib = _lib.load()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def interleave_rows(per_scan, chunk, width_rows):
    """per_scan: list of [rows_s, ...] arrays, all padded to width_rows (multiple of chunk).  Returns the interleaved array and
    the map old (s, r) -> new row."""
    nch = width_rows // chunk
    out = np.zeros((B * width_rows,) + per_scan[0].shape[1:], per_scan[0].dtype)
    for s, a in enumerate(per_scan):
        v = out.reshape((nch, B, chunk) + a.shape[1:])
        v[:, s] = a.reshape((nch, chunk) + a.shape[1:])
    return out


def main():
    scans = []
    for s in range(B):
        pos = torch.from_numpy(synthetic.lidar_cloud(N, s)).to(dev)
        lat = Lattice(sigmas=[0.9] * 3, capacity=100000, device=dev)
        lat.begin_splat()
        idx, w = lat.just_create_verts(pos, True)
        m = lat.nr_lattice_vertices()
        nbr = lat.neighbours(lat, 1, False)[:m].clone()
        scans.append(dict(m=m, idx=idx.cpu().numpy().reshape(N, 4), w=w.cpu().numpy().reshape(N, 4), nbr=nbr.cpu().numpy()))
    mpad = (max(s["m"] for s in scans) + 63) // 64 * 64
    rng = np.random.default_rng(0)
    vals = [np.concatenate([rng.standard_normal((s["m"], V)).astype(np.float32), np.zeros((mpad - s["m"], V), np.float32)]) for s in scans]
    Wt = torch.from_numpy((rng.standard_normal((E * V, F)) / 17).astype(np.float32)).to(dev)
    st = _lib.stream_ptr(dev)

    def row_map_A(s, r):
        return np.where(r >= 0, r + s * mpad, r)

    def row_map_B(s, r):
        return np.where(r >= 0, ((r // 64) * B + s) * 64 + r % 64, r)

    results = {}
    for layout, rmap in (("A scan-major", row_map_A), ("B xcd-interleaved", row_map_B)):
        # vertex side
        nbrs = []
        for s, sc in enumerate(scans):
            nb = np.full((mpad, E), -1, np.int32)
            nb[: sc["m"]] = rmap(s, sc["nbr"])
            nbrs.append(nb)
        if layout[0] == "A":
            values = np.concatenate(vals)
            nbr = np.concatenate(nbrs)
            idx = np.concatenate([rmap(s, sc["idx"]) for s, sc in enumerate(scans)]).astype(np.int32)
            w = np.concatenate([sc["w"] for sc in scans])
        else:
            values = interleave_rows(vals, 64, mpad)
            nbr = interleave_rows(nbrs, 64, mpad)
            npad = (N + 31) // 32 * 32
            pad = npad - N
            idx = interleave_rows([np.concatenate([rmap(s, sc["idx"]), np.full((pad, 4), -1)]).astype(np.int32) for s, sc in enumerate(scans)], 32, npad)
            w = interleave_rows([np.concatenate([sc["w"], np.zeros((pad, 4), np.float32)]) for sc in scans], 32, npad)
        tv, tn, ti, tw = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (values, nbr, idx, w))
        mtot, ntot = tv.shape[0], ti.shape[0]
        out_c = torch.empty((mtot, F), device=dev)
        out_s = torch.empty((ntot, V), device=dev)

        def conv():
            _lib.check(lib.ln_conv_forward(_lib.ptr(tn), _lib.ptr(tv), _lib.ptr(Wt), mtot, E, V, F, 0, _lib.ptr(out_c), st))

        def slc():
            _lib.check(lib.ln_slice_forward(_lib.ptr(out_c), _lib.ptr(ti), _lib.ptr(tw), ntot, 3, V, _lib.ptr(out_s), st))

        tc, ts = timed(conv), timed(slc)
        results[layout] = (tc, ts, float(out_c.double().abs().sum()), float(out_s.double().abs().sum()))
        print(f"{layout:20s} conv fwd {tc:8.1f} us   slice fwd {ts:8.1f} us   per scan: {tc / B:6.1f} / {ts / B:6.1f} us"
              f"   checksums {results[layout][2]:.6e} {results[layout][3]:.6e}")
    # one scan alone, for scale
    sc = scans[0]
    tv = torch.from_numpy(vals[0]).to(dev)
    tn = torch.from_numpy(np.concatenate([sc["nbr"], np.full((mpad - sc["m"], E), -1, np.int32)])).to(dev)
    ti, tw = torch.from_numpy(sc["idx"].astype(np.int32)).to(dev), torch.from_numpy(sc["w"]).to(dev)
    out_c = torch.empty((mpad, F), device=dev)
    out_s = torch.empty((N, V), device=dev)
    tc = timed(lambda: _lib.check(lib.ln_conv_forward(_lib.ptr(tn), _lib.ptr(tv), _lib.ptr(Wt), mpad, E, V, F, 0, _lib.ptr(out_c), st)))
    ts = timed(lambda: _lib.check(lib.ln_slice_forward(_lib.ptr(out_c), _lib.ptr(ti), _lib.ptr(tw), N, 3, V, _lib.ptr(out_s), st)))
    print(f"{'one scan alone':20s} conv fwd {tc:8.1f} us   slice fwd {ts:8.1f} us")


if __name__ == "__main__":
    main()
