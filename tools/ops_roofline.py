#!/usr/bin/env python3
"""Per-operator roofline table for every SURVEY.md §8(a) row outside the headline chain, at the C3 scan (120k LiDAR-like
points, sigma 0.9, capacity 100k) with the SemanticKITTI network's widths.

    python tools/ops_roofline.py [--reps 24]             # one JSON line {"ops": [...]}
    python bench.py --workload ops                        # the same table behind bench.py's contract

Every operator is called through the Python operator surface (lattice.py / lattice_funcs.py -> C ABI) `reps` times with nothing
else on the GPU; every kernel launch of the library carries an event pair bound to the dispatch (ln_profile_begin("*") /
ln_profile_end_table: the kernel's own begin-to-end time, what rocprofv3 --kernel-trace reports), so an entry lists the launches
an operator call makes, their average durations, and

    achieved = ALGORITHMIC bytes (SURVEY.md §8d per-unit figures x units per call) / sum of the kernels' average durations
    frac     = achieved / 8 TB/s                                              (bound "hbm")
    flops    = 2 M E V F / the dense kernel's average duration against the peak of the matrix instruction it issues:
               157.3 TFLOP/s for v_mfma_f32_16x16x4_f32, 2.5 PFLOP/s for the bf16 instructions of the bf16x3 kernels (which
               execute 6 bf16 products per fp32 product: `mfma_executed_frac` prices what the pipe actually did)

profiles/r4_ops_kernel_stats.csv is the rocprofv3 --kernel-trace --stats summary of the same command."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0
N, SIGMA, CAP, D = 120000, 0.9, 100000, 3
E = 2 * (D + 1) + 1


PREWARM_MS = float(os.environ.get("LN_OPS_PREWARM_MS", "30"))


def _profile(lib, fn, reps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    # the chip needs tens of ms of load to reach the clocks of a busy GPU: the set-up between two operators of the table
    # leaves it idle, so every operator is run untimed for PREWARM_MS first (LN_OPS_PREWARM_MS=0: as before)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < PREWARM_MS:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    if lib.ln_profile_begin(b"*", 64 * reps + 64) != 0:
        raise RuntimeError(lib.ln_last_error_string())
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(16384)
    if lib.ln_profile_end_table(buf, len(buf)) != 0:  # (also when launches went unsampled or rows did not fit: "DROPPED" line)
        raise RuntimeError(lib.ln_last_error_string())
    kernels = []
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        assert name != "DROPPED"
        kernels.append({"kernel": name, "launches_per_call": round(int(cnt) / reps, 2), "avg_us": round(float(ms) / int(cnt) * 1e3, 2),
                        "us_per_call": round(float(ms) / reps * 1e3, 2)})
    return kernels


def _entry(row, what, kernels, hbm_bytes, reps, dense=None, note=None):
    """`dense`: {launch name: (flop per call, uses bf16x3)} for the launches that are matrix products."""
    us = sum(k["us_per_call"] for k in kernels)
    e = {"row": row, "op": what, "kernels": kernels, "us_per_call": round(us, 2), "calls_timed": reps, "bound": "hbm",
         "algorithmic_bytes": int(hbm_bytes), "achieved": round(hbm_bytes / us / 1e3, 1) if us else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(hbm_bytes / us / 1e3 / HBM_PEAK_GBS, 4) if us else None}
    if dense:
        e["mfma"] = []
        for k in kernels:
            if k["kernel"] not in dense or not k["us_per_call"]:
                continue
            flop, b3 = dense[k["kernel"]]
            tf = flop / k["us_per_call"] / 1e6
            peak = MFMA_BF16_PEAK_TFLOPS if b3 else MFMA_F32_PEAK_TFLOPS
            e["mfma"].append({"kernel": k["kernel"], "flop_per_call": flop, "us_per_call": k["us_per_call"], "fp32_equivalent_tflops": round(tf, 2),
                              "instruction": "v_mfma_f32_16x16x32_bf16, fp32 operands split three ways: 6 bf16 products per fp32 product" if b3 else "v_mfma_f32_16x16x4_f32",
                              "executed_tflops": round((6.0 if b3 else 1.0) * tf, 2), "peak_tflops": peak,
                              "frac_of_instruction_peak": round((6.0 if b3 else 1.0) * tf / peak, 4),
                              "frac_fp32_equivalent_of_f32_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4)})
        # the row's headline figure is the bound the launches sit closer to: a dense launch on the matrix cores is priced against the
        # dense bf16 / fp32 MFMA peak when that fraction exceeds the HBM one (HBM-bound below 64 channels, MFMA-bound from 128)
        if e["mfma"]:
            top = max(e["mfma"], key=lambda r: r["frac_of_instruction_peak"])
            e["hbm"] = {"achieved": e["achieved"], "peak": e["peak"], "unit": e["unit"], "frac": e["frac"]}
            if e["frac"] is None or top["frac_of_instruction_peak"] > e["frac"]:
                e.update(bound="mfma", achieved=top["executed_tflops"], peak=top["peak_tflops"], unit="TFLOP/s", frac=top["frac_of_instruction_peak"])
    if note:
        e["note"] = note
    return e


def run(dev=None, reps: int = 24):
    import lattice_net_amd as L
    from lattice_net_amd import synthetic
    from lattice_net_amd.lattice_funcs import CoarsenLattice, ConvIm2RowLattice, FinefyLattice, GatherLattice, SliceClassifyLattice
    dev = dev or torch.device("cuda", 0)
    lib = L.load_library()
    torch.manual_seed(0)
    pos = torch.from_numpy(synthetic.lidar_cloud(N, 0)).to(dev)
    ops = []

    def fresh():
        return L.Lattice(sigmas=[SIGMA] * D, capacity=CAP, device=dev)

    # ---- a4 distribute (LatticeGPU.cuh:534-650, Lattice.cu:351-410): build + [N(d+1), d + V + 1] rows, V = 1 (values_mode none)
    lat = fresh()
    vals1 = torch.zeros((N, 1), device=dev)
    state = {}

    def distribute():
        lat.begin_splat()
        state["dl"], state["rows"], state["idx"], state["w"] = lat.distribute(pos, vals1)
    k = _profile(lib, distribute, reps)
    dl, idx, w = state["dl"], state["idx"], state["w"]
    m = dl.nr_lattice_vertices()
    ops.append(_entry("a4", "distribute: hash build + distributed rows [4N, d+V+1], V = 1", k,
                      N * (4 * D + 4 * 1 + 8 * (D + 1)) + m * 4 * D + N * (D + 1) * (D + 1 + 1) * 4, reps,
                      note=f"{m} vertices; algorithmic bytes = splat-forward reads / writes without the value rows + the distributed rows"))

    # ---- a4 tail / f1: the glue of DistributeLatticeModule and PointNetModule (lattice_modules.py:72-94, 688-712) as fused launches
    from lattice_net_amd.lattice_modules import DistributeLatticeModule, PointNetReduceFunction
    dmod = DistributeLatticeModule()

    def distribute_module():
        state["dm"] = dmod(fresh(), pos, vals1)
    k = _profile(lib, distribute_module, reps)
    tail = [x for x in k if x["kernel"] in ("k_distribute_centre", "k_csr_reduce_segments", "k_csr_group_sizes", "ln_k_zero_words")]
    T, wd = N * (D + 1), D + 1 + 1
    ops.append(_entry("a4", "DistributeLatticeModule tail: per-vertex mean position, centred rows, vertex-0 rule (mods:72-94)", tail,
                      T * (2 * wd * 4 + 4) + m * (4 * D + 4), reps,
                      note="algorithmic bytes = distributed rows read + centred rows written + indices + per-vertex sums and degrees; the "
                           "launches of the hash build itself are the row above"))
    dlm, drows, didx, _ = state["dm"]
    feat = torch.randn((T, 32), device=dev, requires_grad=True)
    gred = torch.randn((dlm.nr_lattice_vertices(), 64), device=dev)

    def pn_fwd():
        state["pn"] = PointNetReduceFunction.apply(feat, drows, dlm, didx)
    k = _profile(lib, pn_fwd, reps)
    mm = dlm.nr_lattice_vertices()
    ops.append(_entry("f1", "PointNet vertex reduction forward: max + winners' barycentric weights + <4-points / vertex-0 rules, C = 32 (mods:688-712)",
                      k, T * (32 * 4 + 4) + mm * (64 * 4 + 32 * 4 + 4), reps,
                      note="algorithmic bytes = token features + indices read, [M, 2C] rows + [M, C] winners written"))

    def pn_bwd():
        feat.grad = None
        PointNetReduceFunction.apply(feat, drows, dlm, didx).backward(gred)
    k = _profile(lib, pn_bwd, reps)
    kb = [x for x in k if x["kernel"] == "k_pointnet_reduce_backward"]
    ops.append(_entry("f1", "PointNet vertex reduction backward (token-major, every element written once)", kb,
                      T * (32 * 4 + 4) + mm * (32 * 4 + 32 * 4), reps))

    # ---- a9 coarse vertex set (Lattice.cu:706-740): keys only from positions / (2 sigma)
    def coarse():
        state["c1"] = dl.create_coarse_verts_naive(pos)
    k = _profile(lib, coarse, reps)
    c1 = state["c1"]
    m1 = c1.nr_lattice_vertices()
    ops.append(_entry("a9", "create_coarse_verts_naive: level-2 vertex set from the positions", k, N * 4 * D + m1 * 4 * D, reps,
                      note=f"{m1} coarse vertices; algorithmic bytes = positions read + coarse keys written"))

    # ---- a10 level-crossing convolutions (LatticeGPU.cuh:1488-1564 via lattice_funcs.py:323-462), widths of the SemanticKITTI net
    def conv_case(row, what, fn, mq, mn, v, f, bwd):
        kk = _profile(lib, fn, reps)
        # which matrix instruction a launch issues (ln_conv.hip): the convolutions take the bf16x3 form when the gathered width is a
        # multiple of 32 and the produced one a multiple of 16 (per-slot kernel behind k_conv_split_bank, or the 32 x 32 fast paths);
        # the filter gradient when both widths are multiples of 32 and the lattice has >= 4096 vertices (k_grad_filter_b3), else it
        # is on v_mfma_f32_16x16x4_f32
        b3_fwd = v % 32 == 0 and f % 16 == 0
        b3_vg = f % 32 == 0 and v % 16 == 0
        b3_fg = v % 32 == 0 and f % 32 == 0 and mq >= 4096  # k_grad_filter_b3 (launch name k_grad_filter_mfma)
        fwd_fl, vg_fl, fg_fl = 2.0 * mq * E * v * f, 2.0 * mn * E * v * f, 2.0 * mq * E * v * f
        if not bwd:
            by = mn * 4 * v + mq * 4 * E + 4 * E * v * f + mq * 4 * f
            dense = {"k_conv_mfma": (fwd_fl, b3_fwd)}
        else:  # forward + value gradient (the forward with V <-> F) + filter gradient
            by = (mn * 4 * v + mq * 4 * E + 4 * E * v * f + mq * 4 * f) + (mq * 4 * f + mn * 4 * E + 4 * E * v * f + mn * 4 * v) + \
                 (mn * 4 * v + mq * 4 * f + mq * 4 * E + 4 * E * v * f)
            dense = {"k_conv_mfma": (fwd_fl + vg_fl, b3_fwd and b3_vg), "k_grad_filter_mfma": (fg_fl, b3_fg),
                     "k_conv_backward_fused": (vg_fl + fg_fl, True)}
            if any(x["kernel"] == "k_conv_backward_fused" for x in kk):
                dense["k_conv_mfma"] = (fwd_fl, b3_fwd)
        ops.append(_entry(row, what, kk, by, reps, dense=dense,
                          note=f"query rows {mq}, neighbour rows {mn}, V {v} -> F {f}; SURVEY 8d: HBM-bound below 64 channels, MFMA-bound from 128"))

    def coarsen_case(v, f):
        lv = torch.randn((m, v), device=dev, requires_grad=True)
        fb = (torch.randn((E * v, f), device=dev) * 0.05).requires_grad_(True)
        g = torch.randn((m1, f), device=dev)
        out = {}

        def fwd():
            out["y"], _ = CoarsenLattice.apply(lv, dl, fb, c1)

        def bwd():
            lv.grad = fb.grad = None
            y, _ = CoarsenLattice.apply(lv, dl, fb, c1)
            y.backward(g)
        conv_case("a10", f"CoarsenLattice forward (coarse query x fine table) V {v} -> F {f}", fwd, m1, m, v, f, False)
        conv_case("a10", f"CoarsenLattice forward + backward V {v} -> F {f} (one autograd call)", bwd, m1, m, v, f, True)

    def finefy_case(v, f):
        lv = torch.randn((m1, v), device=dev, requires_grad=True)
        fb = (torch.randn((E * v, f), device=dev) * 0.05).requires_grad_(True)
        out = {}

        def fwd():
            out["y"], _ = FinefyLattice.apply(lv, c1, dl, fb)
        conv_case("a10", f"FinefyLattice forward (fine query x coarse table) V {v} -> F {f}", fwd, m, m1, v, f, False)
    coarsen_case(32, 64)
    finefy_case(128, 64)

    # ---- a13 gather (LatticeGPU.cuh:2886-2929, 3761-3817): V = 8 bottleneck of SliceFastCUDALatticeModule
    v8 = 8
    lv8 = torch.randn((m, v8), device=dev, requires_grad=True)
    gg = torch.randn((N, (D + 1) * (v8 + 1)), device=dev)

    def gather_fwd():
        state["g"] = GatherLattice.apply(lv8, dl, pos, idx, w)
    k = _profile(lib, gather_fwd, reps)
    by = N * (8 * (D + 1) + 4 * (D + 1) * (v8 + 1)) + m * 4 * v8
    ops.append(_entry("a13", "gather forward V = 8 -> [N, 36]", k, by, reps))

    def gather_bwd():
        lv8.grad = None
        GatherLattice.apply(lv8, dl, pos, idx, w).backward(gg)
    k = _profile(lib, gather_bwd, reps)
    kb = [x for x in k if x["kernel"] != "k_gather_forward"]
    ops.append(_entry("a13", "gather backward V = 8", kb, by, reps))

    # ---- a14 slice_classify (LatticeGPU.cuh:3387-3464, 3628-3756): V = 64, C = 20 (and the net's own 96)
    for v in (64, 96):
        c = 20
        lv = torch.randn((m, v), device=dev, requires_grad=True)
        dw = (torch.randn((N, D + 1), device=dev) * 0.01).requires_grad_(True)
        lw = torch.randn((c, v), device=dev, requires_grad=True)
        lb = torch.zeros((c,), device=dev, requires_grad=True)
        g = torch.randn((N, c), device=dev)

        def sc_fwd():
            state["l"] = SliceClassifyLattice.apply(lv, dl, pos, dw, lw, lb, c, idx, w)
        k = _profile(lib, sc_fwd, reps)
        by = N * (12 * (D + 1) + 4 * c) + m * 4 * v
        ops.append(_entry("a14", f"slice_classify forward V = {v}, C = {c}", k, by, reps,
                          note=f"the classifier ({2.0 * N * c * v / 1e9:.2f} GFLOP) runs on the vector ALUs in the reference's serial order (bit-exact logits)"))

        def sc_bwd():
            for t in (lv, dw, lw, lb):
                t.grad = None
            SliceClassifyLattice.apply(lv, dl, pos, dw, lw, lb, c, idx, w).backward(g)
        k = _profile(lib, sc_bwd, reps)
        kb = [x for x in k if x["kernel"] != "k_slice_classify_forward"]
        byb = N * (16 * (D + 1) + 4 * c) + 2 * m * 4 * v
        ops.append(_entry("a14", f"slice_classify backward V = {v}, C = {c}", kb, byb, reps, dense={"k_slice_classify_backward": (4.0 * N * c * v, False)},
                          note="k_slice_classify_backward: both dense products on v_mfma_f32_16x16x4_f32; the lattice-value gradient is the CSR segment reduce"))

    # ---- a7 convolution + both gradients at 64 and 128 channels on the finest lattice (the dense contraction, MFMA-bound per §8d)
    for v in (64, 128):
        lv = torch.randn((m, v), device=dev, requires_grad=True)
        fb = (torch.randn((E * v, v), device=dev) * 0.05).requires_grad_(True)
        g = torch.randn((m, v), device=dev)

        def cfwd():
            state["y"], _ = ConvIm2RowLattice.apply(lv, dl, fb, 1)

        def cbwd():
            lv.grad = fb.grad = None
            y, _ = ConvIm2RowLattice.apply(lv, dl, fb, 1)
            y.backward(g)
        conv_case("a7", f"ConvIm2RowLattice forward V = F = {v}", cfwd, m, m, v, v, False)

        fb_frozen = fb.detach().clone()

        def cinf():  # inference with frozen weights (no gradient required for the filter): its split bank is reused from call to call
            with torch.no_grad():
                state["y"], _ = ConvIm2RowLattice.apply(lv, dl, fb_frozen, 1)
        conv_case("a7", f"ConvIm2RowLattice forward V = F = {v}, frozen filter (split bank reused)", cinf, m, m, v, v, False)
        conv_case("a7", f"ConvIm2RowLattice forward + backward V = F = {v} (one autograd call)", cbwd, m, m, v, v, True)
    return {"scan": {"points": N, "vertices": m, "coarse_vertices": m1, "sigma": SIGMA, "capacity": CAP, "pos_dim": D}, "reps": reps, "ops": ops}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=24)
    a = ap.parse_args()
    print(json.dumps(run(reps=a.reps)))
