#!/usr/bin/env python3
"""Algorithmic floor of the lattice operators inside a whole-network step (bench.py `full_unet_ms.algorithmic_floor_ms`).

`trace(step)` runs `step()` once with the Lattice methods of the hot path wrapped, records (operator, sizes) of every call — forward and
backward — and prices each with SURVEY.md 8(d)'s per-unit byte / flop figures at the chip's peaks: HBM 8 TB/s, and for the
contraction the rate at which the matrix cores can deliver fp32-class products — the dense bf16 peak (2.5 PFLOP/s) divided by the 6
bf16 products the exact 3-way operand split spends per fp32 product = 417 TFLOP/s (the fp32-input MFMA peak, 157 TFLOP/s, is LOWER
than what the bf16x3 kernels already deliver and would not be a floor): floor of an operator = max(bytes / HBM peak, flop / 417 T).  The sum is what the lattice operators of the step would cost if every one of them ran at its roofline; GroupNorm,
the PointNet MLP, the optimizer and the loss are outside 8(a) and not in it."""
from __future__ import annotations

import collections

HBM = 8.0e12
MFMA_F32 = 2500.0e12 / 6.0  # fp32-class products on the bf16 pipe (bf16x3: 6 products each)


def _floor(nbytes, flop=0.0):
    return max(nbytes / HBM, flop / MFMA_F32)


def trace(step, Lattice):
    ops = collections.Counter()
    saved = {}

    def wrap(name, fn):
        saved[name] = getattr(Lattice, name)
        orig = saved[name]

        def inner(self, *a, **k):
            fn(self, *a, **k)
            return orig(self, *a, **k)
        setattr(Lattice, name, inner)

    def rows(lat):
        return int(lat.nr_lattice_vertices())

    def conv_fwd(self, filter_bank, dilation, nb, flip, filter_is_transposed=False):
        nbl = nb if nb is not None else self
        v = nbl.val_dim()
        f = filter_bank.shape[0] // 9 if filter_is_transposed else filter_bank.shape[1]
        ops[("conv", rows(self), int(v), int(f))] += 1

    def conv_bwd(self, grad_out, filter_bank, dilation, query, neighbours, filter_grad_fp32=False):
        q = query if query is not None else self
        nb = neighbours if neighbours is not None else self
        ops[("conv_backward", rows(q), int(nb.val_dim()), int(filter_bank.shape[1]))] += 1

    def distribute(self, positions_raw, values, reset_hashmap=True):
        ops[("distribute", int(positions_raw.shape[0]), int(positions_raw.shape[1]), int(values.shape[1]))] += 1

    def splat(self, positions_raw, values):
        ops[("splat", int(positions_raw.shape[0]), int(positions_raw.shape[1]), int(values.shape[1]))] += 1

    def coarse(self, positions_raw):
        ops[("coarse_verts", int(positions_raw.shape[0]), int(positions_raw.shape[1]), 0)] += 1

    def slice_f(self, positions_raw, idx, w, grad_accumulator=None):
        ops[("slice", int(positions_raw.shape[0]), int(positions_raw.shape[1]), int(self.val_dim()))] += 1

    def slice_b(self, positions_raw, grad, idx, w, *a, **k):
        ops[("slice", int(positions_raw.shape[0]), int(positions_raw.shape[1]), int(grad.shape[1]))] += 1

    def gather_f(self, positions_raw, idx, w):
        ops[("gather", int(positions_raw.shape[0]), int(positions_raw.shape[1]), int(self.val_dim()))] += 1

    def gather_b(self, positions_raw, grad, idx, w, *a, **k):
        ops[("gather", int(positions_raw.shape[0]), int(positions_raw.shape[1]), int(grad.shape[1]) // (int(positions_raw.shape[1]) + 1) - 1)] += 1

    def sc_f(self, positions_raw, delta_weights, lw, lb, nr_classes, idx, w):
        ops[("slice_classify", int(positions_raw.shape[0]), int(self.val_dim()), int(nr_classes))] += 1

    def sc_b(self, grad_logits, positions_raw, initial_values, *a, **k):
        ops[("slice_classify_backward", int(positions_raw.shape[0]), int(initial_values.shape[1]), int(grad_logits.shape[1]))] += 1

    wrap("convolve_im2row_standalone", conv_fwd)
    wrap("convolve_im2row_backward", conv_bwd)
    wrap("distribute", distribute)
    wrap("splat_standalone", splat)
    wrap("create_coarse_verts_naive", coarse)
    wrap("slice_standalone_with_precomputation", slice_f)
    wrap("slice_backwards_standalone_with_precomputation_no_homogeneous", slice_b)
    wrap("gather_standalone_with_precomputation", gather_f)
    wrap("gather_backwards_standalone_with_precomputation", gather_b)
    wrap("slice_classify_with_precomputation", sc_f)
    wrap("slice_classify_backwards_with_precomputation", sc_b)
    try:
        step()
    finally:
        for name, orig in saved.items():
            setattr(Lattice, name, orig)
    return ops


def price(ops, vertices_per_point_level=None):
    """{"floor_ms", "bytes", "flop", "ops"} — M of the point-side operators (distribute / slice / ...) is not known to the call, so their
    vertex-side bytes use `vertices_per_point_level` (the vertex count of the finest lattice) when given."""
    E = 9
    total_s, total_b, total_f = 0.0, 0.0, 0.0
    table = []
    m1 = float(vertices_per_point_level or 0)
    for (op, a, b, c), cnt in sorted(ops.items()):
        nbytes, flop = 0.0, 0.0
        if op in ("conv", "conv_backward"):
            m, v, f = a, b, c
            one = m * (4.0 * v + 4.0 * E + 4.0 * f) + 4.0 * E * v * f
            nbytes, flop = (one, 2.0 * m * E * v * f) if op == "conv" else (2.0 * one, 4.0 * m * E * v * f)
        elif op in ("distribute", "splat"):
            n, d, v = a, b, c
            nbytes = n * (4.0 * d + 4.0 * v + 8.0 * (d + 1)) + m1 * (4.0 * d + 4.0 * v)
            if op == "distribute":
                nbytes += n * (d + 1) * 4.0 * (d + v + 1)
        elif op == "coarse_verts":
            n, d = a, b
            nbytes = n * (4.0 * d + 8.0 * (d + 1))
        elif op in ("slice", "gather"):
            n, d, v = a, b, c
            nbytes = n * (8.0 * (d + 1) + 4.0 * v * (1 if op == "slice" else (d + 1))) + m1 * 4.0 * v
        elif op in ("slice_classify", "slice_classify_backward"):
            n, v, cls = a, b, c
            nbytes = n * (8.0 * 4 + 4.0 * cls) + m1 * 4.0 * v
            flop = 2.0 * n * v * cls
            if op.endswith("backward"):
                nbytes, flop = 2.0 * nbytes, 2.0 * flop
        s = _floor(nbytes, flop) * cnt
        total_s += s
        total_b += nbytes * cnt
        total_f += flop * cnt
        table.append({"op": op, "sizes": [a, b, c], "calls": cnt, "floor_us": round(s * 1e6, 2)})
    return {"floor_ms": round(total_s * 1e3, 4), "bytes": int(total_b), "flop": int(total_f), "ops": table}
