#!/usr/bin/env python3
"""Golden-vector generator — runs ONLY in the build container (needs /root/reference).

It drives the reference's own kernel source (LatticeGPU.cuh / HashTableGPU.cuh), compiled
serially for the host by ``oracle/ref_shim`` (``make -C oracle/ref_shim``), on seeded inputs and
writes small ``.npz`` fixtures next to this file.  The fixtures (data only) are committed; the
reference source never is, and nothing on the GPU box reads /root/reference.

    python tests/golden/make_goldens.py          # regenerates tests/golden/*.npz
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_kernels.so")

F32 = np.float32
I32 = np.int32


def _load():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref_shim")])
    return C.CDLL(LIB)


def P(a):  # raw pointer of a contiguous numpy array
    assert a.flags["C_CONTIGUOUS"], "array must be contiguous"
    return C.c_void_p(a.ctypes.data)


class RefTable:
    """Host arrays laid out like HashTable.cu:31-34 and reset like HashTable.cu:49-57."""

    def __init__(self, capacity, d, v):
        self.cap, self.d = capacity, d
        self.keys = np.zeros((capacity, d), I32)
        self.entries = np.full((capacity,), -1, I32)
        self.values = np.zeros((capacity, v), F32)
        self.nr = np.zeros((1,), I32)

    def args(self):
        return (C.c_int(self.cap), P(self.keys), P(self.entries), P(self.values), P(self.nr))

    @property
    def m(self):
        return int(self.nr[0])


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} returned {rc}")


def splat(lib, t, pos, write=True):
    n, d = pos.shape
    idx = np.full((n * (d + 1),), -1, I32)
    w = np.full((n * (d + 1),), -1, F32)
    check(lib.ref_kernel_splat(P(pos), n, d, *t.args(), P(idx), P(w), int(write)), "kernel_splat")
    return idx, w


def accumulate(lib, t, vals, idx, w):
    n, v = vals.shape
    check(lib.ref_splat_accumulate(P(vals), n, t.d, v, *t.args(), P(idx), P(w)), "splatCacheNaive")


def with_values(t, values):
    """A view of table ``t`` whose values pointer is rebound (HashTable::set_values, HashTable.cu:112)."""
    u = RefTable.__new__(RefTable)
    u.cap, u.d, u.keys, u.entries, u.nr = t.cap, t.d, t.keys, t.entries, t.nr
    u.values = np.ascontiguousarray(values, dtype=F32)
    return u


def im2row(lib, tq, tn, v, lvl_q, lvl_n, dilation, flip):
    m = tq.m
    e = 2 * (tq.d + 1) + 1
    out = np.zeros((m, e * v), F32)
    check(lib.ref_im2row(m, tq.d, v, P(out), e, dilation, *tq.args(), *tn.args(), lvl_q, lvl_n, int(flip)), "im2row")
    return out


def im2rowindices(lib, tq, tn, v, lvl_q, lvl_n, dilation, flip):
    m = tq.m
    e = 2 * (tq.d + 1) + 1
    out = np.zeros((m, e * v), I32)  # L.cu:600 zero-initialised
    check(lib.ref_im2rowindices(m, tq.d, v, P(out), e, dilation, *tq.args(), *tn.args(), lvl_q, lvl_n, int(flip)),
          "im2rowindices")
    return out


def row2im(lib, tq, tn, v, rowified, lvl_q, lvl_n, dilation):
    e = 2 * (tq.d + 1) + 1
    out_t = with_values(tq, np.zeros((tq.m, v), F32))  # L.cu:656 zeroed [M,V]
    check(lib.ref_row2im(tq.d, v, P(rowified), e, dilation, *out_t.args(), *tn.args(), lvl_q, lvl_n), "row2im")
    return out_t.values


def scaled(pos_raw, sigma):
    return (pos_raw / np.full((pos_raw.shape[1],), sigma, F32)).astype(F32)


def lidar_cloud(rng, n):
    """SURVEY.md §8d C3 generator (LiDAR-like)."""
    r = 2.0 + 58.0 * rng.random(n) * rng.random(n)
    az = rng.random(n) * 2 * np.pi
    z = -1.7 + 0.3 * rng.standard_normal(n)
    tall = rng.random(n) < 0.2
    z = z + tall * rng.random(n) * 3.0
    r = np.minimum(r, 60.0)
    return np.stack([r * np.cos(az), r * np.sin(az), z], axis=1).astype(F32)


def main():
    lib = _load()
    out = {}

    # ---------------- F1: config-1 cloud, every same-level op ----------------
    rng = np.random.default_rng(0)
    n, d, v, cap, sigma = 1000, 3, 4, 60000, 0.2
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    vals = rng.standard_normal((n, v)).astype(F32)
    pos = scaled(pos_raw, sigma)
    t = RefTable(cap, d, v)
    idx, w = splat(lib, t, pos)
    accumulate(lib, t, vals, idx, w)
    m = t.m
    tv = with_values(t, t.values[:m].copy())
    f1 = dict(pos_raw=pos_raw, sigma=F32(sigma), vals=vals, capacity=I32(cap), nr_filled=I32(m), keys=t.keys[:m].copy(),
              idx=idx, w=w, values=t.values[:m].copy())
    f1["im2rowindices_d1"] = im2rowindices(lib, tv, tv, v, 1, 1, 1, False)
    f1["im2row_d1"] = im2row(lib, tv, tv, v, 1, 1, 1, False)
    f1["im2row_d1_flip"] = im2row(lib, tv, tv, v, 1, 1, 1, True)
    grad_rows = rng.standard_normal((m, 9 * v)).astype(F32)
    f1["grad_rowified"] = grad_rows
    f1["row2im_d1"] = row2im(lib, tv, tv, v, grad_rows, 1, 1, 1)
    sl = np.zeros((n, v), F32)
    check(lib.ref_slice_with_precomputation(P(pos), P(sl), n, d, v, *tv.args(), P(idx), P(w)), "slice")
    f1["slice"] = sl
    # slice at *new* positions without precomputation
    qpos_raw = rng.uniform(-1.1, 1.1, (300, d)).astype(F32)
    qpos = scaled(qpos_raw, sigma)
    sl2 = np.zeros((300, v), F32)
    idx2 = np.full((300 * (d + 1),), -1, I32)
    w2 = np.full((300 * (d + 1),), -1, F32)
    check(lib.ref_slice_no_precomputation(P(qpos), P(sl2), 300, d, v, *tv.args(), P(idx2), P(w2)), "slice_no_pre")
    f1.update(qpos_raw=qpos_raw, slice_nopre=sl2, idx_nopre=idx2, w_nopre=w2)
    ga = np.zeros((n, (d + 1) * (v + 1)), F32)
    check(lib.ref_gather_with_precomputation(P(pos), P(ga), n, d, v, *tv.args(), P(idx), P(w)), "gather")
    f1["gather"] = ga
    g_sl = rng.standard_normal((n, v)).astype(F32)
    tb = with_values(t, np.zeros((m, v), F32))
    check(lib.ref_slice_backwards(P(g_sl), n, d, v, *tb.args(), P(idx), P(w)), "slice_bwd")
    f1.update(grad_sliced=g_sl, slice_bwd=tb.values.copy())
    g_ga = rng.standard_normal((n, (d + 1) * (v + 1))).astype(F32)
    tb = with_values(t, np.zeros((m, v), F32))
    check(lib.ref_gather_backwards(P(g_ga), n, d, v, *tb.args(), P(idx), P(w)), "gather_bwd")
    f1.update(grad_gathered=g_ga, gather_bwd=tb.values.copy())
    out["F1_config1"] = f1

    # ---------------- F2: boundary cases (ties, origin, duplicates, negatives) ----------------
    rng = np.random.default_rng(2)
    base = rng.uniform(-3, 3, (40, 3)).astype(F32)
    special = np.array(
        [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -2, 0], [0, 0, -3], [1, 1, 1], [-1, -1, -1],
         [0.5, 0.5, 0.5], [2, 2, 0], [0, 2, 2], [1e-30, 0, 0], [-1e-30, 1e-30, 0], [1e4, -1e4, 1e4], [123.456, -654.321, 0.001],
         [0.25, 0.25, 0.25], [0.75, -0.75, 0.75], [4, 4, 4], [-4, 4, -4]], dtype=F32)
    # points whose elevated coordinates are exact integers: integer combinations of coarse steps
    grid = np.stack(np.meshgrid(np.arange(-2, 3), np.arange(-2, 3), np.arange(-2, 3), indexing="ij"), -1).reshape(-1, 3)
    grid = (grid.astype(F32) * F32(0.5)).astype(F32)
    pos_raw = np.concatenate([base, special, base[:10], grid, -base[:5]], axis=0).astype(F32)
    pos_raw = np.ascontiguousarray(pos_raw)
    n = pos_raw.shape[0]
    sigma = 1.0
    pos = scaled(pos_raw, sigma)
    t = RefTable(4096, 3, 1)
    idx, w = splat(lib, t, pos)
    out["F2_boundary"] = dict(pos_raw=pos_raw, sigma=F32(sigma), capacity=I32(4096), nr_filled=I32(t.m),
                              keys=t.keys[: t.m].copy(), idx=idx, w=w)

    # ---------------- F3: two levels (naive coarse, key coarsen, cross-level traversals) ----------------
    rng = np.random.default_rng(3)
    n, d, v, cap, sigma = 500, 3, 4, 20000, 0.3
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    vals = rng.standard_normal((n, v)).astype(F32)
    fine = RefTable(cap, d, v)
    idx, w = splat(lib, fine, scaled(pos_raw, sigma))
    accumulate(lib, fine, vals, idx, w)
    mf = fine.m
    fine_v = with_values(fine, fine.values[:mf].copy())
    coarse = RefTable(cap, d, 1)  # create_coarse_verts_naive L.cu:706-740: sigma*2, insert only
    splat(lib, coarse, scaled(pos_raw, 2 * sigma), write=False)
    mc = coarse.m
    coarse_vals = rng.standard_normal((mc, v)).astype(F32)
    coarse_v = with_values(coarse, coarse_vals)
    kc = RefTable(cap, d, 1)  # create_coarse_verts L.cu:670-703 (key based)
    check(lib.ref_coarsen(d, *fine.args(), *kc.args()), "coarsen")
    f3 = dict(pos_raw=pos_raw, sigma=F32(sigma), vals=vals, capacity=I32(cap), fine_nr=I32(mf), fine_keys=fine.keys[:mf].copy(),
              fine_values=fine_v.values.copy(), coarse_nr=I32(mc), coarse_keys=coarse.keys[:mc].copy(),
              coarse_values=coarse_vals, keycoarse_nr=I32(kc.m), keycoarse_keys=kc.keys[: kc.m].copy())
    for flip in (False, True):
        s = "_flip" if flip else ""
        f3["idx_coarse_from_fine" + s] = im2rowindices(lib, coarse_v, fine_v, v, 2, 1, 1, flip)
        f3["row_coarse_from_fine" + s] = im2row(lib, coarse_v, fine_v, v, 2, 1, 1, flip)
        f3["idx_fine_from_coarse" + s] = im2rowindices(lib, fine_v, coarse_v, v, 1, 2, 1, flip)
        f3["row_fine_from_coarse" + s] = im2row(lib, fine_v, coarse_v, v, 1, 2, 1, flip)
    out["F3_two_level"] = f3

    # ---------------- F4: dilation 2 ----------------
    rng = np.random.default_rng(4)
    n, d, v, cap, sigma = 800, 3, 2, 20000, 0.25
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    vals = rng.standard_normal((n, v)).astype(F32)
    t = RefTable(cap, d, v)
    idx, w = splat(lib, t, scaled(pos_raw, sigma))
    accumulate(lib, t, vals, idx, w)
    m = t.m
    tv = with_values(t, t.values[:m].copy())
    grad_rows = rng.standard_normal((m, 9 * v)).astype(F32)
    out["F4_dilation2"] = dict(pos_raw=pos_raw, sigma=F32(sigma), vals=vals, capacity=I32(cap), nr_filled=I32(m),
                               keys=t.keys[:m].copy(), values=tv.values.copy(),
                               im2rowindices_d2=im2rowindices(lib, tv, tv, v, 1, 1, 2, False),
                               im2row_d2=im2row(lib, tv, tv, v, 1, 1, 2, False), grad_rowified=grad_rows,
                               row2im_d2=row2im(lib, tv, tv, v, grad_rows, 1, 1, 2))

    # ---------------- F5: distribute rows ----------------
    rng = np.random.default_rng(5)
    n, d, v, cap, sigma = 400, 3, 2, 20000, 0.3
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    vals = rng.standard_normal((n, v)).astype(F32)
    pos = scaled(pos_raw, sigma)
    t = RefTable(cap, d, v)
    idx = np.full((n * (d + 1),), -1, I32)
    w = np.full((n * (d + 1),), -1, F32)
    dist = np.zeros((n * (d + 1), d + v + 1), F32)
    check(lib.ref_distribute(P(pos), P(vals), n, d, v, *t.args(), P(idx), P(w), P(dist)), "distribute")
    out["F5_distribute"] = dict(pos_raw=pos_raw, sigma=F32(sigma), vals=vals, capacity=I32(cap), nr_filled=I32(t.m),
                                keys=t.keys[: t.m].copy(), idx=idx, w=w, distributed=dist)

    # ---------------- F6: slice_classify fwd/bwd, C=5 ----------------
    rng = np.random.default_rng(6)
    n, d, v, c, cap, sigma = 600, 3, 8, 5, 20000, 0.3
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    pos = scaled(pos_raw, sigma)
    t = RefTable(cap, d, 1)
    idx, w = splat(lib, t, pos)
    m = t.m
    lat_vals = rng.standard_normal((m, v)).astype(F32)
    tv = with_values(t, lat_vals)
    dw = (0.1 * rng.standard_normal((n, d + 1))).astype(F32)
    lw = rng.standard_normal((c, v)).astype(F32)
    lb = rng.standard_normal((c,)).astype(F32)
    logits = np.zeros((n, c), F32)
    check(lib.ref_slice_classify(P(pos), P(logits), P(dw), P(lw), P(lb), n, d, v, c, *tv.args(), P(idx), P(w)), "slice_classify")
    gl = rng.standard_normal((n, c)).astype(F32)
    g_vals = np.zeros((m, v), F32)
    g_dw = np.zeros((n, d + 1), F32)
    g_lw = np.zeros((c, v), F32)
    g_lb = np.zeros((c,), F32)
    check(lib.ref_slice_classify_backwards(P(gl), P(lat_vals), n, d, v, c, P(dw), P(lw), P(lb), P(g_vals), P(g_dw), P(g_lw),
                                           P(g_lb), *tv.args(), P(idx), P(w)), "slice_classify_bwd")
    out["F6_slice_classify"] = dict(pos_raw=pos_raw, sigma=F32(sigma), capacity=I32(cap), nr_filled=I32(m), idx=idx, w=w,
                                    lattice_values=lat_vals, delta_w=dw, lin_w=lw, lin_b=lb, logits=logits, grad_logits=gl,
                                    g_values=g_vals, g_delta_w=g_dw, g_lin_w=g_lw, g_lin_b=g_lb)

    # ---------------- F11: slice_classify at the widths of the SemanticKITTI head (C = 20; V = 64 and 32), ragged tiles ----------------
    # (the shapes the wave-tiled kernels of ln_classify.hip take: 64-point tiles forward, 16-point tiles backward; n = 1111 leaves both ragged)
    rng = np.random.default_rng(11)
    n, d, c, cap, sigma = 1111, 3, 20, 20000, 0.45
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    pos = scaled(pos_raw, sigma)
    t = RefTable(cap, d, 1)
    idx, w = splat(lib, t, pos)
    m = t.m
    f11 = dict(pos_raw=pos_raw, sigma=F32(sigma), capacity=I32(cap), nr_filled=I32(m), idx=idx, w=w)
    f11["delta_w"] = dw = (0.1 * rng.standard_normal((n, d + 1))).astype(F32)
    f11["grad_logits"] = gl = rng.standard_normal((n, c)).astype(F32)
    for v in (64, 32):
        lat_vals = rng.standard_normal((m, v)).astype(F32)
        tv = with_values(t, lat_vals)
        lw = rng.standard_normal((c, v)).astype(F32)
        lb = rng.standard_normal((c,)).astype(F32)
        logits = np.zeros((n, c), F32)
        check(lib.ref_slice_classify(P(pos), P(logits), P(dw), P(lw), P(lb), n, d, v, c, *tv.args(), P(idx), P(w)), "slice_classify F11")
        g_vals, g_dw, g_lw, g_lb = np.zeros((m, v), F32), np.zeros((n, d + 1), F32), np.zeros((c, v), F32), np.zeros((c,), F32)
        gl_arg = gl.copy()  # (kept alive across the call: the kernel takes a non-const pointer)
        check(lib.ref_slice_classify_backwards(P(gl_arg), P(lat_vals), n, d, v, c, P(dw), P(lw), P(lb), P(g_vals), P(g_dw), P(g_lw),
                                               P(g_lb), *tv.args(), P(idx), P(w)), "slice_classify_bwd F11")
        assert np.array_equal(gl_arg, gl)
        f11.update({f"lattice_values_{v}": lat_vals, f"lin_w_{v}": lw, f"lin_b_{v}": lb, f"logits_{v}": logits, f"g_values_{v}": g_vals,
                    f"g_delta_w_{v}": g_dw, f"g_lin_w_{v}": g_lw, f"g_lin_b_{v}": g_lb})
    out["F11_slice_classify_kitti_head"] = f11

    # ---------------- F7: near-full table (probe chains, 300-probe retrieve cap) ----------------
    rng = np.random.default_rng(7)
    n, d, v, sigma = 3000, 3, 1, 0.05
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    pos = scaled(pos_raw, sigma)
    probe = RefTable(200000, d, v)
    splat(lib, probe, pos)
    cap = int(probe.m / 0.97) + 1
    t = RefTable(cap, d, v)
    idx, w = splat(lib, t, pos)
    m = t.m
    vals = rng.standard_normal((m, v)).astype(F32)
    tv = with_values(t, vals)
    sl = np.zeros((n, v), F32)
    idx2 = np.full((n * (d + 1),), -1, I32)
    w2 = np.full((n * (d + 1),), -1, F32)
    check(lib.ref_slice_no_precomputation(P(pos), P(sl), n, d, v, *tv.args(), P(idx2), P(w2)), "slice_no_pre")
    out["F7_near_full"] = dict(pos_raw=pos_raw, sigma=F32(sigma), capacity=I32(cap), nr_filled=I32(m), keys=t.keys[:m].copy(),
                               entries=t.entries.copy(), idx=idx, w=w, lattice_values=vals, slice_nopre=sl, idx_nopre=idx2,
                               w_nopre=w2, im2rowindices_d1=im2rowindices(lib, tv, tv, v, 1, 1, 1, False))

    # ---------------- F8: pos_dim 2 (odd d+1 traversal branch), both levels ----------------
    rng = np.random.default_rng(8)
    n, d, v, cap, sigma = 400, 2, 4, 8000, 0.2
    pos_raw = rng.uniform(-1, 1, (n, d)).astype(F32)
    vals = rng.standard_normal((n, v)).astype(F32)
    fine = RefTable(cap, d, v)
    idx, w = splat(lib, fine, scaled(pos_raw, sigma))
    accumulate(lib, fine, vals, idx, w)
    mf = fine.m
    fine_v = with_values(fine, fine.values[:mf].copy())
    coarse = RefTable(cap, d, 1)
    splat(lib, coarse, scaled(pos_raw, 2 * sigma), write=False)
    mc = coarse.m
    coarse_vals = rng.standard_normal((mc, v)).astype(F32)
    coarse_v = with_values(coarse, coarse_vals)
    out["F8_posdim2"] = dict(pos_raw=pos_raw, sigma=F32(sigma), vals=vals, capacity=I32(cap), fine_nr=I32(mf),
                             fine_keys=fine.keys[:mf].copy(), idx=idx, w=w, fine_values=fine_v.values.copy(),
                             coarse_nr=I32(mc), coarse_keys=coarse.keys[:mc].copy(), coarse_values=coarse_vals,
                             idx_same=im2rowindices(lib, fine_v, fine_v, v, 1, 1, 1, False),
                             row_same=im2row(lib, fine_v, fine_v, v, 1, 1, 1, False),
                             idx_coarse_from_fine=im2rowindices(lib, coarse_v, fine_v, v, 2, 1, 1, False),
                             idx_fine_from_coarse=im2rowindices(lib, fine_v, coarse_v, v, 1, 2, 1, False),
                             row_fine_from_coarse=im2row(lib, fine_v, coarse_v, v, 1, 2, 1, False))

    # ---------------- F9: LiDAR-like cloud, integer outputs only (keeps the file small) ----------------
    rng = np.random.default_rng(9)
    n, d, cap, sigma = 6000, 3, 30000, 0.9
    pos_raw = lidar_cloud(rng, n)
    t = RefTable(cap, d, 1)
    idx, w = splat(lib, t, scaled(pos_raw, sigma))
    tv = with_values(t, np.zeros((t.m, 1), F32))
    out["F9_lidar"] = dict(pos_raw=pos_raw, sigma=F32(sigma), capacity=I32(cap), nr_filled=I32(t.m), keys=t.keys[: t.m].copy(),
                           idx=idx, w=w, im2rowindices_d1=im2rowindices(lib, tv, tv, 1, 1, 1, 1, False))

    for name, arrays in out.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    sys.exit(main())
