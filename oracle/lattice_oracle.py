"""CPU oracle for the permutohedral-lattice hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``lattice_net_amd``) never does: it fails loudly when the HIP
extension is missing.

This is a NumPy restatement (strict IEEE fp32, no FMA contraction, no fast-math) of the
arithmetic in the reference's device kernels.  Citations are ``file:line`` relative to
``/root/reference``; LG = include/lattice_net/kernels/LatticeGPU.cuh,
HG = include/lattice_net/kernels/HashTableGPU.cuh, L.cu = src/Lattice.cu.

Parity pinning: every function here is checked in ``tests/test_oracle_golden.py`` against
golden vectors in ``tests/golden/*.npz`` that were produced by running the reference's own
kernel source serially on the host (``oracle/ref_shim`` + ``tests/golden/make_goldens.py``).
The reference ships no tests or known-answer vectors of its own.

Strength of that pin, stated plainly: the reference itself (CUDA + jitify + EasyPBR) cannot be built
or run here, so no output of an actual reference run exists.  The goldens come from the reference's two
kernel headers compiled as serial host C++ behind a qualifier shim (SURVEY.md §8c's recipe); the shim
restates no arithmetic but it IS a stand-in for the CUDA toolchain.  Read strictly, that makes this
oracle "parity unpinned" against a real reference run; what is pinned is every arithmetic source line of
the reference's kernels, executed in IEEE fp32, in serial thread order.

Canonical vertex numbering: rows are numbered by first occurrence in (point, remainder)
order — what a serial run of HG:425-484 produces (the CUDA run numbers by thread arrival
order, HG:454, and is not reproducible).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
MAX_RETRIEVE_CONFLICTS = 300  # HG:494


# --------------------------------------------------------------------------------------
# hashing + open-addressing table (HG:35-54, 425-519)
# --------------------------------------------------------------------------------------
def hash_keys(keys: np.ndarray) -> np.ndarray:
    """HG:35-43: k=0; for i<d: k += key[i]; k *= 2531011 (uint32 wrap)."""
    keys = np.asarray(keys)
    k = np.zeros(keys.shape[:-1], dtype=np.uint64)
    for i in range(keys.shape[-1]):
        k = (k + keys[..., i].astype(np.int64).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
        k = (k * np.uint64(2531011)) & np.uint64(0xFFFFFFFF)
    return k.astype(np.uint32)


class OracleHashTable:
    """Serial emulation of HashTableGPU: keys [CAP,d] row-indexed, entries [CAP] slot->row."""

    def __init__(self, capacity: int, pos_dim: int):
        self.capacity = int(capacity)
        self.pos_dim = int(pos_dim)
        self.clear()

    def clear(self):  # HashTable.cu:49-57
        self.keys = np.zeros((self.capacity, self.pos_dim), dtype=np.int32)
        self.entries = np.full((self.capacity,), -1, dtype=np.int32)
        self.nr_filled = 0
        self._dict = {}

    def insert(self, keys: np.ndarray) -> np.ndarray:
        """Insert keys [K,d] in order; returns the row id of each (HG:425-484 run serially)."""
        keys = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, self.pos_dim)
        K = keys.shape[0]
        rows = np.empty((K,), dtype=np.int32)
        if K == 0:
            return rows
        # distinct keys in first-occurrence order
        uniq, first, inv = np.unique(keys, axis=0, return_index=True, return_inverse=True)
        inv = inv.reshape(-1)
        order = np.argsort(first, kind="stable")
        home = (hash_keys(uniq) % np.uint32(self.capacity)).astype(np.int64)
        uniq_rows = np.empty((uniq.shape[0],), dtype=np.int32)
        entries = self.entries
        cap = self.capacity
        for u in order:
            t = tuple(int(x) for x in uniq[u])
            r = self._dict.get(t)
            if r is None:
                if self.nr_filled >= cap:
                    raise RuntimeError("oracle hash table full (the reference would spin forever, HG:443)")
                h = int(home[u])
                while entries[h] != -1:  # linear probing HG:479-482
                    h += 1
                    if h >= cap:
                        h = 0
                r = self.nr_filled
                entries[h] = r
                self.keys[r] = uniq[u]
                self._dict[t] = r
                self.nr_filled += 1
            uniq_rows[u] = r
        rows[:] = uniq_rows[inv]
        return rows

    def retrieve(self, keys: np.ndarray) -> np.ndarray:
        """HG:491-519: linear probe, stop at empty slot (-1) or after 300 mismatching probes."""
        keys = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, self.pos_dim)
        K = keys.shape[0]
        res = np.full((K,), -1, dtype=np.int32)
        if K == 0:
            return res
        h = (hash_keys(keys) % np.uint32(self.capacity)).astype(np.int64)
        active = np.arange(K)
        for _ in range(MAX_RETRIEVE_CONFLICTS):
            if active.size == 0:
                break
            e = self.entries[h[active]]
            nonempty = e != -1
            match = np.zeros(active.shape, dtype=bool)
            ne = np.nonzero(nonempty)[0]
            if ne.size:
                match[ne] = np.all(self.keys[e[ne]] == keys[active[ne]], axis=1)
            res[active[match]] = e[match]
            keep = nonempty & ~match
            active = active[keep]
            h[active] = (h[active] + 1) % self.capacity
        return res


# --------------------------------------------------------------------------------------
# simplex location (LG:718-806)
# --------------------------------------------------------------------------------------
def scale_factors(pos_dim: int) -> np.ndarray:
    """LG:725-729: scaleFactor[i] = 1/sqrt((i+1)(i+2)) * (d+1)*sqrt(2/3), all fp32."""
    inv_std_dev = F32(pos_dim + 1) * np.sqrt(F32(2.0) / F32(3))
    sf = np.empty((pos_dim,), dtype=F32)
    for i in range(pos_dim):
        sf[i] = F32(1.0) / np.sqrt(F32(i + 1) * F32(i + 2)) * inv_std_dev
    return sf


def scale_positions(positions_raw: np.ndarray, sigmas: np.ndarray) -> np.ndarray:
    """L.cu:226: positions = positions_raw / sigmas (fp32 elementwise)."""
    return (np.asarray(positions_raw, dtype=F32) / np.asarray(sigmas, dtype=F32)).astype(F32)


def elevate(pos: np.ndarray) -> np.ndarray:
    """LG:731-741 (also LG:435-457)."""
    pos = np.asarray(pos, dtype=F32)
    n, d = pos.shape
    sf = scale_factors(d)
    elevated = np.zeros((n, d + 1), dtype=F32)
    sm = np.zeros((n,), dtype=F32)
    for i in range(d, 0, -1):
        cf = pos[:, i - 1] * sf[i - 1]
        elevated[:, i] = sm - F32(i) * cf
        sm = sm + cf
    elevated[:, 0] = sm
    return elevated


def simplex(pos: np.ndarray):
    """Returns (rem0 [N,d+1] i32, rank [N,d+1] i32, bary [N,d+2] f32) per LG:744-795."""
    elevated = elevate(pos)
    n, dp1 = elevated.shape
    d = dp1 - 1
    inv = 1.0 / (d + 1)  # double, LG:748
    v = (elevated.astype(np.float64) * inv).astype(F32)
    up = np.ceil(v) * F32(d + 1)
    down = np.floor(v) * F32(d + 1)
    choose_up = (up - elevated) < (elevated - down)
    rem0 = np.where(choose_up, up, down).astype(np.int32)
    s = rem0.sum(axis=1, dtype=np.int64)
    s = (np.sign(s) * (np.abs(s) // (d + 1))).astype(np.int32)  # C integer division

    diff = elevated - rem0.astype(F32)  # fp32 subtraction, LG:765
    rank = np.zeros((n, dp1), dtype=np.int32)
    for i in range(d):
        for j in range(i + 1, dp1):
            lt = diff[:, i] < diff[:, j]
            rank[:, i] += lt
            rank[:, j] += ~lt  # ties go to j, LG:767-770
    rank += s[:, None]
    lo = rank < 0
    hi = rank > d
    rank = np.where(lo, rank + dp1, np.where(hi, rank - dp1, rank)).astype(np.int32)
    rem0 = np.where(lo, rem0 + dp1, np.where(hi, rem0 - dp1, rem0)).astype(np.int32)

    diff = elevated - rem0.astype(F32)  # rem0 after the wrap fix, LG:790
    delta = (diff.astype(np.float64) * inv).astype(F32)
    bary = np.zeros((n, d + 2), dtype=F32)
    ar = np.arange(n)
    for i in range(dp1):
        bary[ar, d - rank[:, i]] += delta[:, i]
        bary[ar, d + 1 - rank[:, i]] -= delta[:, i]
    bary[:, 0] = (bary[:, 0].astype(np.float64) + (1.0 + bary[:, d + 1].astype(np.float64))).astype(F32)  # LG:795
    return rem0, rank, bary


def simplex_keys(rem0: np.ndarray, rank: np.ndarray) -> np.ndarray:
    """LG:798-806: keys [N, d+1 (remainder), d]."""
    n, dp1 = rem0.shape
    d = dp1 - 1
    keys = np.empty((n, dp1, d), dtype=np.int32)
    for r in range(dp1):
        k = rem0[:, :d] + r
        k = k - dp1 * (rank[:, :d] > d - r)
        keys[:, r, :] = k
    return keys


# --------------------------------------------------------------------------------------
# splat / distribute (LG:707-842, 926-973, 534-650)
# --------------------------------------------------------------------------------------
def build_splat(table: OracleHashTable, pos: np.ndarray, write: bool = True):
    """kernel_splat: insert the d+1 simplex vertices of every (sigma-scaled) point.

    Returns (idx [N*(d+1)] i32, w [N*(d+1)] f32), pre-filled with -1 as L.cu:212-215.
    """
    pos = np.asarray(pos, dtype=F32)
    n, d = pos.shape
    rem0, rank, bary = simplex(pos)
    keys = simplex_keys(rem0, rank).reshape(n * (d + 1), d)
    rows = table.insert(keys)
    idx = np.full((n * (d + 1),), -1, dtype=np.int32)
    w = np.full((n * (d + 1),), -1, dtype=F32)
    if write:
        idx[:] = rows
        w[:] = bary[:, : d + 1].reshape(-1)
    return idx, w


def splat_accumulate(values: np.ndarray, vals: np.ndarray, idx: np.ndarray, w: np.ndarray) -> None:
    """splatCacheNaive LG:937-971: values[idx] += val*w, summed in (p, r) order."""
    n, v = vals.shape
    dp1 = idx.shape[0] // n
    contrib = (vals[:, None, :] * w.reshape(n, dp1, 1)).astype(F32).reshape(n * dp1, v)
    ok = idx >= 0
    np.add.at(values, idx[ok], contrib[ok])


def distribute(table: OracleHashTable, pos: np.ndarray, vals: np.ndarray):
    """distribute kernel LG:534-650 -> (distributed [N(d+1), d+V+1], idx, w)."""
    pos = np.asarray(pos, dtype=F32)
    n, d = pos.shape
    v = vals.shape[1]
    idx, w = build_splat(table, pos, True)
    _, _, bary = simplex(pos)
    out = np.zeros((n, d + 1, d + v + 1), dtype=F32)
    out[:, :, :d] = pos[:, None, :]
    out[:, :, d : d + v] = vals[:, None, :]
    out[:, :, d + v] = bary[:, : d + 1]
    return out.reshape(n * (d + 1), d + v + 1), idx, w


# --------------------------------------------------------------------------------------
# neighbour traversal (LG:1479-1684, 1844-1915, 2187-2284)
# --------------------------------------------------------------------------------------
def _round_half_away(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float64)
    return np.trunc(x + np.copysign(0.5, x)).astype(np.int32)


def _all_coords_integer(vec: np.ndarray, thresh: float = 1e-4) -> np.ndarray:
    """LG:460-473: every |frac| <= 1e-4 (fp32 modff)."""
    frac = np.abs(vec - np.trunc(vec)).astype(F32)
    return ~np.any(frac > F32(thresh), axis=-1)


NOT_VISITED = -2  # slot never written by the traversal (im2rowindices leaves the initial 0 there)


def neighbour_rows(query_keys: np.ndarray, table_nb: OracleHashTable, lvl_q: int, lvl_nb: int, dilation: int,
                   flip: bool) -> np.ndarray:
    """Neighbour list [Mq, E] of rows in the neighbour table.

    Entry >=0: row; -1: visited but absent; NOT_VISITED(-2): the traversal never looks at it.
    Slot layout LG:1622-1626,1645-1649,1676: np -> 2a+(flip), nm -> 2a+(1-flip), centre -> E-1.
    """
    qk = np.asarray(query_keys, dtype=np.int32)
    m, d = qk.shape
    dp1 = d + 1
    E = 2 * dp1 + 1
    kf = np.empty((m, dp1), dtype=F32)
    kf[:, :d] = qk.astype(F32)
    ksum = np.zeros((m,), dtype=F32)
    for i in range(d):
        ksum = ksum + kf[:, i]
    kf[:, d] = -ksum
    scale = F32(2.0) ** F32(lvl_q - lvl_nb)  # LG:1488
    kf = (kf * scale).astype(F32)
    all_int = _all_coords_integer(kf) if scale < 1.0 else np.ones((m,), dtype=bool)
    out = np.full((m, E), NOT_VISITED, dtype=np.int32)

    kint = _round_half_away(kf)
    ci = np.nonzero(all_int)[0]
    if ci.size:
        out[ci, E - 1] = table_nb.retrieve(kint[ci, :d])  # centre, LG:1534-1540

    check = np.ones((m,), dtype=bool) if scale >= 1.0 else ~all_int  # LG:1547-1554
    mm = F32(1.0) if scale >= 1.0 else scale
    step = mm * F32(dilation)  # float*int, LG:1574
    big = mm * F32(dilation) * F32(d)
    rows_c = np.nonzero(check)[0]
    for axis in range(dp1):
        npf = kf + step
        nmf = kf - step
        npf[:, axis] = kf[:, axis] - big
        nmf[:, axis] = kf[:, axis] + big
        np_ok = np.ones((m,), dtype=bool)
        nm_ok = np.ones((m,), dtype=bool)
        if dp1 % 2 != 0:  # odd d+1: neighbour itself must be all-integer, LG:1581-1601
            np_ok = _all_coords_integer(npf)
            nm_ok = _all_coords_integer(nmf)
        npi = _round_half_away(npf)
        nmi = _round_half_away(nmf)
        s_np = 2 * axis + (1 if flip else 0)
        s_nm = 2 * axis + (0 if flip else 1)
        sel = rows_c[np_ok[rows_c]]
        if sel.size:
            out[sel, s_np] = table_nb.retrieve(npi[sel, :d])
        sel = rows_c[nm_ok[rows_c]]
        if sel.size:
            out[sel, s_nm] = table_nb.retrieve(nmi[sel, :d])
    return out


def im2rowindices(nbr: np.ndarray, val_dim: int) -> np.ndarray:
    """LG:1690-1920 output [M, E*V] i32: row replicated V times; neighbour slots hold -1 when
    visited-but-absent, the centre is written only when found, untouched slots keep the initial 0
    (L.cu:600)."""
    m, E = nbr.shape
    out = nbr.copy()
    out[out == NOT_VISITED] = 0
    centre = out[:, E - 1]
    centre[nbr[:, E - 1] < 0] = 0
    return np.repeat(out, val_dim, axis=1).astype(np.int32)


def im2row(nbr: np.ndarray, values_nb: np.ndarray) -> np.ndarray:
    """LG:1464-1688: rowified [M, E*V]; absent neighbours stay 0 (L.cu:634)."""
    m, E = nbr.shape
    v = values_nb.shape[1]
    out = np.zeros((m, E, v), dtype=F32)
    ok = nbr >= 0
    out[ok] = values_nb[nbr[ok]]
    return out.reshape(m, E * v)


def row2im(nbr_noflip: np.ndarray, rowified: np.ndarray, val_dim: int) -> np.ndarray:
    """LG:2187-2284. ``nbr_noflip`` is the (un-flipped) neighbour list of the *output* lattice
    against the lattice whose rows index ``rowified``.  out[v] = sum_a rows[np_a][2a+1] +
    rows[nm_a][2a] (+ rows[centre][E-1]), summed in that order."""
    m, E = nbr_noflip.shape
    v = val_dim
    rows = rowified.reshape(-1, E, v)
    out = np.zeros((m, v), dtype=F32)
    dp1 = (E - 1) // 2
    for a in range(dp1):
        npr = nbr_noflip[:, 2 * a]
        ok = npr >= 0
        out[ok] = out[ok] + rows[npr[ok], 2 * a + 1]
        nmr = nbr_noflip[:, 2 * a + 1]
        ok = nmr >= 0
        out[ok] = out[ok] + rows[nmr[ok], 2 * a]
    c = nbr_noflip[:, E - 1]
    ok = c >= 0
    out[ok] = out[ok] + rows[c[ok], E - 1]
    return out


def coarsen_keys(table_fine: OracleHashTable, table_coarse: OracleHashTable) -> None:
    """coarsen kernel LG:2348-2511 run serially over fine rows."""
    d = table_fine.pos_dim
    dp1 = d + 1
    for r in range(table_fine.nr_filled):
        fk = np.empty((dp1,), dtype=np.int64)
        fk[:d] = table_fine.keys[r]
        fk[d] = -fk[:d].sum()
        half = fk.astype(F32) / F32(2.0)
        frac = np.abs(half - np.trunc(half))
        if np.any(frac > F32(0.1)):
            continue
        div = _round_half_away(half).astype(np.int64)
        table_coarse.insert(div[None, :d])
        for axis in range(dp1):
            npk = fk + 1
            nmk = fk - 1
            npk[axis] = fk[axis] - d
            nmk[axis] = fk[axis] + d
            if table_fine.retrieve(npk[None, :d].astype(np.int32))[0] >= 0:
                c = div + 1
                c[axis] = div[axis] - d
                table_coarse.insert(c[None, :d])
            if table_fine.retrieve(nmk[None, :d].astype(np.int32))[0] >= 0:
                c = div - 1
                c[axis] = div[axis] + d
                table_coarse.insert(c[None, :d])


# --------------------------------------------------------------------------------------
# slice / gather and their backward scatters (LG:2552-2750, 2886-2929, 3540-3623, 3761-3817)
# --------------------------------------------------------------------------------------
def slice_with_precomputation(values: np.ndarray, idx: np.ndarray, w: np.ndarray, n: int) -> np.ndarray:
    """LG:2567-2591: out[p] = sum_r values[idx_r]*w_r over idx_r != -1, r ascending."""
    v = values.shape[1]
    dp1 = idx.shape[0] // n
    idx2 = idx.reshape(n, dp1)
    w2 = w.reshape(n, dp1)
    out = np.zeros((n, v), dtype=F32)
    for r in range(dp1):
        ok = idx2[:, r] != -1
        out[ok] = out[ok] + values[idx2[ok, r]] * w2[ok, r : r + 1]
    return out


def slice_no_precomputation(table: OracleHashTable, values: np.ndarray, pos: np.ndarray):
    """LG:2598-2750 -> (sliced, idx, w)."""
    pos = np.asarray(pos, dtype=F32)
    n, d = pos.shape
    rem0, rank, bary = simplex(pos)
    keys = simplex_keys(rem0, rank).reshape(n * (d + 1), d)
    rows = table.retrieve(keys)
    idx = np.full((n * (d + 1),), -1, dtype=np.int32)
    w = np.full((n * (d + 1),), -1, dtype=F32)
    ok = rows >= 0
    idx[ok] = rows[ok]
    w[ok] = bary[:, : d + 1].reshape(-1)[ok]
    # the weighted sum uses barycentric[remainder] directly (LG:2741), same values as w where found
    wb = bary[:, : d + 1].reshape(-1).copy()
    out = slice_with_precomputation(values, idx, wb, n)
    return out, idx, w


def gather_with_precomputation(values: np.ndarray, idx: np.ndarray, w: np.ndarray, n: int) -> np.ndarray:
    """LG:2901-2925: row [ (val*w)(V), w ] per simplex vertex; absent vertices stay 0 (L.cu:899)."""
    v = values.shape[1]
    dp1 = idx.shape[0] // n
    idx2 = idx.reshape(n, dp1)
    w2 = w.reshape(n, dp1)
    out = np.zeros((n, dp1, v + 1), dtype=F32)
    for r in range(dp1):
        ok = idx2[:, r] >= 0
        out[ok, r, :v] = values[idx2[ok, r]] * w2[ok, r : r + 1]
        out[ok, r, v] = w2[ok, r]
    return out.reshape(n, dp1 * (v + 1))


def slice_backwards(grad: np.ndarray, idx: np.ndarray, w: np.ndarray, m: int) -> np.ndarray:
    """LG:3574-3613: values[idx_r] += grad[p]*w_r into a zeroed [M,V] (L.cu:1079)."""
    n, v = grad.shape
    out = np.zeros((m, v), dtype=F32)
    splat_accumulate(out, grad, idx, w)
    return out


def gather_backwards(grad: np.ndarray, idx: np.ndarray, w: np.ndarray, m: int, pos_dim: int) -> np.ndarray:
    """LG:3778-3814: values[idx_r] += grad[p, r*(V+1)+j]*w_r; the weight column's grad is dropped."""
    n = grad.shape[0]
    dp1 = pos_dim + 1
    v = grad.shape[1] // dp1 - 1
    g = grad.reshape(n, dp1, v + 1)[:, :, :v]
    contrib = (g * w.reshape(n, dp1, 1)).astype(F32).reshape(n * dp1, v)
    out = np.zeros((m, v), dtype=F32)
    ok = idx >= 0
    np.add.at(out, idx[ok], contrib[ok])
    return out


# --------------------------------------------------------------------------------------
# fused slice + classify (LG:3405-3460, 3648-3751)
# --------------------------------------------------------------------------------------
def slice_classify(values, delta_w, lin_w, lin_b, idx, w, n):
    v = values.shape[1]
    dp1 = idx.shape[0] // n
    idx2 = idx.reshape(n, dp1)
    w2 = w.reshape(n, dp1)
    h = np.zeros((n, v), dtype=F32)
    for r in range(dp1):
        ok = idx2[:, r] >= 0
        h[ok] = h[ok] + values[idx2[ok, r]] * (w2[ok, r] + delta_w[ok, r])[:, None]
    c = lin_w.shape[0]
    logits = np.zeros((n, c), dtype=F32)
    for ci in range(c):
        acc = np.zeros((n,), dtype=F32)
        for vi in range(v):
            acc = acc + lin_w[ci, vi] * h[:, vi]
        logits[:, ci] = acc + lin_b[ci]
    return logits


def slice_classify_backwards(grad_logits, values, delta_w, lin_w, lin_b, idx, w, n):
    """Returns (g_values [M,V], g_delta_w [N,d+1], g_lin_w [C,V], g_lin_b [C]) per LG:3668-3751."""
    m, v = values.shape
    c = lin_w.shape[0]
    dp1 = idx.shape[0] // n
    idx2 = idx.reshape(n, dp1)
    w2 = w.reshape(n, dp1)
    g64 = grad_logits.astype(np.float64)
    g_values = np.zeros((m, v), dtype=np.float64)
    gw = g64 @ lin_w.astype(np.float64)  # [N,V]
    for r in range(dp1):
        ok = idx2[:, r] >= 0
        np.add.at(g_values, idx2[ok, r], gw[ok] * (w2[ok, r] + delta_w[ok, r]).astype(np.float64)[:, None])
    h = np.zeros((n, v), dtype=np.float64)
    for r in range(dp1):
        ok = idx2[:, r] >= 0
        h[ok] += values[idx2[ok, r]].astype(np.float64) * (w2[ok, r] + delta_w[ok, r]).astype(np.float64)[:, None]
    g_lin_w = g64.T @ h
    g_lin_b = g64.sum(axis=0)
    g_delta = np.zeros((n, dp1), dtype=np.float64)
    vw = values.astype(np.float64) @ lin_w.astype(np.float64).T  # [M,C]
    for r in range(dp1):
        ok = idx2[:, r] >= 0
        g_delta[ok, r] = np.sum(vw[idx2[ok, r]] * g64[ok], axis=1)
    return g_values.astype(F32), g_delta.astype(F32), g_lin_w.astype(F32), g_lin_b.astype(F32)


# --------------------------------------------------------------------------------------
# dense contraction and the fused-op backward re-layout (L.cu:462; lattice_funcs.py:302-313)
# --------------------------------------------------------------------------------------
def conv_forward(nbr: np.ndarray, values_nb: np.ndarray, filter_bank: np.ndarray) -> np.ndarray:
    return (im2row(nbr, values_nb).astype(np.float64) @ filter_bank.astype(np.float64)).astype(F32)


def backward_filter_layout(filter_bank: np.ndarray, val_dim: int) -> np.ndarray:
    """lattice_funcs.py:307-311: [E*V,F] -> W.T.view(F,E,V).transpose(0,1).reshape(E*F, V)."""
    ev, f = filter_bank.shape
    e = ev // val_dim
    return np.ascontiguousarray(filter_bank.T.reshape(f, e, val_dim).transpose(1, 0, 2)).reshape(e * f, val_dim)


# --------------------------------------------------------------------------------------
# "next" row: vertex-wise aggregations of PointNetModule (lattice_modules.py:688-692)
# --------------------------------------------------------------------------------------
def scatter_max(src: np.ndarray, idx: np.ndarray, m: int):
    """torch_scatter.scatter_max(src, idx, dim=0) restated: per vertex and channel the maximum over its tokens and
    the token attaining it (smallest token on ties); vertices without tokens get 0 / -1.  Tokens with idx < 0 are
    ignored (the reference sends them to vertex 0, which PointNetModule zeroes afterwards, mods:711-712)."""
    t, c = src.shape
    out = np.zeros((m, c), dtype=F32)
    arg = np.full((m, c), -1, dtype=np.int32)
    seen = np.zeros((m,), dtype=bool)
    for tok in range(t):
        r = idx[tok]
        if r < 0:
            continue
        if not seen[r]:
            out[r] = src[tok]
            arg[r] = tok
            seen[r] = True
        else:
            better = src[tok] > out[r]
            out[r][better] = src[tok][better]
            arg[r][better] = tok
    return out, arg


def vertex_point_counts(idx: np.ndarray, m: int) -> np.ndarray:
    return np.bincount(idx[idx >= 0], minlength=m).astype(np.int32)
