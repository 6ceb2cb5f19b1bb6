"""Pure-PyTorch CPU fallback of the hot path — TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.

BASELINE.json's north star asks for "a pure-PyTorch scatter_add CPU fallback (written here as the
correctness oracle, since the reference ships CUDA-only) timed on the same box's host cores".
This module is that fallback: the same op chain {hash build + splat -> neighbour list + one lattice
convolution -> slice}, forward + backward, with torch ops only (index_add_/scatter for the
splat and the backward scatters, index gathers for slice / im2row, torch.mm for the contraction,
torch.unique / searchsorted instead of an open-addressing table).

It is validated against oracle/lattice_oracle.py (itself pinned to the reference's golden vectors)
in tests/test_torch_fallback.py.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Citations: LG = include/lattice_net/kernels/LatticeGPU.cuh.
"""
from __future__ import annotations

import math

import torch

_KEY_BITS = 21  # 3 x 21-bit packing for d = 3 keys (host-side sort key only)


def scale_factors(d: int) -> torch.Tensor:
    inv = torch.tensor(float(d + 1), dtype=torch.float32) * torch.sqrt(torch.tensor(2.0, dtype=torch.float32) / torch.tensor(3.0, dtype=torch.float32))
    sf = [torch.tensor(1.0, dtype=torch.float32) / torch.sqrt(torch.tensor(float(i + 1), dtype=torch.float32) * torch.tensor(float(i + 2), dtype=torch.float32)) * inv
          for i in range(d)]
    return torch.stack(sf)


def simplex(pos: torch.Tensor):
    """LG:718-806 vectorised: returns keys [N, d+1, d] int64 and barycentric weights [N, d+1] f32."""
    n, d = pos.shape
    sf = scale_factors(d)
    elevated = torch.zeros((n, d + 1), dtype=torch.float32)
    sm = torch.zeros((n,), dtype=torch.float32)
    for i in range(d, 0, -1):
        cf = pos[:, i - 1] * sf[i - 1]
        elevated[:, i] = sm - float(i) * cf
        sm = sm + cf
    elevated[:, 0] = sm
    inv = 1.0 / (d + 1)
    v = (elevated.double() * inv).float()
    up = torch.ceil(v) * (d + 1)
    down = torch.floor(v) * (d + 1)
    rem0 = torch.where((up - elevated) < (elevated - down), up, down).to(torch.int64)
    s = torch.div(rem0.sum(1), d + 1, rounding_mode="trunc")
    diff = elevated - rem0.float()
    rank = torch.zeros((n, d + 1), dtype=torch.int64)
    for i in range(d):
        for j in range(i + 1, d + 1):
            lt = diff[:, i] < diff[:, j]
            rank[:, i] += lt
            rank[:, j] += ~lt
    rank = rank + s[:, None]
    lo, hi = rank < 0, rank > d
    rank = torch.where(lo, rank + d + 1, torch.where(hi, rank - d - 1, rank))
    rem0 = torch.where(lo, rem0 + d + 1, torch.where(hi, rem0 - d - 1, rem0))
    delta = ((elevated - rem0.float()).double() * inv).float()
    bary = torch.zeros((n, d + 2), dtype=torch.float32)
    for i in range(d + 1):
        bary.scatter_add_(1, (d - rank[:, i]).unsqueeze(1), delta[:, i : i + 1])
        bary.scatter_add_(1, (d + 1 - rank[:, i]).unsqueeze(1), -delta[:, i : i + 1])
    bary[:, 0] = (bary[:, 0].double() + (1.0 + bary[:, d + 1].double())).float()
    r = torch.arange(d + 1).view(1, d + 1, 1)
    keys = rem0[:, None, :d] + r - (d + 1) * (rank[:, None, :d] > (d - r)).to(torch.int64)
    return keys, bary[:, : d + 1].contiguous()


def pack_keys(keys: torch.Tensor) -> torch.Tensor:
    d = keys.shape[-1]
    off = 1 << (_KEY_BITS - 1)
    p = torch.zeros(keys.shape[:-1], dtype=torch.int64)
    for i in range(d):
        p = p | ((keys[..., i] + off) << (i * _KEY_BITS))
    return p


class TorchLattice:
    """Vertex set of one cloud: sorted packed keys + canonical (first-occurrence) row numbering."""

    def __init__(self, pos_raw: torch.Tensor, sigma: float):
        n, d = pos_raw.shape
        self.n, self.d = n, d
        pos = pos_raw / torch.full((d,), sigma, dtype=torch.float32)
        keys, self.w = simplex(pos)
        packed = pack_keys(keys).reshape(-1)
        uniq, inv = torch.unique(packed, sorted=True, return_inverse=True)
        first = torch.full((uniq.numel(),), packed.numel(), dtype=torch.int64).scatter_reduce_(
            0, inv, torch.arange(packed.numel()), reduce="amin")
        order = torch.argsort(first)  # canonical numbering = first occurrence in (point, remainder) order
        row_of_sorted = torch.empty_like(order)
        row_of_sorted[order] = torch.arange(order.numel())
        self.sorted_keys = uniq
        self.row_of_sorted = row_of_sorted
        self.m = uniq.numel()
        self.idx = row_of_sorted[inv]  # [N*(d+1)]
        self.keys = keys.reshape(-1, d)[first[order]]  # [M, d] row-indexed
        self.w = self.w.reshape(-1)

    def lookup(self, keys: torch.Tensor) -> torch.Tensor:
        p = pack_keys(keys)
        pos = torch.searchsorted(self.sorted_keys, p).clamp(max=self.m - 1)
        found = self.sorted_keys[pos] == p
        return torch.where(found, self.row_of_sorted[pos], torch.full_like(pos, -1))

    def neighbours(self, dilation: int = 1) -> torch.Tensor:
        """Same-level neighbour list [M, E] (LG:1574-1580, slot layout LG:1622-1684)."""
        d = self.d
        full = torch.cat([self.keys, -self.keys.sum(1, keepdim=True)], 1)
        cols = []
        for axis in range(d + 1):
            for sign in (+1, -1):
                nk = full + sign * dilation
                nk[:, axis] = full[:, axis] - sign * dilation * d
                cols.append(self.lookup(nk[:, :d]))
        cols.append(torch.arange(self.m))
        return torch.stack(cols, 1)


def splat(lat: TorchLattice, vals: torch.Tensor) -> torch.Tensor:
    """splatCacheNaive (LG:937-971) with index_add_."""
    d1 = lat.d + 1
    contrib = vals.repeat_interleave(d1, dim=0) * lat.w[:, None]
    return torch.zeros((lat.m, vals.shape[1]), dtype=torch.float32).index_add_(0, lat.idx, contrib)


def im2row(values: torch.Tensor, nbr: torch.Tensor) -> torch.Tensor:
    m, e = nbr.shape
    padded = torch.cat([values, torch.zeros((1, values.shape[1]), dtype=values.dtype)], 0)
    safe = torch.where(nbr >= 0, nbr, torch.full_like(nbr, values.shape[0]))
    return padded[safe.reshape(-1)].reshape(m, e * values.shape[1])


def conv(values: torch.Tensor, nbr: torch.Tensor, filter_bank: torch.Tensor) -> torch.Tensor:
    return im2row(values, nbr).mm(filter_bank)  # Lattice.cu:462


def slice_(lat: TorchLattice, values: torch.Tensor) -> torch.Tensor:
    d1 = lat.d + 1
    g = values[lat.idx] * lat.w[:, None]
    return g.reshape(lat.n, d1, values.shape[1]).sum(1)


def hot_path_step(pos_raw: torch.Tensor, vals: torch.Tensor, filter_bank: torch.Tensor, grad_out: torch.Tensor, sigma: float):
    """One pass of {build + splat -> neighbour list + conv -> slice} forward and backward.
    Returns (out, grad_filter, grad_lattice_values)."""
    lat = TorchLattice(pos_raw, sigma)
    lv = splat(lat, vals).requires_grad_(True)  # splat has no backward in the reference (lattice_funcs.py:41-43)
    fb = filter_bank.detach().requires_grad_(True)
    nbr = lat.neighbours(1)
    cv = conv(lv, nbr, fb)
    out = slice_(lat, cv)
    out.backward(grad_out)
    return out.detach(), fb.grad, lv.grad, lat
