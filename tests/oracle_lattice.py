"""CPU `Lattice` for whole-network checks: the method set of `latticenet.Lattice` (src/PyBridge.cxx:41-113) that the LNN path
uses, backed by the NumPy oracle (oracle/lattice_oracle.py) for every integer decision — simplex location, hash table,
vertex numbering, neighbour traversal — and by plain torch gathers / matmuls in the dtype of the values (float64 in the tests)
for the arithmetic, written from the reference's formulas:

  convolution            rowified @ filter_bank                                   (Lattice.cu:454-462, lattice_modules.py:239-242)
  its backward           grad_filter = rowified^T @ grad;  grad_values = im2row(grad, flipped neighbours) @ filter_bank_backwards
                                                                                    (lattice_funcs.py:298-313, 375-387, 440-452)
  gather / slice-classify and their backwards                                      (LatticeGPU.cuh:2901-2925, 3405-3460, 3648-3814)

Test infrastructure only: nothing under lattice_net_amd/ imports it.  The `torch.autograd.Function`s of
lattice_net_amd.lattice_funcs and the modules of lattice_net_amd.lattice_modules / lattice_blocks / models run unchanged on
an `OracleLattice` with CPU tensors (their HIP-only fast paths are taken for CUDA float32 rows only), so the same network
definition is evaluated twice: on the GPU through the C ABI and here."""
from __future__ import annotations

import numpy as np
import torch

from oracle import lattice_oracle as O


def _np32(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


class OracleLattice:
    def __init__(self, sigmas, capacity: int, lvl: int = 1):
        self.m_sigmas = [float(s) for s in sigmas]
        self.cap = int(capacity)
        self.m_lvl = int(lvl)
        self.table = None
        self._values = None
        self.m_positions = None
        self._nbr_cache = {}
        self._val_dim_hint = 0
        self.m_name = ""
        from lattice_net_amd.lattice import Lattice
        Lattice.m_expected_position_dimensions = len(self.m_sigmas)  # the static the modules size their filter banks with (Lattice.cu:44,135-161)

    # ---- bookkeeping (PyBridge.cxx:86-113)
    def pos_dim(self):
        return len(self.m_sigmas)

    def val_dim(self):
        return int(self._values.shape[1]) if self._values is not None else self._val_dim_hint

    def lvl(self):
        return self.m_lvl

    def capacity(self):
        return self.cap

    def name(self):
        return self.m_name

    def positions(self):
        return self.m_positions

    def nr_lattice_vertices(self):
        return int(self.table.nr_filled)

    def values(self):
        return self._values

    def set_values(self, v):
        self._values = v

    def rows_device(self):
        return None

    def get_filter_extent(self, neighborhood_size: int) -> int:  # Lattice.cu:1353-1358
        assert neighborhood_size == 1
        return 2 * (self.pos_dim() + 1) + 1

    def _clone(self):
        new = OracleLattice(self.m_sigmas, self.cap, self.m_lvl)
        new.table, new._values, new.m_positions = self.table, self._values, self.m_positions
        new._nbr_cache = self._nbr_cache  # same structure: same lists
        return new

    def clone_lattice(self):
        return self._clone()

    def _scaled(self, positions_raw: torch.Tensor) -> np.ndarray:
        return O.scale_positions(_np32(positions_raw), np.asarray(self.m_sigmas, np.float32))

    # ---- build (Lattice.cu:185-193, 351-410, 706-740)
    def begin_splat(self, reset_hashmap: bool = True):
        if reset_hashmap:
            self.table = None

    def distribute(self, positions_raw, values, reset_hashmap: bool = True):
        assert reset_hashmap
        self._val_dim_hint = int(values.shape[1])  # the lattice knows its dimensions from here on (Lattice.cu:362-391)
        new = OracleLattice(self.m_sigmas, self.cap, self.m_lvl)
        new.m_name = "distributed_lattice"
        new.table = O.OracleHashTable(self.cap, self.pos_dim())
        rows, idx, w = O.distribute(new.table, self._scaled(positions_raw), _np32(values))
        new.m_positions = positions_raw
        new._values = torch.zeros((new.nr_lattice_vertices(), values.shape[1]), dtype=values.dtype)
        return new, torch.from_numpy(rows).to(values.dtype), torch.from_numpy(idx), torch.from_numpy(w).to(values.dtype)

    def create_coarse_verts_naive(self, positions_raw):
        new = OracleLattice([2.0 * s for s in self.m_sigmas], self.cap, self.m_lvl + 1)  # Lattice.cu:718-722
        new.m_name = "coarse_lattice"
        new.table = O.OracleHashTable(self.cap, self.pos_dim())
        O.build_splat(new.table, new._scaled(positions_raw), write=False)
        new.m_positions = positions_raw
        new._values = torch.zeros((new.nr_lattice_vertices(), 1), dtype=torch.float64)
        return new

    # ---- vertex-wise aggregations (lattice_modules.py:78, 688-692)
    def _scatter_rows(self, src, idx, w, dst, val_dim, src_div, src_stride):
        t = torch.arange(idx.numel())
        rows = src.reshape(-1, src_stride)[torch.div(t, src_div, rounding_mode="floor"), :val_dim] * w.reshape(-1, 1)
        ok = idx >= 0
        dst.index_add_(0, idx[ok].long(), rows[ok])

    def vertex_point_counts(self, idx):
        return torch.from_numpy(O.vertex_point_counts(idx.numpy(), self.nr_lattice_vertices()))

    def scatter_max(self, src, idx):
        """torch_scatter.scatter_max over the splat indices: per (vertex, channel) maximum and the token attaining it
        (smallest token on ties); vertices without tokens get 0 / -1."""
        m, c = self.nr_lattice_vertices(), src.shape[1]
        s = src.detach().numpy()
        rows = idx.numpy().astype(np.int64)
        ok = rows >= 0
        out = np.full((m, c), -np.inf)
        np.maximum.at(out, rows[ok], s[ok])
        tok = np.arange(rows.shape[0], dtype=np.int64)
        arg = np.full((m, c), np.iinfo(np.int64).max, dtype=np.int64)
        for ch in range(c):
            hit = ok & (s[:, ch] == out[np.maximum(rows, 0), ch])
            np.minimum.at(arg[:, ch], rows[hit], tok[hit])
        empty = ~np.isfinite(out)
        out[empty] = 0.0
        arg[empty] = -1
        return torch.from_numpy(out).to(src.dtype), torch.from_numpy(arg.astype(np.int32))

    # ---- neighbour traversal + convolution
    def _nbr(self, nb: "OracleLattice", dilation: int, flip: bool) -> torch.Tensor:
        key = (id(self.table), id(nb.table), self.m_lvl, nb.m_lvl, int(dilation), bool(flip))
        hit = self._nbr_cache.get(key)
        if hit is None:
            m = self.nr_lattice_vertices()
            hit = torch.from_numpy(O.neighbour_rows(self.table.keys[:m], nb.table, self.m_lvl, nb.m_lvl, int(dilation), bool(flip))).long()
            self._nbr_cache[key] = hit
        return hit

    @staticmethod
    def _im2row(nbr: torch.Tensor, values_nb: torch.Tensor) -> torch.Tensor:
        m, e = nbr.shape
        rows = values_nb[nbr.clamp(min=0)]  # [M, E, V]
        rows = rows * (nbr >= 0).unsqueeze(-1).to(values_nb.dtype)  # absent neighbours contribute zeros (Lattice.cu:634)
        return rows.reshape(m, e * values_nb.shape[1])

    def convolve_im2row_standalone(self, filter_bank, dilation, lattice_neighbours, flip_neighbours):
        nb = lattice_neighbours if lattice_neighbours is not None else self
        assert abs(self.m_lvl - nb.m_lvl) <= 1  # Lattice.cu:439
        assert filter_bank.shape[0] == self.get_filter_extent(1) * nb.val_dim()
        out = self._im2row(self._nbr(nb, dilation, flip_neighbours), nb.values()[: nb.nr_lattice_vertices()]) @ filter_bank
        new = self._clone()
        new._values = out
        return new

    def convolve_im2row_backward(self, grad_out, filter_bank, dilation, query, neighbours):
        q = query if query is not None else self
        nb = neighbours if neighbours is not None else self
        e = q.get_filter_extent(1)
        v, f = nb.val_dim(), filter_bank.shape[1]
        rowified = self._im2row(q._nbr(nb, dilation, False), nb.values()[: nb.nr_lattice_vertices()])
        grad_filter = rowified.t() @ grad_out  # lattice_funcs.py:302
        # lattice_funcs.py:307-313: the errors sit on the query vertices; every neighbour vertex gathers them with flipped
        # neighbours and the bank re-laid out as [E*F, V]
        fb = filter_bank.t().reshape(f, e, v).transpose(0, 1).reshape(e * f, v)
        grad_values = self._im2row(nb._nbr(q, dilation, True), grad_out) @ fb
        return grad_values, grad_filter

    # ---- gather / slice-classify (Lattice.cu:878-917, 982-1039, 1091-1142)
    def _simplex(self, idx, w, n):
        dp1 = self.pos_dim() + 1
        i2 = idx.reshape(n, dp1).long()
        return i2.clamp(min=0), (i2 >= 0), w.reshape(n, dp1)

    def gather_standalone_with_precomputation(self, positions_raw, idx, w):
        n = positions_raw.shape[0]
        vals = self.values()
        rows, ok, w2 = self._simplex(idx, w, n)
        w2 = w2.to(vals.dtype) * ok.to(vals.dtype)
        out = torch.cat((vals[rows] * w2.unsqueeze(-1), w2.unsqueeze(-1)), dim=2)  # [(val * w)(V), w] per simplex vertex
        return out.reshape(n, -1)

    def gather_backwards_standalone_with_precomputation(self, positions_raw, grad, idx, w):
        n = positions_raw.shape[0]
        v = grad.shape[1] // (self.pos_dim() + 1) - 1  # Lattice.cu:1124-1127: the width comes from the gradient, not from the lattice
        rows, ok, w2 = self._simplex(idx, w, n)
        g = grad.reshape(n, self.pos_dim() + 1, v + 1)[:, :, :v] * (w2.to(grad.dtype) * ok.to(grad.dtype)).unsqueeze(-1)
        out = torch.zeros((self.nr_lattice_vertices(), v), dtype=grad.dtype)
        out.index_add_(0, rows.reshape(-1), g.reshape(-1, v))
        self._values = out

    def slice_classify_with_precomputation(self, positions_raw, delta_w, lin_w, lin_b, nr_classes, idx, w):
        n = positions_raw.shape[0]
        vals = self.values()
        rows, ok, w2 = self._simplex(idx, w, n)
        weff = (w2.to(vals.dtype) + delta_w) * ok.to(vals.dtype)
        h = (vals[rows] * weff.unsqueeze(-1)).sum(1)
        return h @ lin_w.t() + lin_b

    def slice_classify_backwards_with_precomputation(self, grad_logits, positions_raw, initial_values, delta_w, lin_w, lin_b, nr_classes,
                                                     g_values, g_delta_w, g_lin_w, g_lin_b, idx, w):
        n = positions_raw.shape[0]
        rows, ok, w2 = self._simplex(idx, w, n)
        okf = ok.to(initial_values.dtype)
        weff = (w2.to(initial_values.dtype) + delta_w) * okf
        gathered = initial_values[rows]                      # [N, d+1, V]
        gh = grad_logits @ lin_w                             # dL/dh [N, V]
        g_values.index_add_(0, rows.reshape(-1), (gh.unsqueeze(1) * weff.unsqueeze(-1)).reshape(-1, gh.shape[1]))
        g_delta_w.copy_((gathered * gh.unsqueeze(1)).sum(-1) * okf)
        g_lin_w.copy_(grad_logits.t() @ (gathered * weff.unsqueeze(-1)).sum(1))
        g_lin_b.copy_(grad_logits.sum(0))
