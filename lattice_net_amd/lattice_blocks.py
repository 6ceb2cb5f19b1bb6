"""Norm / activation / residual blocks and the classification head that LNN is assembled from
(reference latticenet_py/lattice/lattice_modules.py:26-43, 424-616, 788-1360).

These are compositions of the lattice operator modules (lattice_modules.py) with plain torch layers; they contain
no kernels of their own.  Class names, constructor arguments and sub-module attribute names follow the reference so
that model definitions and `state_dict` keys carry over (`...conv1.norm.gn.weight`, `...coarse.weight`, ...).
Every pre-activation block is one `_PreActBlock`: optional GroupNorm -> activation -> optional channel dropout ->
the wrapped operator; the named classes only choose the pieces.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .lattice_funcs import GatherLattice, SliceClassifyLattice
from .lattice_modules import CoarsenLatticeModule, ConvLatticeIm2RowModule, FinefyLatticeModule, LinearWN, linear_leaky_relu

__all__ = ["DropoutLattice", "BatchNormLatticeModule", "GroupNormLatticeModule", "Conv1x1", "GnRelu1x1", "GnGelu1x1", "Gn", "ConvAct",
           "GnReluConv", "GnGeluConv", "BnReluConv", "CoarsenAct", "GnCoarsen", "GnReluCoarsen", "GnGeluCoarsen", "FinefyAct", "GnFinefy",
           "GnReluFinefy", "GnGeluFinefy", "ResnetBlock", "BottleneckBlock", "SliceFastCUDALatticeModule", "Conv1x1WN", "Conv1x1WNAct",
           "TwoConv", "ResnetBlock2", "DensenetBlock", "GnReluDepthwiseConv"]


def _require_2d(lv: torch.Tensor):
    if lv.dim() != 2:
        raise ValueError(f"lattice values must be [nr_vertices, val_dim], got {tuple(lv.shape)}")


class DropoutLattice(torch.nn.Module):  # mods:26-43: drops whole channels of the [M, C] value matrix
    def __init__(self, prob: float):
        super().__init__()
        self.dropout = torch.nn.Dropout2d(p=prob)

    def forward(self, lv):
        _require_2d(lv)
        return self.dropout(lv.t()[None, :, :, None])[0, :, :, 0].t()


class BatchNormLatticeModule(torch.nn.Module):  # mods:570-583
    def __init__(self, nr_params: int, affine: bool = True, device="cuda"):
        super().__init__()
        self.bn = torch.nn.BatchNorm1d(num_features=nr_params, momentum=0.1, affine=affine).to(device)

    def forward(self, lattice_values, lattice_py):
        _require_2d(lattice_values)
        if lattice_py is not None and lattice_py.rows_device() is not None:
            # static-rows mode: the value matrix is taller than its lattice and torch's BatchNorm would count the padding rows
            raise ValueError("BatchNormLatticeModule cannot run in static-rows mode (its statistics would include the padded rows); "
                             "use GroupNorm blocks, or run eagerly (set_static_rows(None))")
        lattice_values = self.bn(lattice_values)
        lattice_py.set_values(lattice_values)
        return lattice_values, lattice_py


_GN_WORKSPACES = {}
_GN_PRIVATE = [None]  # accumulator pair of the captured step being warmed up / captured (use_gn_workspace)


def new_gn_workspace(device, nbytes: int = 0):
    """A zero-initialised accumulator pair for GroupNorm launches.  `nbytes` 0: large enough for the widest supported layer."""
    if nbytes <= 0:
        nbytes = int(_lib.load().ln_group_norm_workspace_bytes(1024))
    n = max(nbytes // 8, 1)
    return {"bufs": [torch.zeros((n,), dtype=torch.float64, device=device), torch.zeros((n,), dtype=torch.float64, device=device)],
            "dirty": [0, 0], "cur": 0}


class use_gn_workspace:
    """Context: GroupNorm launches inside it alternate between the two buffers of `entry` instead of the (device, stream) pair.
    A captured step owns such an entry for life: eager GroupNorm launches between its replays (validation passes, other models)
    use the shared pair, so they can neither desynchronise the host's idea of which of the graph's buffers is zero nor free a
    buffer the graph points at."""

    def __init__(self, entry):
        self.entry = entry

    def __enter__(self):
        self.prev, _GN_PRIVATE[0] = _GN_PRIVATE[0], self.entry
        return self.entry

    def __exit__(self, *exc):
        _GN_PRIVATE[0] = self.prev
        return False


def _gn_workspace_pair(device, stream: int, nbytes: int):
    """(this call's accumulators, the next call's, bytes of the latter to zero) for GroupNorm launches on (device, stream): two
    zero-initialised buffers used alternately — every call zeroes what the other buffer's last user dirtied, so no call needs
    a fill launch of its own."""
    entry = _GN_PRIVATE[0]
    if entry is not None:
        if entry["bufs"][0].numel() * 8 < nbytes or entry["bufs"][0].device != device:
            raise _lib.LatticeNetHipError("the captured step's GroupNorm accumulators are too small for this layer / on another device")
    else:
        key = (device, stream)
        entry = _GN_WORKSPACES.get(key)
        if entry is None or entry["bufs"][0].numel() * 8 < nbytes:
            entry = new_gn_workspace(device, nbytes)
            _GN_WORKSPACES[key] = entry
    cur = entry["cur"]
    nxt = cur ^ 1
    entry["cur"] = nxt
    to_zero = entry["dirty"][nxt]
    entry["dirty"][nxt] = 0
    entry["dirty"][cur] = nbytes
    return entry["bufs"][cur], entry["bufs"][nxt], to_zero


def reset_gn_workspaces(device=None):
    """Zero-fills the GroupNorm accumulator pair in use — the private pair of the captured step being built (use_gn_workspace), else
    the pairs of the CURRENT stream — and restarts its alternation.  A captured step calls this first, so that every replay starts
    from the state the capture started from (whether the step holds an even or an odd number of GroupNorm launches)."""
    entries = [_GN_PRIVATE[0]] if _GN_PRIVATE[0] is not None else \
        [e for (dev, stream), e in _GN_WORKSPACES.items() if (device is None or dev == device) and stream == _lib.stream_ptr(dev)]
    for entry in entries:
        entry["bufs"][0].zero_()
        entry["bufs"][1].zero_()
        entry["dirty"] = [0, 0]
        entry["cur"] = 0


def _gn_check(rc: int, what: str, device, stream: int):
    if rc != 0 and _GN_PRIVATE[0] is None:
        _GN_WORKSPACES.pop((device, stream), None)  # the zero invariant of the pair may be broken: start over with fresh buffers
    _lib.check(rc, what)


class GroupNormReluFunction(torch.autograd.Function):
    """GroupNorm (+ optional fused ReLU) over an [M, C] value matrix on the HIP kernels of csrc/ln_norm.hip
    (ln_group_norm_forward / _backward): statistics per group over all vertices x the group's channels."""

    @staticmethod
    def forward(ctx, x, weight, bias, num_groups, eps, relu, rows_dev=None):
        lib = _lib.load()
        x = x.contiguous()
        m, c = x.shape
        y = torch.empty_like(x)
        mean_rstd = torch.empty((2 * num_groups,), dtype=torch.float32, device=x.device)
        scale_shift = torch.empty((2 * c,), dtype=torch.float32, device=x.device)
        stream = _lib.stream_ptr(x.device)
        ws, ws_next, zero_bytes = _gn_workspace_pair(x.device, stream, lib.ln_group_norm_workspace_bytes(c))
        _gn_check(lib.ln_group_norm_forward_rows(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), m, c, num_groups, float(eps), int(relu),
                                                 _lib.ptr(y), _lib.ptr(mean_rstd), _lib.ptr(scale_shift), _lib.ptr(ws), ws.numel() * 8,
                                                 _lib.ptr(ws_next), zero_bytes, _lib.ptr(rows_dev), stream), "ln_group_norm_forward", x.device, stream)
        ctx.rows_dev = rows_dev  # device-side row count of the lattice (static-rows mode), None otherwise
        ctx.save_for_backward(x, weight, mean_rstd, scale_shift)
        ctx.args = (num_groups, bool(relu), bias is not None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        lib = _lib.load()
        x, weight, mean_rstd, scale_shift = ctx.saved_tensors
        num_groups, relu, has_bias = ctx.args
        grad_y = grad_y.contiguous()
        m, c = x.shape
        grad_x = torch.empty_like(x)
        grad_w = torch.empty((c,), dtype=torch.float32, device=x.device) if weight is not None else None
        grad_b = torch.empty((c,), dtype=torch.float32, device=x.device) if has_bias else None
        stream = _lib.stream_ptr(x.device)
        ws, ws_next, zero_bytes = _gn_workspace_pair(x.device, stream, lib.ln_group_norm_workspace_bytes(c))
        _gn_check(lib.ln_group_norm_backward_rows(_lib.ptr(x), _lib.ptr(grad_y), _lib.ptr(weight), _lib.ptr(mean_rstd), _lib.ptr(scale_shift), m, c,
                                                  num_groups, int(relu), _lib.ptr(grad_x), _lib.ptr(grad_w), _lib.ptr(grad_b), _lib.ptr(ws),
                                                  ws.numel() * 8, _lib.ptr(ws_next), zero_bytes, _lib.ptr(ctx.rows_dev), stream),
                  "ln_group_norm_backward", x.device, stream)
        return grad_x, grad_w, grad_b, None, None, None, None


class MaxCentreFunction(torch.autograd.Function):
    """x [N, K, C] -> x - (gamma * max_k x + beta) on the kernels of csrc/ln_centre.hip: the max-centring of the gathered
    simplex rows in the DeformSlice head (mods:525-529).  The gradients of gamma and beta are sums over all N points."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        lib = _lib.load()
        x = x.contiguous()
        n, k, c = x.shape
        out = torch.empty_like(x)
        max_vals = torch.empty((n, c), dtype=torch.float32, device=x.device)
        arg_max = torch.empty((n, c), dtype=torch.uint8, device=x.device)
        gamma, beta = gamma.contiguous(), beta.contiguous()
        _lib.check(lib.ln_max_centre_forward(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), n, k, c, _lib.ptr(out), _lib.ptr(max_vals),
                                             _lib.ptr(arg_max), _lib.stream_ptr(x.device)), "ln_max_centre_forward")
        ctx.save_for_backward(max_vals, arg_max, gamma)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        max_vals, arg_max, gamma = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        n, k, c = grad_out.shape
        grad_x = torch.empty_like(grad_out)
        grad_gb = torch.empty((2, c), dtype=torch.float32, device=grad_out.device)
        ws = torch.empty((lib.ln_max_centre_backward_workspace_bytes(n, k, c),), dtype=torch.uint8, device=grad_out.device)
        _lib.check(lib.ln_max_centre_backward(_lib.ptr(grad_out), _lib.ptr(max_vals), _lib.ptr(arg_max), _lib.ptr(gamma), n, k, c,
                                              _lib.ptr(grad_x), _lib.ptr(grad_gb), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(grad_out.device)),
                   "ln_max_centre_backward")
        return grad_x, grad_gb[0], grad_gb[1]


def max_centre_rows(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """[N, K, C] rows minus (gamma * their maximum over K + beta)."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and 1 <= x.shape[1] <= 8 and 1 <= x.shape[2] <= 64:
        return MaxCentreFunction.apply(x, gamma, beta)
    return x - (gamma * x.max(1, keepdim=True)[0] + beta)


def group_norm_rows(x: torch.Tensor, gn: torch.nn.GroupNorm, relu: bool = False, rows_dev=None) -> torch.Tensor:
    """GroupNorm of an [M, C] matrix with the parameters of `gn` (statistics over rows x group channels).  `rows_dev`: device int
    holding the number of rows that count (Lattice.rows_device(): static-rows mode, where x is taller than its lattice)."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.shape[1] <= 1024 and x.shape[0] > 0:
        return GroupNormReluFunction.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, rows_dev)
    if rows_dev is not None:
        raise ValueError("static-rows mode needs the HIP GroupNorm (float32 CUDA rows, channels % 4 == 0, <= 1024 channels)")
    y = gn(x.t().unsqueeze(0)).squeeze(0).t()  # torch's layout: [1, C, M]
    return torch.relu(y) if relu else y


class GroupNormLatticeModule(torch.nn.Module):  # mods:585-616: 32 groups, or C/2 groups when 32 does not divide C
    def __init__(self, nr_params: int, affine: bool = True, device="cuda"):
        super().__init__()
        nr_groups = 32 if nr_params % 32 == 0 else max(int(nr_params / 2), 1)
        self.gn = torch.nn.GroupNorm(nr_groups, nr_params, affine=affine).to(device)

    def forward(self, lattice_values, lattice_py, do_set_values: bool = True, fuse_relu: bool = False):
        _require_2d(lattice_values)
        lattice_values = group_norm_rows(lattice_values, self.gn, fuse_relu, lattice_py.rows_device() if lattice_py is not None else None)
        if do_set_values:
            lattice_py.set_values(lattice_values)
        return lattice_values, lattice_py


class Conv1x1(torch.nn.Module):  # mods:788-804 (the linear layer is created from the first input)
    def __init__(self, out_channels: int, bias: bool, in_channels: Optional[int] = None, device="cuda"):
        super().__init__()
        self.out_channels, self.use_bias, self.device = out_channels, bias, device
        self.linear = None
        if in_channels is not None:
            self._make(in_channels)

    def _make(self, in_channels):
        self.linear = torch.nn.Linear(in_channels, self.out_channels, bias=self.use_bias).to(self.device)
        torch.nn.init.kaiming_normal_(self.linear.weight, mode="fan_in", nonlinearity="relu")

    def forward(self, lv):
        if self.linear is None:
            self._make(lv.shape[1])
        return self.linear(lv)


_ACTS = {"relu": lambda: torch.nn.ReLU(inplace=False), "gelu": torch.nn.GELU, "leaky": lambda: torch.nn.LeakyReLU(0.2), None: None}


class _PreActBlock(torch.nn.Module):
    """[norm] -> [act] -> [dropout] -> op, or op -> act when `act_after` (the *Act variants)."""

    def _setup(self, in_channels, norm: Optional[str], act: Optional[str], with_dropout: bool = False, act_after: bool = False, device="cuda"):
        if norm == "gn":
            self.norm = GroupNormLatticeModule(in_channels, device=device)
        elif norm == "bn":
            self.bn = BatchNormLatticeModule(in_channels, device=device)
        self._norm_kind = norm
        # attribute names as in the reference: `relu` for pre-activations, `act` for post-activations
        self._act_name = None if act is None else ("act" if act_after else "relu")
        if act is not None:
            setattr(self, self._act_name, _ACTS[act]())
        self._act_after = act_after
        self.with_dropout = with_dropout
        if with_dropout:
            self.drop = DropoutLattice(0.2)

    def _pre(self, lv, ls):
        ls.set_values(lv)
        pre_act = getattr(self, self._act_name) if (self._act_name is not None and not self._act_after) else None
        if self._norm_kind == "gn":
            fuse = isinstance(pre_act, torch.nn.ReLU)  # GroupNorm + ReLU in one pass over the values
            lv, ls = self.norm(lv, ls, fuse_relu=fuse)
            if fuse:
                pre_act = None
        elif self._norm_kind == "bn":
            lv, ls = self.bn(lv, ls)
        if pre_act is not None:
            lv = pre_act(lv)
        if self.with_dropout:
            lv = self.drop(lv)
        ls.set_values(lv)
        return lv, ls

    def _post(self, lv, ls):
        if self._act_name is not None and self._act_after:
            lv = getattr(self, self._act_name)(lv)
        ls.set_values(lv)
        return lv, ls


# ---- 1x1 (per-vertex linear) -------------------------------------------------------------------------------------
class _Pre1x1(_PreActBlock):
    def __init__(self, in_channels, out_channels, bias, norm, act, device="cuda"):
        super().__init__()
        self._setup(in_channels, norm, act, device=device)
        self.linear = torch.nn.Linear(in_channels, out_channels, bias=bias).to(device)
        torch.nn.init.kaiming_normal_(self.linear.weight, mode="fan_in", nonlinearity="relu")

    def forward(self, lv, ls):
        lv, ls = self._pre(lv, ls)
        lv = linear_leaky_relu(lv, self.linear.weight, self.linear.bias, -1.0)  # plain per-vertex linear on the streaming kernels
        ls.set_values(lv)
        return lv, ls


class GnRelu1x1(_Pre1x1):  # mods:806-832
    def __init__(self, in_channels, out_channels, bias, device="cuda"):
        super().__init__(in_channels, out_channels, bias, "gn", "relu", device)


class GnGelu1x1(_Pre1x1):  # mods:834-861 (explicit in_channels instead of lazy creation)
    def __init__(self, in_channels, out_channels, bias, device="cuda"):
        super().__init__(in_channels, out_channels, bias, "gn", "gelu", device)


class Gn(_PreActBlock):  # mods:863-878
    def __init__(self, in_channels, device="cuda"):
        super().__init__()
        self._setup(in_channels, "gn", None, device=device)

    def forward(self, lv, ls):
        return self._pre(lv, ls)


# ---- same-level convolution ----------------------------------------------------------------------------------------
class _PreConv(_PreActBlock):
    def __init__(self, in_channels, out_channels, dilation, bias, with_dropout, norm, act, act_after=False, device="cuda"):
        super().__init__()
        # (the operator is registered before the norm, as in the reference's constructors: state_dict order, mods:939-940)
        self.conv = ConvLatticeIm2RowModule(in_channels=in_channels, out_channels=out_channels, neighbourhood_size=1, dilation=dilation,
                                            bias=bias, device=device)
        self._setup(in_channels, norm, act, with_dropout, act_after, device)

    def forward(self, lv, ls):
        lv, ls = self._pre(lv, ls)
        lv_1, ls_1 = self.conv(lv, ls)
        return self._post(lv_1, ls_1)


class ConvAct(_PreConv):  # mods:908-933: [dropout] -> conv -> LeakyReLU(0.2)
    def __init__(self, in_channels, out_channels, dilation, bias, with_dropout, device="cuda"):
        super().__init__(in_channels, out_channels, dilation, bias, with_dropout, None, "leaky", True, device)


class GnReluConv(_PreConv):  # mods:935-960
    def __init__(self, in_channels, out_channels, dilation, bias, with_dropout, device="cuda"):
        super().__init__(in_channels, out_channels, dilation, bias, with_dropout, "gn", "relu", False, device)


class GnReluDepthwiseConv(torch.nn.Module):  # mods:880-905
    """Present for name compatibility only: the reference's block builds on `DepthwiseConvLatticeModule`, which the
    reference never defines (every use, mods:1233-1310, is commented out) — constructing it there raises NameError."""

    def __init__(self, nr_filters, dilation, bias, with_dropout):
        super().__init__()
        raise NotImplementedError("GnReluDepthwiseConv needs DepthwiseConvLatticeModule, which the reference does not define "
                                  "(lattice_modules.py:884); use GnReluConv")


class GnGeluConv(_PreConv):  # mods:962-986
    def __init__(self, in_channels, out_channels, dilation, bias, with_dropout, device="cuda"):
        super().__init__(in_channels, out_channels, dilation, bias, with_dropout, "gn", "gelu", False, device)


class BnReluConv(_PreConv):  # mods:988-1009
    def __init__(self, in_channels, out_channels, dilation, bias, device="cuda"):
        super().__init__(in_channels, out_channels, dilation, bias, False, "bn", "relu", False, device)


# ---- coarsen ---------------------------------------------------------------------------------------------------
class _PreCoarsen(_PreActBlock):
    def __init__(self, in_channels, out_channels, norm, act, act_after=False, device="cuda"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.coarse = CoarsenLatticeModule(in_channels=in_channels, out_channels=out_channels, device=device)
        self._setup(in_channels, norm, act, False, act_after, device)

    def forward(self, lv, ls, concat_connection=None):
        lv, ls = self._pre(lv, ls)
        lv_1, ls_1 = self.coarse(lv, ls)
        lv_1, ls_1 = self._post(lv_1, ls_1)
        if concat_connection is not None:
            lv_1 = torch.cat((lv_1, concat_connection), 1)
            ls_1.set_values(lv_1)
        return lv_1, ls_1


class CoarsenAct(_PreCoarsen):  # mods:1011-1041
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, None, "leaky", True, device)


class GnCoarsen(_PreCoarsen):  # mods:1043-1066
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, "gn", None, False, device)


class GnReluCoarsen(_PreCoarsen):  # mods:1068-1095
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, "gn", "relu", False, device)


class GnGeluCoarsen(_PreCoarsen):  # mods:1097-1122
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, "gn", "gelu", False, device)


# ---- finefy ----------------------------------------------------------------------------------------------------
class _PreFinefy(_PreActBlock):
    def __init__(self, in_channels, out_channels, norm, act, act_after=False, device="cuda"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.fine = FinefyLatticeModule(in_channels=in_channels, out_channels=out_channels, device=device)
        self._setup(in_channels, norm, act, False, act_after, device)

    def forward(self, lv_coarse, ls_coarse, ls_fine):
        lv_coarse, ls_coarse = self._pre(lv_coarse, ls_coarse)
        lv_1, ls_1 = self.fine(lv_coarse, ls_coarse, ls_fine)
        return self._post(lv_1, ls_1)


class FinefyAct(_PreFinefy):  # mods:1124-1150
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, None, "leaky", True, device)


class GnFinefy(_PreFinefy):  # mods:1198-1219
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, "gn", None, False, device)


class GnReluFinefy(_PreFinefy):  # mods:1152-1174
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, "gn", "relu", False, device)


class GnGeluFinefy(_PreFinefy):  # mods:1176-1196
    def __init__(self, in_channels, out_channels, device="cuda"):
        super().__init__(in_channels, out_channels, "gn", "gelu", False, device)


# ---- residual blocks -----------------------------------------------------------------------------------------------
class ResnetBlock(torch.nn.Module):  # mods:1255-1289: two pre-activation convolutions + identity
    def __init__(self, in_channels, out_channels, dilations, biases, with_dropout, device="cuda"):
        super().__init__()
        self.conv1 = GnReluConv(in_channels, out_channels, dilations[0], biases[0], with_dropout=False, device=device)
        self.conv2 = GnReluConv(in_channels, out_channels, dilations[1], biases[1], with_dropout=with_dropout, device=device)

    def forward(self, lv, ls):
        identity = lv
        ls.set_values(lv)
        lv, ls = self.conv1(lv, ls)
        lv, ls = self.conv2(lv, ls)
        lv = lv + identity
        ls.set_values(lv)
        return lv, ls


class BottleneckBlock(torch.nn.Module):  # mods:1336-1361: 1x1 contract (C/4) -> conv -> 1x1 expand + identity
    def __init__(self, in_channels, out_channels, biases, device="cuda"):
        super().__init__()
        self.downsample = 4
        mid = int(out_channels / self.downsample)
        self.contract = GnRelu1x1(in_channels=in_channels, out_channels=mid, bias=biases[0], device=device)
        self.conv = GnReluConv(in_channels=mid, out_channels=mid, dilation=1, bias=biases[1], with_dropout=False, device=device)
        self.expand = GnRelu1x1(in_channels=mid, out_channels=out_channels, bias=biases[2], device=device)

    def forward(self, lv, ls):
        ls.set_values(lv)
        identity = lv
        lv, ls = self.contract(lv, ls)
        lv, ls = self.conv(lv, ls)
        lv, ls = self.expand(lv, ls)
        lv = lv + identity
        ls.set_values(lv)
        return lv, ls


class Conv1x1WN(torch.nn.Module):  # mods:736-759: weight-normalised per-vertex linear layer
    def __init__(self, in_channels, out_channels, bias, device="cuda"):
        super().__init__()
        self.linear = LinearWN(in_channels, out_channels, bias=bias, device=device)

    def forward(self, lv, ls):
        ls.set_values(lv)
        lv = self.linear(lv)
        ls.set_values(lv)
        return lv, ls


class Conv1x1WNAct(Conv1x1WN):  # mods:761-786: ... followed by LeakyReLU(0.2)
    def __init__(self, in_channels, out_channels, bias, device="cuda"):
        super().__init__(in_channels, out_channels, bias, device)
        self.act = torch.nn.LeakyReLU(0.2)

    def forward(self, lv, ls):
        lv, ls = super().forward(lv, ls)
        lv = self.act(lv)
        ls.set_values(lv)
        return lv, ls


class TwoConv(torch.nn.Module):  # mods:1221-1253
    def __init__(self, in_channels, out_channels, dilations, biases, with_dropout, device="cuda"):
        super().__init__()
        self.conv1 = ConvAct(in_channels, out_channels, dilations[0], biases[0], with_dropout=False, device=device)
        self.conv2 = ConvAct(in_channels, out_channels, dilations[1], biases[1], with_dropout=with_dropout, device=device)

    def forward(self, lv, ls):
        ls.set_values(lv)
        lv, ls = self.conv1(lv, ls)
        lv, ls = self.conv2(lv, ls)
        ls.set_values(lv)
        return lv, ls


class ResnetBlock2(torch.nn.Module):  # mods:1291-1334: conv -> layer norm -> conv -> LeakyReLU, + identity
    def __init__(self, in_channels, out_channels, dilations, biases, with_dropout, device="cuda"):
        super().__init__()
        self.conv1 = ConvLatticeIm2RowModule(in_channels=in_channels, out_channels=out_channels, neighbourhood_size=1, dilation=dilations[0],
                                             bias=biases[0], device=device)
        self.norm = torch.nn.GroupNorm(1, out_channels).to(device)
        self.conv2 = ConvLatticeIm2RowModule(in_channels=out_channels, out_channels=out_channels, neighbourhood_size=1, dilation=dilations[1],
                                             bias=biases[1], device=device)
        self.act = torch.nn.LeakyReLU(0.2)

    def forward(self, lv, ls):
        identity = lv
        ls.set_values(lv)
        lv, ls = self.conv1(lv, ls)
        lv = self.norm(lv)  # as the reference: GroupNorm(1, C) applied to the [M, C] matrix (statistics per vertex row)
        ls.set_values(lv)
        lv, ls = self.conv2(lv, ls)
        lv = self.act(lv) + identity
        ls.set_values(lv)
        return lv, ls


class DensenetBlock(torch.nn.Module):  # mods:1363-1395: every layer sees the concatenation of everything before it
    def __init__(self, nr_filters, dilation_list, nr_layers, in_channels=None, device="cuda"):
        super().__init__()
        self.nr_filters = nr_filters
        width = in_channels if in_channels is not None else nr_filters
        self.layers = torch.nn.ModuleList([])
        for i in range(nr_layers):
            self.layers.append(GnReluConv(width, nr_filters, dilation_list[i], False, False, device=device))
            width += nr_filters

    def forward(self, lv, ls):
        ls.set_values(lv)
        stack, outputs = lv, []
        for layer in self.layers:
            lv_new, ls = layer(stack, ls)
            stack = torch.cat((stack, lv_new), 1)
            outputs.append(lv_new)
        out = torch.cat(outputs, 1)
        ls.set_values(out)
        return out, ls


# ---- classification head ---------------------------------------------------------------------------------------------
class SliceFastCUDALatticeModule(torch.nn.Module):
    """DeformSlice head (mods:424-567): two 1x1 step-downs (C, C/2) and an 8-channel bottleneck on the lattice, gather
    of the d+1 vertex rows per point, per-vertex barycentric offsets predicted from the max-centred gathered
    features, then the fused slice + linear classifier (`SliceClassifyLattice`).

    The reference creates `linear_deltaW`, `gamma`, `beta` and `linear_clasify` inside the first forward; their
    shapes only depend on constructor arguments (bottleneck 8 -> 9 gathered values per vertex), so they are created
    here, under the same names — `load_state_dict` of a reference checkpoint then works before any forward."""

    def __init__(self, in_channels, nr_classes, dropout_prob, experiment, device="cuda"):
        super().__init__()
        self.in_channels, self.nr_classes = in_channels, nr_classes
        self.bottleneck_size = 8
        self.dropout_prob = dropout_prob
        self.experiment = experiment
        self.tanh = torch.nn.Tanh()
        if dropout_prob > 0.0:
            self.dropout = DropoutLattice(dropout_prob)
        self.stepdown = torch.nn.ModuleList([])
        cur = in_channels
        for i in range(2):
            nr_out = int(in_channels / (2 ** i))
            if nr_out < self.bottleneck_size:
                raise ValueError(f"{in_channels} input channels are too few for two step-downs above a bottleneck of {self.bottleneck_size}")
            self.stepdown.append(GnRelu1x1(cur, nr_out, False, device=device))
            cur = nr_out
        self.bottleneck = GnRelu1x1(cur, self.bottleneck_size, False, device=device)
        per_vertex = self.bottleneck_size + 1  # gather appends the barycentric weight to every vertex row (LG:2901-2925)
        self.linear_deltaW = torch.nn.Linear(per_vertex, 1, bias=True).to(device)
        with torch.no_grad():
            torch.nn.init.kaiming_uniform_(self.linear_deltaW.weight, mode="fan_in", nonlinearity="tanh")
            self.linear_deltaW.weight *= 0.1  # start close to "no deformation"
            torch.nn.init.zeros_(self.linear_deltaW.bias)
        self.gamma = torch.nn.Parameter(torch.ones(per_vertex, device=device))
        self.beta = torch.nn.Parameter(torch.zeros(per_vertex, device=device))
        self.linear_clasify = torch.nn.Linear(in_channels, nr_classes, bias=True).to(device)
        with torch.no_grad():  # utils.leaky_relu_init(m, 1.0) (utils.py:381-462): U(-b, b), b = sqrt(3)*sqrt(2/(n_in+n_out)); zero bias
            torch.nn.init.xavier_uniform_(self.linear_clasify.weight, gain=1.0)
            torch.nn.init.zeros_(self.linear_clasify.bias)

    def forward(self, lv, ls, positions, splatting_indices, splatting_weights):
        ls.set_values(lv)
        assert self.in_channels == ls.val_dim(), f"in_channels {self.in_channels} != lattice val_dim {ls.val_dim()}"
        nr_positions = positions.shape[0]
        lv_b, ls_b = lv, ls
        for step in self.stepdown:
            lv_b, ls_b = step(lv_b, ls_b)
        lv_b, ls_b = self.bottleneck(lv_b, ls_b)
        gathered = GatherLattice.apply(lv_b, ls_b, positions, splatting_indices, splatting_weights)
        nr_vertices_per_simplex = ls.pos_dim() + 1
        per_vertex = gathered.shape[1] // nr_vertices_per_simplex
        gathered = gathered.view(nr_positions, nr_vertices_per_simplex, per_vertex)
        gathered = max_centre_rows(gathered, self.gamma, self.beta)  # mods:525-529: minus (gamma * max over the simplex + beta)
        # [N(d+1), 9] -> 1: a BLAS GEMM with K = 9, N = 1 over 480 k rows; the streaming linear kernels instead
        delta_weights = linear_leaky_relu(gathered.reshape(nr_positions * nr_vertices_per_simplex, per_vertex), self.linear_deltaW.weight,
                                          self.linear_deltaW.bias, -1.0).reshape(nr_positions, nr_vertices_per_simplex)
        if self.experiment == "slice_no_deform":
            delta_weights = delta_weights * 0
        if self.dropout_prob > 0.0:
            lv = self.dropout(lv)
        ls.set_values(lv)
        return SliceClassifyLattice.apply(lv, ls, positions, delta_weights, self.linear_clasify.weight, self.linear_clasify.bias,
                                          self.nr_classes, splatting_indices, splatting_weights)
