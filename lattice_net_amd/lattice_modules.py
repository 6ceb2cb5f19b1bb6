"""Thin nn.Modules around the lattice operators — the five op modules that sit directly on the hot
path (reference latticenet_py/lattice/lattice_modules.py:46-96,174-418), same constructor
arguments, parameter shapes ([E*V, F] filter banks) and initialisation.  Norm / activation / block
wrappers of the reference are plain PyTorch and out of scope here.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from .lattice import Lattice
from .lattice_funcs import (CoarsenLattice, ConvIm2RowLattice, DistributeLattice, ExpandLattice, FinefyLattice, GatherLattice,
                            ScatterMaxLattice, SliceLattice, SplatLattice)

__all__ = ["SplatLatticeModule", "DistributeLatticeModule", "PointNetModule", "ConvLatticeIm2RowModule", "CoarsenLatticeModule",
           "FinefyLatticeModule", "SliceLatticeModule", "GatherLatticeModule", "LinearWN", "ConvLatticeIm2RowWNModule",
           "CoarsenLatticeWNModule", "FinefyLatticeWNModule", "ExpandLatticeModule", "ConvLatticeModule"]


def _kaiming_uniform_fan_out_(weight: torch.Tensor, fan_scale: float = 1.0, std_scale: float = 1.0):
    """lattice_modules.py:202-207 / 277-283: uniform(-b, b), b = sqrt(3) * gain / sqrt(fan_out)."""
    fan = torch.nn.init._calculate_correct_fan(weight, "fan_out") * fan_scale
    gain = torch.nn.init.calculate_gain("relu", 1)
    std = gain / math.sqrt(fan) * std_scale
    bound = math.sqrt(3.0) * std
    with torch.no_grad():
        weight.uniform_(-bound, bound)


def _bias_init_(bias: torch.Tensor, weight: torch.Tensor):
    _, fan_out = torch.nn.init._calculate_fan_in_and_fan_out(weight)
    torch.nn.init.uniform_(bias, -1 / math.sqrt(fan_out), 1 / math.sqrt(fan_out))


def _leaky_relu_init_bound(n_in: int, n_out: int, extent: int = 1, alpha: float = 0.2) -> float:
    """utils.leaky_relu_init (utils.py:381-462): U(-b, b) with b = sqrt(3) * gain * sqrt(2 / ((n_in + n_out) * extent))."""
    gain = math.sqrt(2.0 / (1.0 + alpha ** 2))
    return math.sqrt(3.0) * gain * math.sqrt(2.0 / ((n_in + n_out) * extent))


FUSED_GLUE = {"weight_norm", "distribute", "pointnet"}  # tests remove entries to compare the fused glue kernels (csrc/ln_glue.hip) with the torch chains they replace


class WeightNormFunction(torch.autograd.Function):
    """w = v * g / ||v||_F (utils.py:72-158, v_dim=None) as one launch each way (csrc/ln_glue.hip) instead of the norm / div / mul
    chain and its ~10 backward launches on a parameter of a few thousand numbers."""

    @staticmethod
    def forward(ctx, v, g, g_dim):
        lib = _lib.load()
        v, gc = v.contiguous(), g.contiguous()
        rows, cols = v.shape
        w = torch.empty_like(v)
        norm = torch.empty((1,), dtype=torch.float32, device=v.device)
        _lib.check(lib.ln_weight_norm_forward(_lib.ptr(v), _lib.ptr(gc), rows, cols, g_dim, _lib.ptr(w), _lib.ptr(norm), _lib.stream_ptr(v.device)),
                   "ln_weight_norm_forward")
        ctx.save_for_backward(v, gc, norm)
        ctx.g_dim = g_dim
        return w

    @staticmethod
    def backward(ctx, grad_w):
        lib = _lib.load()
        v, g, norm = ctx.saved_tensors
        grad_w = grad_w.contiguous()
        rows, cols = v.shape
        gv, gg = torch.empty_like(v), torch.empty_like(g)
        _lib.check(lib.ln_weight_norm_backward(_lib.ptr(v), _lib.ptr(g), _lib.ptr(grad_w), _lib.ptr(norm), rows, cols, ctx.g_dim, _lib.ptr(gv),
                                               _lib.ptr(gg), _lib.stream_ptr(v.device)), "ln_weight_norm_backward")
        return gv, gg, None


class _WeightNormed:
    """The reference's weight_norm_wrapper (utils.py:72-158) with v_dim=None: parameters `weight_v` (direction, the
    layer's weight shape) and `weight_g` (one magnitude per output unit, kept along `g_dim`), effective weight
    = weight_v * weight_g / ||weight_v||_F.  Same parameter names as torch's WeightNorm, so reference checkpoints
    (`...weight_g`, `...weight_v`) load unchanged."""

    def _install_weight_norm(self, v: torch.Tensor, g_dim: int):
        g_shape = [1] * v.dim()
        g_shape[g_dim] = v.shape[g_dim]
        # registration order of torch's WeightNorm.apply: <name>_g, then <name>_v, both behind the parameters that stay (bias)
        self.weight_g = torch.nn.Parameter(torch.full(g_shape, float(v.norm()), dtype=v.dtype, device=v.device))  # unfuse(): g := ||v||
        self.weight_v = torch.nn.Parameter(v)
        self._g_dim = g_dim

    @property
    def weight(self) -> torch.Tensor:
        v, g = self.weight_v, self.weight_g
        # (the kernels read g[j] for j < v.shape[g_dim] and write grad_g with the same extent: a magnitude vector of any other length —
        # a reshaped or hand-loaded parameter, a scalar g — takes the torch chain)
        if "weight_norm" in FUSED_GLUE and v.is_cuda and v.dtype == torch.float32 and g.dtype == torch.float32 and v.dim() == 2 and 1 <= g.numel() <= 1024 and \
                0 < v.numel() <= (1 << 24) and getattr(self, "_g_dim", None) in (0, 1) and g.numel() == v.shape[self._g_dim]:
            return WeightNormFunction.apply(v, g, self._g_dim)
        return v * (g / v.norm())


class LinearWN(_WeightNormed, torch.nn.Module):  # utils.py:291 (weight_norm_wrapper(Linear, g_dim=0, v_dim=None))
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        b = _leaky_relu_init_bound(in_features, out_features)  # PointNetModule applies leaky_relu_init to its layers (mods:651)
        self.bias = torch.nn.Parameter(torch.zeros(out_features, device=device)) if bias else None
        self._install_weight_norm(torch.empty((out_features, in_features), device=device).uniform_(-b, b), g_dim=0)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias)


class LinearLeakyReluFunction(torch.autograd.Function):
    """y = LeakyReLU(x @ w^T + b) for tall-and-skinny x ([tokens, <=128 channels]) on the streaming kernels of
    csrc/ln_mlp.hip (ln_linear_act_forward / _backward) instead of a BLAS GEMM with K = 4..32."""

    @staticmethod
    def forward(ctx, x, w, b, slope):
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        rows, cin = x.shape
        cout = w.shape[0]
        y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        _lib.check(lib.ln_linear_act_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, cin, cout, float(slope), _lib.ptr(y),
                                             _lib.stream_ptr(x.device)), "ln_linear_act_forward")
        ctx.save_for_backward(x, w, y)
        ctx.slope = float(slope)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, grad_y):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        rows, cin = x.shape
        cout = w.shape[0]
        dev = x.device
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty((cout,), dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = torch.empty((lib.ln_linear_act_backward_workspace_bytes(cin, cout),), dtype=torch.uint8, device=dev)
        _lib.check(lib.ln_linear_act_backward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(grad_y), rows, cin, cout, ctx.slope, _lib.ptr(gx),
                                              _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
                   "ln_linear_act_backward")
        return gx, gw, gb, None


_IDENTITY_ROWS = {}
_MFMA_LINEAR_WIDTHS = (16, 32, 48, 64, 96, 128, 192, 256)


def _identity_rows(device, rows: int) -> torch.Tensor:
    """int32 [rows, 1] = 0..rows-1: the neighbour list that turns a lattice convolution into a per-vertex linear layer."""
    bufs = _IDENTITY_ROWS.setdefault(device, [])
    if not bufs or bufs[-1].shape[0] < rows:
        # grown, never freed: a captured hipGraph may point at an earlier (shorter) buffer
        size = 65536
        while size < rows:
            size *= 2
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the identity neighbour list would be allocated inside a stream capture: run the step eagerly once first")
        bufs.append(torch.arange(size, dtype=torch.int32, device=device).unsqueeze(1))
    return bufs[-1][:rows]


def _conv_workspace(lib, rows: int, cin: int, cout: int, device):
    nbytes = int(lib.ln_conv_forward_workspace_bytes(rows, 1, cin, cout))
    return torch.empty((nbytes,), dtype=torch.uint8, device=device) if nbytes > 256 else None


class LinearMfmaFunction(torch.autograd.Function):
    """y = x @ w^T for a per-vertex 1x1 layer (GnRelu1x1 and friends: no bias, no activation) on the MFMA kernels of the lattice
    convolution (csrc/ln_conv.hip) — a convolution with a filter extent of 1 over the identity neighbour list; `w` [cout, cin]
    is that convolution's bank in its transposed layout, so nothing is copied.  Both gradients come from the same kernels:
    grad_x = conv(grad_y; w as a plain bank), grad_w = the filter gradient with the roles of x and grad_y swapped."""

    @staticmethod
    def forward(ctx, x, w):
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        rows, cin = x.shape
        cout = w.shape[0]
        ident = _identity_rows(x.device, rows)
        y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        # with its workspace the convolution takes the bf16x3 kernel from 4096 rows on (widths that are multiples of 32 / 16)
        ws = _conv_workspace(lib, rows, cin, cout, x.device)
        _lib.check(lib.ln_conv_forward_ws(_lib.ptr(ident), _lib.ptr(x), _lib.ptr(w), rows, 1, cin, cout, 2, _lib.ptr(y), _lib.ptr(ws),
                                          0 if ws is None else ws.numel(), _lib.stream_ptr(x.device)), "ln_conv_forward(1x1)")  # 2 = LN_CONV_TRANSPOSED_FILTER
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        rows, cin = x.shape
        cout = w.shape[0]
        dev = x.device
        ident = _identity_rows(dev, rows)
        stream = _lib.stream_ptr(dev)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        # weight gradient first, then the input gradient's convolution: the slab sum of the first rides in the bank split of the second
        ws = torch.empty((lib.ln_linear_backward_workspace_bytes(rows, cin, cout),), dtype=torch.uint8, device=dev)
        _lib.check(lib.ln_linear_backward(_lib.ptr(ident), _lib.ptr(x), _lib.ptr(grad_y), _lib.ptr(w), rows, cin, cout, _lib.ptr(gx), _lib.ptr(gw),
                                          _lib.ptr(ws), ws.numel(), stream), "ln_linear_backward")
        return gx, gw


def linear_leaky_relu(x, weight, bias, slope: float):
    """LeakyReLU(linear(x)) (slope < 0: plain linear) for [rows, channels] inputs.  Bias-free plain layers whose widths the
    convolution's MFMA tiles cover run as a 1x1 lattice convolution; other float32 CUDA rows with <= 128 channels take the
    streaming kernels; everything else goes through torch."""
    cin, cout = x.shape[1], weight.shape[0]
    gpu_rows = x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0
    if gpu_rows and bias is None and slope < 0 and cin in _MFMA_LINEAR_WIDTHS and cout in _MFMA_LINEAR_WIDTHS and x.shape[0] >= 64:
        return LinearMfmaFunction.apply(x, weight)
    ok = gpu_rows and cin <= 128 and cout <= 128 and cin * cout <= 10240
    if ok:
        return LinearLeakyReluFunction.apply(x, weight, bias, slope)
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.nn.functional.leaky_relu(y, slope) if slope >= 0 else y


_UNIT_WEIGHTS = {}


def _unit_weights(device, tokens: int) -> torch.Tensor:
    """float32 ones [tokens] (the weights of an unweighted segment reduce); grown, never freed: a captured hipGraph may point at
    an earlier, shorter buffer."""
    bufs = _UNIT_WEIGHTS.setdefault(device, [])
    if not bufs or bufs[-1].shape[0] < tokens:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the unit weights would be allocated inside a stream capture: run the step eagerly once first")
        size = 1 << 16
        while size < tokens:
            size *= 2
        bufs.append(torch.ones((size,), dtype=torch.float32, device=device))
    return bufs[-1][:tokens]


class PointNetReduceFunction(torch.autograd.Function):
    """The vertex side of PointNetModule (lattice_modules.py:688-712): per-vertex max of the token features with the barycentric
    weight of each winning token appended, vertices with fewer than four tokens and vertex 0 zeroed.  `distributed` [tokens, width]
    holds the barycentric weight in its last column and takes no gradient (it comes from DistributeLattice)."""

    @staticmethod
    def forward(ctx, x, distributed, lattice, splatting_indices):
        import ctypes as C
        lib = _lib.load()
        x = x.contiguous()
        tokens, c = x.shape
        if splatting_indices.numel() != tokens or distributed.shape[0] != tokens:
            raise ValueError("features, distributed rows and splat indices must describe the same tokens")
        m = lattice.nr_lattice_vertices()
        _, csr, max_seg, grp_row, _ = lattice._csr(splatting_indices)
        dev = x.device
        width = distributed.shape[1]
        ws = torch.empty((lib.ln_pointnet_reduce_workspace_bytes(m, c),), dtype=torch.uint8, device=dev)
        out = torch.empty((m, 2 * c), dtype=torch.float32, device=dev)
        arg = torch.empty((m, c), dtype=torch.int32, device=dev)
        _lib.check(lib.ln_pointnet_reduce_forward(C.byref(csr), _lib.ptr(grp_row), max_seg, _lib.ptr(x), c,
                                                  distributed.data_ptr() + 4 * (width - 1), width, m, 4, _lib.ptr(ws), ws.numel(), _lib.ptr(out),
                                                  _lib.ptr(arg), _lib.stream_ptr(dev)), "ln_pointnet_reduce_forward")
        ctx.save_for_backward(arg, splatting_indices)
        ctx.tokens, ctx.channels = tokens, c
        return out

    @staticmethod
    def backward(ctx, grad_out):
        arg, idx = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        gx = torch.empty((ctx.tokens, ctx.channels), dtype=torch.float32, device=grad_out.device)
        _lib.check(_lib.load().ln_pointnet_reduce_backward(_lib.ptr(grad_out), grad_out.shape[1], _lib.ptr(arg), _lib.ptr(idx), ctx.tokens,
                                                           ctx.channels, _lib.ptr(gx), _lib.stream_ptr(grad_out.device)),
                   "ln_pointnet_reduce_backward")
        return gx, None, None, None


class SplatLatticeModule(torch.nn.Module):  # lattice_modules.py:46-51
    def forward(self, lattice_py, positions, values):
        lv, ls_wrap, indices, weights = SplatLattice.apply(lattice_py, positions, values)
        return lv, ls_wrap.lattice, indices, weights


class DistributeLatticeModule(torch.nn.Module):  # lattice_modules.py:52-96
    """The per-vertex mean of the token positions (torch_scatter.scatter_mean in the reference, mods:78) is a segment
    reduce over the slot -> tokens adjacency the build emitted (no atomics, no extra host synchronisation); vertex 0 doubles
    as the bucket of the tokens that could not be inserted, and its tokens are zeroed (mods:72-94)."""

    def forward(self, lattice, positions, values, reset_hashmap=True):
        wrap, distributed, splatting_indices, splatting_weights = DistributeLattice.apply(lattice, positions, values, reset_hashmap)
        distributed_lattice = wrap.lattice
        pos_dim = positions.shape[1]
        nr_rows = distributed_lattice.nr_lattice_vertices()
        if "distribute" in FUSED_GLUE and distributed.is_cuda and distributed.dtype == torch.float32 and distributed.is_contiguous() and \
                not distributed.requires_grad and splatting_indices.dtype == torch.int32 and splatting_indices.is_contiguous():
            # sums of the positions straight from the token rows, degrees, then ONE pass for mean / subtract / zeroing
            tokens, width = distributed.shape
            sums = torch.zeros((nr_rows, pos_dim), dtype=torch.float32, device=distributed.device)
            distributed_lattice._scatter_rows(distributed, splatting_indices, _unit_weights(distributed.device, tokens), sums, pos_dim, 1, width)
            counts = distributed_lattice.vertex_point_counts(splatting_indices)
            centred = torch.empty_like(distributed)
            _lib.check(_lib.load().ln_distribute_centre(_lib.ptr(distributed), _lib.ptr(splatting_indices), _lib.ptr(sums), _lib.ptr(counts), tokens,
                                                        width, pos_dim, _lib.ptr(centred), _lib.stream_ptr(distributed.device)),
                       "ln_distribute_centre")
            return distributed_lattice, centred, splatting_indices, splatting_weights
        distributed_positions = distributed[:, :pos_dim].contiguous()
        sums = torch.zeros((nr_rows, pos_dim), dtype=distributed.dtype, device=distributed.device)
        ones = torch.ones((splatting_indices.numel(),), dtype=distributed.dtype, device=distributed.device)
        distributed_lattice._scatter_rows(distributed_positions, splatting_indices, ones, sums, pos_dim, 1, pos_dim)
        counts = distributed_lattice.vertex_point_counts(splatting_indices).to(distributed.dtype)
        mean_positions = sums / counts.clamp(min=1).unsqueeze(1)
        mean_positions[0, :] = 0  # vertex 0 doubles as the "invalid" bucket (mods:79-81)
        indices_long = splatting_indices.long().clamp(min=0)  # -1 -> bucket 0 (mods:72)
        distributed_mean_positions = torch.index_select(mean_positions, 0, indices_long)
        distributed = torch.cat([distributed_positions - distributed_mean_positions, distributed[:, pos_dim:]], dim=1)
        distributed = distributed.masked_fill((indices_long == 0).unsqueeze(1), 0)  # mods:88-94
        return distributed_lattice, distributed, splatting_indices, splatting_weights


class PointNetModule(torch.nn.Module):  # lattice_modules.py:618-733 (the step right before the hot path in LNN)
    """Per-token MLP -> vertex-wise max (+ the barycentric weight of the winning token) -> rows with fewer than 4
    points and vertex 0 zeroed -> lattice convolution.  Layers are weight-normalised as in the reference (LinearWN,
    ConvLatticeIm2RowWNModule), with its parameter names."""

    def __init__(self, nr_output_channels_per_layer, nr_outputs_last_layer, nr_input_channels=None, device="cuda"):
        super().__init__()
        self.nr_output_channels_per_layer = list(nr_output_channels_per_layer)
        self.nr_outputs_last_layer = nr_outputs_last_layer
        self.layers = torch.nn.ModuleList([])
        self.act = torch.nn.LeakyReLU(0.2)
        self.device = device
        self.last_conv = ConvLatticeIm2RowWNModule(self.nr_output_channels_per_layer[-1] * 2, nr_outputs_last_layer, 1, 1, True, device=device)
        if nr_input_channels is not None:
            self._make_layers(nr_input_channels)

    def _make_layers(self, nr_input_channels):
        for nr_out in self.nr_output_channels_per_layer:  # created lazily from the first input, mods:636-647
            self.layers.append(LinearWN(nr_input_channels, nr_out, bias=True, device=self.device))
            nr_input_channels = nr_out

    def forward(self, lattice_py, distributed, indices):
        if len(self.layers) == 0:
            self._make_layers(distributed.shape[1] - 1)
        barycentric_weights = distributed[:, -1]
        x = distributed[:, : distributed.shape[1] - 1]
        for layer in self.layers:  # linear + LeakyReLU(0.2) per token, fused (mods:669-671)
            x = linear_leaky_relu(x, layer.weight, layer.bias, self.act.negative_slope)
        # (the fused reduction hands raw pointers of `indices` to the kernels and returns no gradient for `distributed`: a caller that
        # wants gradients through the positions, or passes a strided index view, takes the torch chain below)
        if "pointnet" in FUSED_GLUE and x.is_cuda and x.dtype == torch.float32 and distributed.dtype == torch.float32 and \
                distributed.is_contiguous() and not distributed.requires_grad and indices.dtype == torch.int32 and indices.is_contiguous():
            # scatter_max + degrees + the winners' barycentric weights + both zeroing rules in three launches (mods:688-712)
            reduced = PointNetReduceFunction.apply(x, distributed, lattice_py, indices)
            lattice_py.set_values(reduced)
            reduced, lattice_py = self.last_conv(reduced, lattice_py)
            reduced = self.act(reduced)
            lattice_py.set_values(reduced)
            return reduced, lattice_py
        reduced, argmax = ScatterMaxLattice.apply(x, lattice_py, indices)               # mods:688
        nr_points = lattice_py.vertex_point_counts(indices).unsqueeze(1)                  # mods:692
        safe = torch.where(argmax >= 0, argmax, torch.zeros_like(argmax)).long()
        bary = torch.index_select(barycentric_weights, 0, safe.flatten()).view(argmax.shape[0], argmax.shape[1])  # mods:696-698
        reduced = torch.cat((reduced, bary), 1)
        reduced = reduced.masked_fill(nr_points < 4, 0)                                   # mods:705-707
        keep = torch.ones((reduced.shape[0], 1), dtype=reduced.dtype, device=reduced.device)
        keep[0] = 0                                                                       # vertex 0 = invalid bucket, mods:711-712
        reduced = reduced * keep
        lattice_py.set_values(reduced)
        reduced, lattice_py = self.last_conv(reduced, lattice_py)
        reduced = self.act(reduced)
        lattice_py.set_values(reduced)
        return reduced, lattice_py


class ExpandLatticeModule(torch.nn.Module):  # lattice_modules.py:98-118
    """Adds vertices around jittered copies of the positions (Lattice.cu:292-348)."""

    def __init__(self, point_multiplier, noise_stddev, expand_values):
        super().__init__()
        self.point_multiplier, self.noise_stddev, self.expand_values = point_multiplier, noise_stddev, expand_values

    def forward(self, lattice_values, lattice_structure, positions):
        lattice_structure.set_values(lattice_values)
        lv, ls_wrap = ExpandLattice.apply(lattice_values, lattice_structure, positions, self.point_multiplier, self.noise_stddev,
                                          self.expand_values)
        ls = ls_wrap.lattice
        ls.set_values(lv)
        return lv, ls


class ConvLatticeModule(torch.nn.Module):  # lattice_modules.py:120-172: the filter bank is sized from the first input
    def __init__(self, nr_filters, neighbourhood_size, dilation=1, bias=True, device="cuda"):
        super().__init__()
        self.nr_filters, self.neighbourhood_size, self.dilation, self.use_bias, self.device = nr_filters, neighbourhood_size, dilation, bias, device
        self.weight = None
        self.bias = None

    def forward(self, lattice_values, lattice_structure):
        lattice_structure.set_values(lattice_values)
        if self.weight is None:
            rows = lattice_structure.get_filter_extent(self.neighbourhood_size) * lattice_structure.val_dim()
            self.weight = torch.nn.Parameter(torch.empty(rows, self.nr_filters, device=self.device))
            _kaiming_uniform_fan_out_(self.weight)
            if self.use_bias:
                self.bias = torch.nn.Parameter(torch.empty(self.nr_filters, device=self.device))
                _bias_init_(self.bias, self.weight)
        lv, ls_wrap = ConvIm2RowLattice.apply(lattice_values, lattice_structure, self.weight, self.dilation)
        ls = ls_wrap.lattice
        if self.use_bias:
            lv = lv + self.bias
        ls.set_values(lv)
        return lv, ls


class ConvLatticeIm2RowModule(torch.nn.Module):  # lattice_modules.py:174-250
    def __init__(self, in_channels, out_channels, neighbourhood_size=1, dilation=1, bias=True, device="cuda"):
        super().__init__()
        self.neighbourhood_size = neighbourhood_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.dilation = dilation
        self.use_bias = bias
        self.filter_extent = Lattice.get_expected_filter_extent(neighbourhood_size)
        self.weight = torch.nn.Parameter(torch.empty(self.filter_extent * in_channels, out_channels, device=device))
        self.bias = torch.nn.Parameter(torch.empty(out_channels, device=device)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        _kaiming_uniform_fan_out_(self.weight)
        if self.bias is not None:
            _bias_init_(self.bias, self.weight)

    def forward(self, lattice_values, lattice_structure):
        lattice_structure.set_values(lattice_values)
        assert self.in_channels == lattice_structure.val_dim(), \
            f"in_channels {self.in_channels} does not match the lattice val_dim {lattice_structure.val_dim()}"
        # im2row + mm of the reference (mods:239-242) == one fused gather-GEMM here; same maths, no [M, E*V] tensor
        lv, ls_wrap = ConvIm2RowLattice.apply(lattice_values, lattice_structure, self.weight, self.dilation)
        ls = ls_wrap.lattice
        if self.use_bias:
            lv = lv + self.bias
        ls.set_values(lv)
        return lv, ls


class CoarsenLatticeModule(torch.nn.Module):  # lattice_modules.py:253-319
    def __init__(self, in_channels, out_channels, bias=False, device="cuda"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.neighbourhood_size = 1
        self.use_bias = bias
        self.filter_extent = Lattice.get_expected_filter_extent(1)
        self.weight = torch.nn.Parameter(torch.empty(self.filter_extent * in_channels, out_channels, device=device))
        self.bias = torch.nn.Parameter(torch.empty(out_channels, device=device)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        _kaiming_uniform_fan_out_(self.weight, fan_scale=0.5, std_scale=2.0)  # mods:277-283
        if self.bias is not None:
            _bias_init_(self.bias, self.weight)

    def forward(self, lattice_fine_values, lattice_fine_structure, coarsened_lattice=None):
        lattice_fine_structure.set_values(lattice_fine_values)
        assert self.in_channels == lattice_fine_structure.val_dim()
        lv, ls_wrap = CoarsenLattice.apply(lattice_fine_values, lattice_fine_structure, self.weight, coarsened_lattice)
        ls = ls_wrap.lattice
        if self.use_bias:
            lv = lv + self.bias
        ls.set_values(lv)
        return lv, ls


class FinefyLatticeModule(torch.nn.Module):  # lattice_modules.py:321-387
    def __init__(self, in_channels, out_channels, bias=False, device="cuda"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.neighbourhood_size = 1
        self.use_bias = bias
        self.filter_extent = Lattice.get_expected_filter_extent(1)
        self.weight = torch.nn.Parameter(torch.empty(self.filter_extent * in_channels, out_channels, device=device))
        self.bias = torch.nn.Parameter(torch.empty(out_channels, device=device)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        _kaiming_uniform_fan_out_(self.weight, fan_scale=0.5, std_scale=2.0)
        if self.bias is not None:
            _bias_init_(self.bias, self.weight)

    def forward(self, lattice_coarse_values, lattice_coarse_structure, lattice_fine_structure):
        lattice_coarse_structure.set_values(lattice_coarse_values)
        assert self.in_channels == lattice_coarse_structure.val_dim()
        lv, ls_wrap = FinefyLattice.apply(lattice_coarse_values, lattice_coarse_structure, lattice_fine_structure, self.weight)
        ls = ls_wrap.lattice
        if self.use_bias:
            lv = lv + self.bias
        ls.set_values(lv)
        return lv, ls


class _LatticeWN(_WeightNormed):
    """lattice_modules.py:413-415: weight_norm_wrapper(<lattice op module>, g_dim=1, v_dim=None) — one magnitude per
    output filter (column of the [E*V, F] bank)."""

    def _to_weight_norm(self, extent_scale: int = 1):
        v = self._parameters.pop("weight").data
        b = _leaky_relu_init_bound(self.in_channels, self.out_channels, max(self.filter_extent // extent_scale, 1))
        v.uniform_(-b, b)  # leaky_relu_init on the fused module (utils.py:424-462), then unfuse
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)
        self._install_weight_norm(v, g_dim=1)


class ConvLatticeIm2RowWNModule(_LatticeWN, ConvLatticeIm2RowModule):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._to_weight_norm()


class CoarsenLatticeWNModule(_LatticeWN, CoarsenLatticeModule):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._to_weight_norm(extent_scale=8)  # utils.py:437-443


class FinefyLatticeWNModule(_LatticeWN, FinefyLatticeModule):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._to_weight_norm(extent_scale=8)  # utils.py:444-452


class SliceLatticeModule(torch.nn.Module):  # lattice_modules.py:389-397
    def forward(self, lattice_values, lattice_structure, positions, splatting_indices=None, splatting_weights=None):
        lattice_structure.set_values(lattice_values)
        return SliceLattice.apply(lattice_values, lattice_structure, positions, splatting_indices, splatting_weights)


class GatherLatticeModule(torch.nn.Module):  # lattice_modules.py:399-408
    def forward(self, lattice_values, lattice_structure, positions, splatting_indices, splatting_weights):
        lattice_structure.set_values(lattice_values)
        return GatherLattice.apply(lattice_values, lattice_structure, positions, splatting_indices, splatting_weights)
