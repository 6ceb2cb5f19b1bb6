"""torch.autograd.Function operator surface — same class names, argument order and return tuples as
the reference's latticenet_py/lattice/lattice_funcs.py (cited per class), written against this
package's `Lattice`.  Forward/backward wiring follows the reference; the only structural change is
that the fused convolutions never rebuild the [M, E*V] rowified tensor in backward: the filter
gradient comes from `Lattice.convolve_im2row_grad_filter` (a transposed gather-GEMM).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from .lattice import Lattice  # noqa: F401
from .lattice_wrapper import LatticeWrapper

__all__ = ["ScatterMaxLattice", "SplatLattice", "DistributeLattice", "ExpandLattice", "Im2RowIndicesLattice", "Im2RowLattice", "ConvIm2RowLattice",
           "CoarsenLattice", "FinefyLattice", "SliceLattice", "SliceClassifyLattice", "GatherLattice"]


def _backward_filter(filter_bank: torch.Tensor, nr_filters: int, filter_extent: int, val_dim: int) -> torch.Tensor:
    """lattice_funcs.py:307-311: [E*V, F] -> [E*F, V] so that convolving the gradient with flipped
    neighbours yields grad wrt the input values."""
    fb = filter_bank.transpose(0, 1).reshape(nr_filters, filter_extent, val_dim)
    return fb.transpose(0, 1).contiguous().reshape(filter_extent * nr_filters, val_dim)


class ScatterMaxLattice(Function):
    """Vertex-wise max of per-token feature rows with argmax — the aggregation PointNetModule performs with
    torch_scatter.scatter_max (lattice_modules.py:688), computed on the CSR adjacency of the splat indices.
    Not part of the reference's lattice_funcs.py (torch_scatter is not available on the ROCm image)."""

    @staticmethod
    def forward(ctx, features, lattice, splatting_indices):
        features = features.contiguous()
        vmax, arg = lattice.scatter_max(features, splatting_indices)
        ctx.save_for_backward(arg)
        ctx.nr_tokens = features.shape[0]
        ctx.mark_non_differentiable(arg)
        return vmax, arg

    @staticmethod
    def backward(ctx, grad_max, grad_arg):
        (arg,) = ctx.saved_tensors
        valid = arg >= 0
        grad_src = torch.zeros((ctx.nr_tokens, grad_max.shape[1]), dtype=grad_max.dtype, device=grad_max.device)
        # every valid (vertex, channel) picked exactly one token of that vertex (conflict-free); vertices without a token
        # (argmax -1) are routed to row 0 with a zero contribution, so the scatter has to ADD: a plain scatter_ of those
        # zeros would race with the real gradient of whichever vertex holds token 0
        grad_src.scatter_add_(0, torch.where(valid, arg, torch.zeros_like(arg)).long(), torch.where(valid, grad_max, torch.zeros_like(grad_max)))
        return grad_src, None, None


class SplatLattice(Function):  # lattice_funcs.py:30-43
    @staticmethod
    def forward(ctx, lattice, positions, values):
        lattice.begin_splat()
        splatting_indices, splatting_weights = lattice.splat_standalone(positions, values)
        ctx.mark_non_differentiable(splatting_indices, splatting_weights)
        return lattice.values(), LatticeWrapper.wrap(lattice), splatting_indices, splatting_weights

    @staticmethod
    def backward(ctx, grad_lattice_values, grad_lattice_structure, grad_indices=None, grad_weights=None):
        return None, None, None


class DistributeLattice(Function):  # lattice_funcs.py:49-114
    @staticmethod
    def forward(ctx, lattice, positions, values, reset_hashmap=True):
        lattice.begin_splat(reset_hashmap)
        distributed_lattice, distributed, splatting_indices, splatting_weights = lattice.distribute(positions, values, reset_hashmap)
        ctx.save_for_backward(splatting_indices, splatting_weights)
        ctx.pos_dim = lattice.pos_dim()
        ctx.val_dim = lattice.val_dim()
        ctx.nr_positions = positions.shape[0]
        ctx.mark_non_differentiable(splatting_indices, splatting_weights)
        return LatticeWrapper.wrap(distributed_lattice), distributed, splatting_indices, splatting_weights

    @staticmethod
    def backward(ctx, grad_lattice, grad_distributed, grad_indices, grad_weights):
        pos_dim, val_dim, nr_positions = ctx.pos_dim, ctx.val_dim, ctx.nr_positions
        # each point's value row was copied to its pos_dim+1 vertices: sum those copies back (funcs:87-91)
        g = grad_distributed[:, pos_dim:pos_dim + val_dim].reshape(nr_positions, pos_dim + 1, val_dim)
        grad_values = g.sum(dim=1)
        return None, None, grad_values, None


class ExpandLattice(Function):  # lattice_funcs.py:118-143
    @staticmethod
    def forward(ctx, lattice_values, lattice_structure, positions, point_multiplier, noise_stddev, expand_values):
        lattice_structure.set_values(lattice_values)
        expanded_lattice = lattice_structure.expand(positions, point_multiplier, noise_stddev, expand_values)
        ctx.nr_values_original_lattice = lattice_structure.nr_lattice_vertices()
        return expanded_lattice.values(), LatticeWrapper.wrap(expanded_lattice)

    @staticmethod
    def backward(ctx, grad_lattice_values, grad_lattice_structure):
        return grad_lattice_values[0:ctx.nr_values_original_lattice, :], None, None, None, None, None


class Im2RowIndicesLattice(Function):  # lattice_funcs.py:145-185
    @staticmethod
    def forward(ctx, lattice_values, lattice, filter_extent, dilation, nr_filters):
        lattice.set_values(lattice_values)
        lattice_rowified = lattice.im2rowindices(lattice, filter_extent, dilation, False)
        ctx.lattice = lattice
        ctx.filter_extent, ctx.dilation, ctx.nr_filters = filter_extent, dilation, nr_filters
        ctx.mark_non_differentiable(lattice_rowified)
        return lattice_rowified

    @staticmethod
    def backward(ctx, grad_lattice_rowified):
        lattice = ctx.lattice
        grad_values = lattice.row2im(grad_lattice_rowified.contiguous(), ctx.dilation, ctx.filter_extent, ctx.nr_filters, lattice)
        ctx.lattice = 0
        return grad_values, None, None, None, None


class Im2RowLattice(Function):  # lattice_funcs.py:187-246
    @staticmethod
    def forward(ctx, lattice_values, lattice, filter_extent, dilation, nr_filters):
        lattice.set_values(lattice_values)
        lattice_rowified = lattice.im2row(lattice, filter_extent, dilation, False)
        ctx.lattice = lattice
        ctx.filter_extent, ctx.dilation, ctx.nr_filters = filter_extent, dilation, nr_filters
        ctx.val_dim = lattice.val_dim()
        return lattice_rowified

    @staticmethod
    def backward(ctx, grad_lattice_rowified):
        lattice = ctx.lattice
        if lattice.val_dim() != ctx.val_dim:  # funcs:227-229 (the reference exits the process)
            raise RuntimeError(f"lattice val_dim changed between forward ({ctx.val_dim}) and backward ({lattice.val_dim()})")
        grad_values = lattice.row2im(grad_lattice_rowified.contiguous(), ctx.dilation, ctx.filter_extent, ctx.nr_filters, lattice)
        ctx.lattice = 0
        return grad_values, None, None, None, None


class ConvIm2RowLattice(Function):  # lattice_funcs.py:250-320
    @staticmethod
    def forward(ctx, lattice_values, lattice, filter_bank, dilation):
        lattice.set_values(lattice_values)
        # fp16 features with fp32 master weights (the usual mixed-precision arrangement): the bank is rounded to fp16 here, per call,
        # and its gradient comes back in fp32 straight from the filter-gradient kernel's fp32 sums — no fp32 -> fp16 -> fp32 round trip
        ctx.mixed = lattice_values.dtype == torch.float16 and filter_bank.dtype == torch.float32
        if ctx.mixed:
            filter_bank = filter_bank.detach().half()
        convolved_lattice = lattice.convolve_im2row_standalone(filter_bank, dilation, lattice, False)
        ctx.save_for_backward(filter_bank, lattice_values)
        ctx.lattice = lattice
        ctx.filter_extent = int(filter_bank.shape[0] / lattice_values.shape[1])
        ctx.nr_filters = int(filter_bank.shape[1])
        ctx.dilation = dilation
        ctx.val_dim = lattice.val_dim()
        return convolved_lattice.values(), LatticeWrapper.wrap(convolved_lattice)

    @staticmethod
    def backward(ctx, grad_lattice_values, grad_lattice_structure):
        lattice = ctx.lattice
        nr_filters, dilation, val_dim = ctx.nr_filters, ctx.dilation, ctx.val_dim
        filter_bank, lattice_values = ctx.saved_tensors
        filter_extent = int(filter_bank.shape[0] / val_dim)
        lattice.set_values(lattice_values)
        # funcs:298-313: grad_filter = im2row^T @ grad, grad_values = conv(grad, flipped neighbours, re-laid-out bank);
        # both run as gather-GEMMs
        extra = {"filter_grad_fp32": True} if ctx.mixed else {}  # (only this backend knows the argument; other lattices never see it)
        grad_values, grad_filter = lattice.convolve_im2row_backward(grad_lattice_values, filter_bank, dilation, lattice, lattice, **extra)
        ctx.lattice = 0
        return grad_values, None, grad_filter, None


class CoarsenLattice(Function):  # lattice_funcs.py:323-398
    @staticmethod
    def forward(ctx, lattice_fine_values, lattice_fine_structure, filter_bank, coarsened_lattice=None):
        lattice_fine_structure.set_values(lattice_fine_values)
        positions = lattice_fine_structure.positions()
        if coarsened_lattice is None:
            coarsened_lattice = lattice_fine_structure.create_coarse_verts_naive(positions)  # funcs:335
        dilation = 1
        convolved_lattice = coarsened_lattice.convolve_im2row_standalone(filter_bank, dilation, lattice_fine_structure, False)
        ctx.save_for_backward(filter_bank, lattice_fine_values)
        ctx.coarsened_lattice = coarsened_lattice
        ctx.lattice_fine_structure = lattice_fine_structure
        ctx.filter_extent = int(filter_bank.shape[0] / lattice_fine_values.shape[1])
        ctx.nr_filters = int(filter_bank.shape[1])
        ctx.dilation = dilation
        ctx.val_dim = lattice_fine_structure.val_dim()
        return convolved_lattice.values(), LatticeWrapper.wrap(convolved_lattice)

    @staticmethod
    def backward(ctx, grad_lattice_values, grad_lattice_structure):
        coarsened_lattice = ctx.coarsened_lattice
        lattice_fine_structure = ctx.lattice_fine_structure
        nr_filters, dilation, val_dim = ctx.nr_filters, ctx.dilation, ctx.val_dim
        filter_bank, lattice_fine_values = ctx.saved_tensors
        filter_extent = int(filter_bank.shape[0] / val_dim)
        lattice_fine_structure.set_values(lattice_fine_values)
        # funcs:375-387: the filter gradient gathers fine values at the coarse vertices; the value gradient convolves at
        # the fine vertices with the coarse ones (which carry the errors) as neighbours
        grad_values, grad_filter = coarsened_lattice.convolve_im2row_backward(grad_lattice_values, filter_bank, dilation, coarsened_lattice,
                                                                              lattice_fine_structure)
        ctx.coarsened_lattice = 0
        ctx.lattice_fine_structure = 0
        return grad_values, None, grad_filter, None


class FinefyLattice(Function):  # lattice_funcs.py:401-462
    @staticmethod
    def forward(ctx, lattice_coarse_values, lattice_coarse_structure, lattice_fine_structure, filter_bank):
        lattice_coarse_structure.set_values(lattice_coarse_values)
        dilation = 1
        convolved_lattice = lattice_fine_structure.convolve_im2row_standalone(filter_bank, dilation, lattice_coarse_structure, False)
        ctx.save_for_backward(filter_bank, lattice_coarse_values)
        ctx.lattice_fine_structure = convolved_lattice
        ctx.lattice_coarse_structure = lattice_coarse_structure
        ctx.filter_extent = int(filter_bank.shape[0] / lattice_coarse_values.shape[1])
        ctx.nr_filters = int(filter_bank.shape[1])
        ctx.dilation = dilation
        ctx.val_dim = lattice_coarse_structure.val_dim()
        return convolved_lattice.values(), LatticeWrapper.wrap(convolved_lattice)

    @staticmethod
    def backward(ctx, grad_lattice_values, grad_lattice_structure):
        lattice_fine_structure = ctx.lattice_fine_structure
        lattice_coarse_structure = ctx.lattice_coarse_structure
        nr_filters, dilation, val_dim = ctx.nr_filters, ctx.dilation, ctx.val_dim
        filter_bank, lattice_coarse_values = ctx.saved_tensors
        filter_extent = int(filter_bank.shape[0] / val_dim)
        lattice_coarse_structure.set_values(lattice_coarse_values)
        grad_values, grad_filter = lattice_fine_structure.convolve_im2row_backward(grad_lattice_values, filter_bank, dilation,
                                                                                   lattice_fine_structure, lattice_coarse_structure)
        ctx.lattice_coarse_structure = 0
        ctx.lattice_fine_structure = 0
        return grad_values, None, None, grad_filter


class SliceLattice(Function):  # lattice_funcs.py:467-516
    @staticmethod
    def forward(ctx, lattice_values, lattice_structure, positions, splatting_indices=None, splatting_weights=None):
        lattice_structure.set_values(lattice_values)
        if splatting_indices is None and splatting_weights is None:
            sliced_values, splatting_indices, splatting_weights = lattice_structure.slice_standalone_no_precomputation(positions)
        else:
            # the [M, V] accumulator of the backward scatter is zero-filled by the forward launch when a gradient will be asked for
            # (fp32 also for fp16 features: the scatter accumulates in fp32 and the result is rounded once)
            ctx.grad_accumulator = torch.empty(lattice_values.shape, dtype=torch.float32, device=lattice_values.device) if (
                ctx.needs_input_grad[0] and lattice_values.dtype in (torch.float32, torch.float16) and lattice_values.is_contiguous()) else None
            sliced_values = lattice_structure.slice_standalone_with_precomputation(positions, splatting_indices, splatting_weights,
                                                                                   grad_accumulator=ctx.grad_accumulator)
        ctx.save_for_backward(positions, splatting_indices, splatting_weights)
        ctx.lattice_structure = lattice_structure
        return sliced_values

    @staticmethod
    def backward(ctx, grad_sliced_values):
        positions, splatting_indices, splatting_weights = ctx.saved_tensors
        lattice_structure = ctx.lattice_structure
        if lattice_structure.val_dim() != grad_sliced_values.shape[1]:  # funcs:501-502
            raise RuntimeError("the values stored in the lattice do not have the dimension of the gradient")
        grad_sliced_values = grad_sliced_values.contiguous()
        acc = getattr(ctx, "grad_accumulator", None)
        ctx.grad_accumulator = None  # single use: a second backward through this node gets a freshly zeroed buffer
        lattice_structure.slice_backwards_standalone_with_precomputation_no_homogeneous(positions, grad_sliced_values, splatting_indices,
                                                                                        splatting_weights, zeroed_accumulator=acc)
        lattice_values = lattice_structure.values()
        ctx.lattice_structure = 0
        return lattice_values, None, None, None, None


class SliceClassifyLattice(Function):  # lattice_funcs.py:518-567
    @staticmethod
    def forward(ctx, lattice_values, lattice_structure, positions, delta_weights, linear_clasify_weight, linear_clasify_bias, nr_classes,
                splatting_indices, splatting_weights):
        lattice_structure.set_values(lattice_values)
        initial_values = lattice_values
        class_logits = lattice_structure.slice_classify_with_precomputation(positions, delta_weights, linear_clasify_weight,
                                                                            linear_clasify_bias, nr_classes, splatting_indices,
                                                                            splatting_weights)
        ctx.save_for_backward(positions, initial_values, delta_weights, linear_clasify_weight, linear_clasify_bias, splatting_indices,
                              splatting_weights)
        ctx.lattice_structure = lattice_structure
        ctx.val_dim = lattice_values.shape[1]
        ctx.nr_classes = nr_classes
        return class_logits

    @staticmethod
    def backward(ctx, grad_class_logits):
        positions, initial_values, delta_weights, linear_clasify_weight, linear_clasify_bias, splatting_indices, splatting_weights = \
            ctx.saved_tensors
        lattice_py = ctx.lattice_structure
        grad_lattice_values = torch.zeros_like(initial_values)
        grad_delta_weights = torch.zeros_like(delta_weights)
        grad_linear_clasify_weight = torch.zeros_like(linear_clasify_weight)
        grad_linear_clasify_bias = torch.zeros_like(linear_clasify_bias)
        lattice_py.slice_classify_backwards_with_precomputation(grad_class_logits.contiguous(), positions, initial_values, delta_weights,
                                                                linear_clasify_weight, linear_clasify_bias, ctx.nr_classes,
                                                                grad_lattice_values, grad_delta_weights, grad_linear_clasify_weight,
                                                                grad_linear_clasify_bias, splatting_indices, splatting_weights)
        ctx.lattice_structure = 0
        return grad_lattice_values, None, None, grad_delta_weights, grad_linear_clasify_weight, grad_linear_clasify_bias, None, None, None


class GatherLattice(Function):  # lattice_funcs.py:569-603
    @staticmethod
    def forward(ctx, lattice_values, lattice_structure, positions, splatting_indices, splatting_weights):
        lattice_structure.set_values(lattice_values)
        gathered_values = lattice_structure.gather_standalone_with_precomputation(positions, splatting_indices, splatting_weights)
        ctx.save_for_backward(positions, splatting_indices, splatting_weights)
        ctx.lattice_structure = lattice_structure
        ctx.val_dim = lattice_values.shape[1]
        return gathered_values

    @staticmethod
    def backward(ctx, grad_sliced_values):
        positions, splatting_indices, splatting_weights = ctx.saved_tensors
        lattice_py = ctx.lattice_structure
        lattice_py.gather_backwards_standalone_with_precomputation(positions, grad_sliced_values.contiguous(), splatting_indices,
                                                                   splatting_weights)
        lattice_values = lattice_py.values()
        ctx.lattice_structure = 0
        return lattice_values, None, None, None, None
