"""Training-side pieces of SURVEY.md §8f-4: the Lovasz-Softmax loss used next to NLL (ln_train.py:128-157, reference
latticenet_py/lattice/lovasz_loss.py:17-57) and the intersection-over-union bookkeeping (callbacks/scores.py).

The loss is the Lovasz extension of the per-class Jaccard index (Berman et al., CVPR 2018): for class c the errors
e_i = |1[y_i = c] - p_i(c)| are sorted in decreasing order and dotted with the discrete gradient of the Jaccard loss
along that order; classes absent from the cloud and the ignore class are skipped, the rest averaged.  All classes are
processed at once ([N, C] sort / cumsum) with no host synchronisation — the reference loops over classes and reads one
scalar per class back to the host.
"""
from __future__ import annotations

import os

import torch

__all__ = ["GeneralizedSoftDiceLoss", "LovaszSoftmax", "Scores", "nll_loss_gather"]


_NO_LABEL = -(1 << 62)  # "no ignore_index": a value no label takes
FUSED_NLL = True


class _NllFunction(torch.autograd.Function):
    """csrc/ln_glue.hip: two launches forward (partial sums in a fixed order, one finishing workgroup), one backward that writes the
    whole [n, C] gradient — instead of gather / mean and the zero fill + scatter of their autograd."""

    @staticmethod
    def forward(ctx, log_probs, target, ignore_index):
        from . import _lib
        lib = _lib.load()
        lp = log_probs.contiguous()
        n, c = lp.shape
        dev = lp.device
        ws = torch.empty((lib.ln_nll_workspace_bytes(),), dtype=torch.uint8, device=dev)
        loss_count = torch.empty((2,), dtype=torch.float32, device=dev)
        _lib.check(lib.ln_nll_forward(_lib.ptr(lp), _lib.ptr(target), n, c, ignore_index, _lib.ptr(ws), ws.numel(), _lib.ptr(loss_count),
                                      _lib.stream_ptr(dev)), "ln_nll_forward")
        ctx.save_for_backward(target, loss_count)
        ctx.shape, ctx.ignore_index = (n, c), ignore_index
        return loss_count[0]

    @staticmethod
    def backward(ctx, grad_loss):
        from . import _lib
        target, loss_count = ctx.saved_tensors
        n, c = ctx.shape
        g = grad_loss.contiguous().float()
        grad = torch.empty((n, c), dtype=torch.float32, device=g.device)
        _lib.check(_lib.load().ln_nll_backward(_lib.ptr(target), _lib.ptr(g), _lib.ptr(loss_count), n, c, ctx.ignore_index, _lib.ptr(grad),
                                               _lib.stream_ptr(g.device)), "ln_nll_backward")
        return grad, None, None


def nll_loss_gather(log_probs: torch.Tensor, target: torch.Tensor, ignore_index=None) -> torch.Tensor:
    """Mean negative log-likelihood, as torch.nn.NLLLoss(ignore_index=...) (ln_train.py:130).  float32 CUDA inputs take the fused
    kernels; everything else is written as gather + masked mean (torch's nll_loss kernels reduce a [120 k, C] input in a single
    workgroup: 0.14 ms forward, 0.09 ms backward on MI355X)."""
    target = target.reshape(-1)
    if FUSED_NLL and log_probs.is_cuda and log_probs.dtype == torch.float32 and log_probs.dim() == 2 and target.is_cuda and \
            target.dtype == torch.int64 and target.shape[0] == log_probs.shape[0] and log_probs.shape[0] > 0:
        return _NllFunction.apply(log_probs, target.contiguous(), _NO_LABEL if ignore_index is None else int(ignore_index))
    picked = log_probs.gather(1, target.clamp(0, log_probs.shape[1] - 1).unsqueeze(1)).squeeze(1)
    if ignore_index is None:
        return -picked.mean()
    keep = (target != ignore_index).to(picked.dtype)
    return -(picked * keep).sum() / keep.sum().clamp(min=1)


class GeneralizedSoftDiceLoss(torch.nn.Module):
    """The alternative to Lovasz-Softmax that ln_train.py:127 keeps at hand (reference latticenet_py/lattice/diceloss.py:172-209):
    with p = exp(log_probs) [N, C] and labels [N], per class  dice_c = 2 sum_i p_ic [y_i = c] / (sum_i p_ic + #{y_i = c} + 1e-6);
    loss = sum_c w_c (1 - dice_c) / C with w = 1 except w[ignore_index] = 0 (the division keeps all C classes, as there).
    No one-hot matrix is built: the intersection is a gather + index_add over the labels."""

    def __init__(self, p=1, smooth=1, reduction="mean", weight=1, ignore_index=None):
        super().__init__()
        self.p, self.smooth, self.reduction = p, smooth, reduction  # accepted and unused, as in the reference
        self.ignore_index = ignore_index

    def forward(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        nr_classes = output.shape[1]
        probs = output.exp()
        target = target.reshape(-1)
        picked = probs.gather(1, target.unsqueeze(1)).squeeze(1)
        intersection = torch.zeros(nr_classes, dtype=probs.dtype, device=probs.device).index_add_(0, target, picked)
        counts = torch.zeros(nr_classes, dtype=probs.dtype, device=probs.device).index_add_(0, target, torch.ones_like(picked))
        union = probs.sum(0) + counts
        loss_per_class = 1.0 - 2.0 * intersection / (union + 1e-6)
        if self.ignore_index is not None:
            weight = (torch.arange(nr_classes, device=probs.device) != int(self.ignore_index)).to(probs.dtype)
            loss_per_class = loss_per_class * weight
        return loss_per_class.sum() / nr_classes


class LovaszSoftmax(torch.nn.Module):
    def __init__(self, ignore_index, reduction: str = "mean"):
        super().__init__()
        self.ignore_index = ignore_index
        self.reduction = reduction

    def forward(self, inputs, targets):
        """inputs: log-probabilities [N, C] (the model's log-softmax, ln_train.py:156); targets: int64 [N]."""
        if inputs.dim() != 2:
            raise ValueError("LovaszSoftmax expects [N, C] log-probabilities")
        probs = inputs.exp()
        n, c = probs.shape
        targets = targets.reshape(-1)
        onehot = torch.zeros((n, c), dtype=probs.dtype, device=probs.device).scatter_(1, targets.clamp(0, c - 1).unsqueeze(1), 1.0)
        # class-major [C, N] so that every class is one contiguous row: row-wise sort / cumsum are far faster than
        # column-wise ones on an [N, C] matrix
        onehot = onehot.t().contiguous()
        errors = (onehot - probs.t()).abs()
        if errors.is_cuda and torch.cuda.is_current_stream_capturing() and os.environ.get("LN_LOVASZ_ROW_SORT"):
            # (kept as a switch: for a while the captured [C, N] sort looked unsafe to replay — the faults were those of training loops
            # that joined replays and eager kernels across streams; on the capture stream the matrix sort replays fine)
            parts = [torch.sort(errors[k], descending=True) for k in range(c)]
            errors_sorted = torch.stack([p[0] for p in parts])
            order = torch.stack([p[1] for p in parts])
        else:
            errors_sorted, order = torch.sort(errors, dim=1, descending=True)
        fg_sorted = torch.gather(onehot, 1, order)
        # gradient of the Jaccard loss along the sorted order: J_k = 1 - (G - cumsum fg) / (G + cumsum (1 - fg))
        gts = fg_sorted.sum(1, keepdim=True)
        intersection = gts - fg_sorted.cumsum(1)
        union = gts + (1.0 - fg_sorted).cumsum(1)
        jaccard = 1.0 - intersection / union
        grad = torch.cat((jaccard[:, :1], jaccard[:, 1:] - jaccard[:, :-1]), 1)
        per_class = (errors_sorted * grad).sum(1)
        present = gts.squeeze(1) > 0
        if self.ignore_index is not None and 0 <= int(self.ignore_index) < c:
            keep = torch.arange(c, device=probs.device) != int(self.ignore_index)  # (no host scalar: safe inside a stream capture)
            present = present & keep
        if self.reduction == "none":
            return per_class[present]
        total = (per_class * present).sum()
        if self.reduction == "sum":
            return total
        return total / present.sum().clamp(min=1)


class Scores:
    """Per-class intersection / union accumulated over clouds (callbacks/scores.py): IoU_c = I_c / U_c over the classes
    that occur (union > 0), mean over those.  Counts stay on the device; `all_reduce(dist)` sums them over ranks."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.start_fresh_eval()
        self.best_iou = -99999999
        self.best_iou_dict = {}

    def start_fresh_eval(self):
        self.intersection_per_class = None
        self.union_per_class = None
        self.nr_classes = None

    def accumulate_scores(self, pred_softmax, gt, unlabeled_idx):
        c = pred_softmax.shape[1]
        pred = pred_softmax.detach().argmax(1)
        gt = gt.detach().reshape(-1)
        if self.intersection_per_class is None:
            self.nr_classes = c
            self.intersection_per_class = torch.zeros(c, dtype=torch.int64, device=gt.device)
            self.union_per_class = torch.zeros(c, dtype=torch.int64, device=gt.device)
        hit = torch.bincount(gt[pred == gt], minlength=c)[:c]
        n_gt = torch.bincount(gt.clamp(0, c - 1), minlength=c)[:c]
        n_pred = torch.bincount(pred, minlength=c)[:c]
        in_gt = n_gt > 0  # the reference only scores the classes present in this cloud's ground truth
        if unlabeled_idx is not None and 0 <= int(unlabeled_idx) < c:
            in_gt[int(unlabeled_idx)] = False
        self.intersection_per_class += hit * in_gt
        self.union_per_class += (n_gt + n_pred - hit) * in_gt

    def all_reduce(self, dist):
        if dist is not None and self.intersection_per_class is not None:
            dist.all_reduce(self.intersection_per_class, op=dist.ReduceOp.SUM)
            dist.all_reduce(self.union_per_class, op=dist.ReduceOp.SUM)

    def compute_stats(self, print_per_class_iou=False):
        inter = self.intersection_per_class.tolist()
        union = self.union_per_class.tolist()
        iou_dict = {i: inter[i] / union[i] for i in range(self.nr_classes) if union[i] > 0}
        if print_per_class_iou:
            for i, v in iou_dict.items():
                print("class iou for idx", i, " is ", v)
        avg = sum(iou_dict.values()) / max(len(iou_dict), 1)
        return avg, iou_dict

    def avg_class_iou(self, print_per_class_iou=False):
        return self.compute_stats(print_per_class_iou)[0]

    def iou_per_class(self, print_per_class_iou=False):
        return self.compute_stats(print_per_class_iou)[1]

    def update_best(self):
        avg, d = self.compute_stats()
        if avg > self.best_iou:
            self.best_iou, self.best_iou_dict = avg, d
