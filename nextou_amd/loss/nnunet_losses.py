"""Dice / cross-entropy / deep-supervision pieces the compound loss needs from nnU-Net v2.0.

The reference imports these from the un-vendored ``nnunetv2==2.0`` (call sites
``loss/compound_bti_loss.py:2-5,29-30``, ``nnUNetTrainer_NexToU_BTI_Synapse.py:3-5,49-51,63``).  Inside
an nnU-Net installation the real classes are used; otherwise these restatements of their published
behaviour stand in so that the standalone harness and bench can build the same loss — **parity
unpinned** (no reference test or golden vector covers third-party code).
"""
from __future__ import annotations

import torch
from torch import nn

try:
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss as _BaseMEDice, SoftDiceLoss as _BaseSoftDice  # type: ignore
    from nnunetv2.training.loss.robust_ce_loss import RobustCrossEntropyLoss as _BaseRobustCE  # type: ignore
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper  # type: ignore
    from nnunetv2.utilities.helpers import softmax_helper_dim1  # type: ignore
    HAVE_NNUNET = True
except ImportError:
    HAVE_NNUNET = False

    def softmax_helper_dim1(x: torch.Tensor) -> torch.Tensor:
        return torch.softmax(x, 1)

    class _AllGatherGrad(torch.autograd.Function):
        """all_gather whose backward returns this rank's slice of the summed gradient."""

        @staticmethod
        def forward(ctx, tensor):
            import torch.distributed as dist
            ctx.world = dist.get_world_size()
            gathered = [torch.zeros_like(tensor) for _ in range(ctx.world)]
            dist.all_gather(gathered, tensor.contiguous())
            return torch.stack(gathered, 0)

        @staticmethod
        def backward(ctx, grad):
            import torch.distributed as dist
            grad = grad.contiguous()
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
            return grad[dist.get_rank()]

    class _BaseMEDice(nn.Module):
        def __init__(self, apply_nonlin=None, batch_dice: bool = False, do_bg: bool = True, smooth: float = 1.,
                     ddp: bool = True):
            super().__init__()
            self.do_bg, self.batch_dice, self.apply_nonlin, self.smooth, self.ddp = \
                do_bg, batch_dice, apply_nonlin, smooth, ddp

        def forward(self, x, y, loss_mask=None):
            if self.apply_nonlin is not None:
                x = self.apply_nonlin(x)
            axes = tuple(range(2, x.dim()))
            with torch.no_grad():
                if x.dim() != y.dim():
                    y = y.view((y.shape[0], 1, *y.shape[1:]))
                if x.shape == y.shape:
                    y_onehot = y
                else:
                    y_onehot = torch.zeros(x.shape, device=x.device, dtype=torch.bool)
                    y_onehot.scatter_(1, y.long(), 1)
                if not self.do_bg:
                    y_onehot = y_onehot[:, 1:]
                sum_gt = y_onehot.sum(axes) if loss_mask is None else (y_onehot * loss_mask).sum(axes)
            if not self.do_bg:
                x = x[:, 1:]
            if loss_mask is None:
                intersect, sum_pred = (x * y_onehot).sum(axes), x.sum(axes)
            else:
                intersect, sum_pred = (x * y_onehot * loss_mask).sum(axes), (x * loss_mask).sum(axes)
            if self.batch_dice:
                if self.ddp:
                    intersect = _AllGatherGrad.apply(intersect).sum(0)
                    sum_pred = _AllGatherGrad.apply(sum_pred).sum(0)
                    sum_gt = _AllGatherGrad.apply(sum_gt).sum(0)
                intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
            dc = (2 * intersect + self.smooth) / (torch.clip(sum_gt + sum_pred + self.smooth, 1e-8))
            return -dc.mean()

    _BaseSoftDice = _BaseMEDice  # same value; the memory-hungry variant is not needed

    class _BaseRobustCE(nn.CrossEntropyLoss):
        """CrossEntropyLoss that accepts a (B,1,...) float target (restatement of nnU-Net's class)."""

        def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
            if target.ndim == input.ndim:
                assert target.shape[1] == 1
                target = target[:, 0]
            return super().forward(input, target.long())

    class DeepSupervisionWrapper(nn.Module):
        """sum_i w_i * loss(output_i, target_i); zero weights are skipped."""

        def __init__(self, loss, weight_factors=None):
            super().__init__()
            assert any(x != 0 for x in weight_factors), "At least one weight factor should be != 0.0"
            self.weight_factors = tuple(weight_factors)
            self.loss = loss

        def forward(self, *args):
            assert all(isinstance(i, (tuple, list)) for i in args), \
                f"all args must be either tuple or list, got {[type(i) for i in args]}"
            weights = self.weight_factors if self.weight_factors is not None else (1,) * len(args[0])
            return sum(weights[i] * self.loss(*inputs) for i, inputs in enumerate(zip(*args)) if weights[i] != 0.0)


class RobustCrossEntropyLoss(_BaseRobustCE):
    """nnU-Net's ``RobustCrossEntropyLoss`` (the real class inside an nnU-Net installation, its restatement otherwise) whose plain
    mean-reduced case runs as ONE kernel each way over the logits where they lie (K5c, ``graph_ops.cross_entropy_mean``: channels-last
    logits stay channels-last, no log_softmax / nll_loss passes, no layout copies).  Defined for both import outcomes (wired
    into the fallback class only, the trainers' deep-supervision CE never took the kernel inside nnU-Net).  Class weights, label
    smoothing, other reductions and ineligible tensors (CPU, reduced precision, > 32 classes) take the base class unchanged."""

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if self.weight is None and self.reduction == "mean" and self.label_smoothing == 0.0:
            t = target
            if t.ndim == input.ndim:
                assert t.shape[1] == 1
                t = t[:, 0]
            t = t.long()
            from .. import graph_ops
            if graph_ops.cross_entropy_mean_eligible(input, t):
                return graph_ops.cross_entropy_mean(input, t, self.ignore_index)
        return super().forward(input, target)


def _gathered_sum(t: torch.Tensor) -> torch.Tensor:
    """batch-dice under DDP: all-gather with gradient, summed over the ranks (nnU-Net's AllGatherGrad / the restatement above)."""
    if HAVE_NNUNET:
        from nnunetv2.utilities.ddp_allgather import AllGatherGrad  # type: ignore
        return AllGatherGrad.apply(t).sum(0)
    return _AllGatherGrad.apply(t).sum(0)


class _FusedDiceMixin:
    """nnU-Net's soft Dice with the three volume reductions as ONE kernel each way (K5d, ``graph_ops.dice_stats``): with
    ``softmax_helper_dim1`` as the non-linearity and a label-map target, ``intersect``, ``sum_pred`` and ``sum_gt`` come from a single
    pass over the logits in their own layout, and the rest — ``do_bg``, ``batch_dice`` (+ DDP all-gather), ``smooth``, the clip, the mean
    — is the base class's formula on (B, L) numbers.  Anything else (one-hot targets, another non-linearity, CPU, reduced precision)
    takes the base class unchanged.  2 tp + fp + fn = sum_pred + sum_gt, so the same sums serve SoftDiceLoss."""

    def forward(self, x, y, loss_mask=None):
        from .. import graph_ops
        # the kernel uses the mask as a 0 / 1 flag and knows nothing of nnunetv2's `clip_tp`: a fractional (float) mask or a clip
        # takes the base class, whose arithmetic it is
        mask_ok = loss_mask is None or loss_mask.dtype in (torch.bool, torch.uint8)
        if self.apply_nonlin is softmax_helper_dim1 and getattr(self, "clip_tp", None) is None and mask_ok and \
                graph_ops.dice_stats_eligible(x, y):
            intersect, sum_pred, sum_gt = graph_ops.dice_stats(x, y, loss_mask)
            if not self.do_bg:
                intersect, sum_pred, sum_gt = intersect[:, 1:], sum_pred[:, 1:], sum_gt[:, 1:]
            if self.batch_dice:
                if getattr(self, "ddp", False):
                    intersect, sum_pred, sum_gt = _gathered_sum(intersect), _gathered_sum(sum_pred), _gathered_sum(sum_gt)
                intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
            dc = (2 * intersect + self.smooth) / (torch.clip(sum_gt + sum_pred + self.smooth, 1e-8))
            return -dc.mean()
        return super().forward(x, y, loss_mask)


class MemoryEfficientSoftDiceLoss(_FusedDiceMixin, _BaseMEDice):
    pass


class SoftDiceLoss(_FusedDiceMixin, _BaseSoftDice):
    pass
