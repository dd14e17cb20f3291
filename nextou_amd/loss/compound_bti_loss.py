"""``DC_and_CE_and_BTI_Loss`` — w_ce * CE + w_dice * Dice + w_ti * BTI.

Interface mirror of the reference's ``loss/compound_bti_loss.py:8-61`` (and, with ``TI_Loss``, of
``loss/compound_ti_loss.py``): same constructor arguments, ignore-label handling (:40-51) and
weighting (:58-60).  Dice and CE are nnU-Net's; the BTI term runs on the HIP critical-voxel kernel.
"""
from __future__ import annotations

import torch
from torch import nn

from .bti_loss import BTI_Loss, TI_Loss
from .nnunet_losses import RobustCrossEntropyLoss, SoftDiceLoss, softmax_helper_dim1


class DC_and_CE_and_BTI_Loss(nn.Module):
    _ti_class = BTI_Loss

    def __init__(self, soft_dice_kwargs, ce_kwargs, ti_kwargs, weight_ce=1, weight_dice=1, weight_ti=1e-6,
                 ignore_label=None, dice_class=SoftDiceLoss):
        """Weights for CE, Dice and TI do not need to sum to one."""
        super().__init__()
        if ignore_label is not None:
            ce_kwargs['ignore_index'] = ignore_label
        self.weight_dice, self.weight_ce, self.weight_ti = weight_dice, weight_ce, weight_ti
        self.ignore_label = ignore_label
        self.ce = RobustCrossEntropyLoss(**ce_kwargs)
        self.dc = dice_class(apply_nonlin=softmax_helper_dim1, **soft_dice_kwargs)
        self.ti = self._ti_class(**ti_kwargs)

    def forward(self, net_output: torch.Tensor, target: torch.Tensor):
        """target: (B,1,*spatial) label map."""
        if self.ignore_label is not None:
            assert target.shape[1] == 1, 'ignore label is not implemented for one hot encoded target variables ' \
                                         '(DC_and_CE_loss)'
            mask = (target != self.ignore_label).bool()
            # ignored voxels get label 0 for the dice term; their gradient is masked out anyway
            target_dice = torch.clone(target)
            target_dice[target == self.ignore_label] = 0
            num_fg = mask.sum()
        else:
            target_dice, mask = target, None

        dc_loss = self.dc(net_output, target_dice, loss_mask=mask) if self.weight_dice != 0 else 0
        ce_loss = self.ce(net_output, target[:, 0].long()) \
            if self.weight_ce != 0 and (self.ignore_label is None or num_fg > 0) else 0
        ti_loss = self.ti(net_output, target) if self.weight_ti != 0 else 0
        return self.weight_ce * ce_loss + self.weight_dice * dc_loss + self.weight_ti * ti_loss


class DC_and_CE_and_TI_Loss(DC_and_CE_and_BTI_Loss):
    """Reference loss/compound_ti_loss.py: identical with the all-pairs TI module."""
    _ti_class = TI_Loss
