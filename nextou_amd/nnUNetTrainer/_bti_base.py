"""Shared ``_build_loss`` of the BTI / TI trainer plug-ins.

The three BTI trainers of the reference (``nnUNetTrainer_NexToU_BTI_Synapse.py:17-64``,
``…_BTI_RAVIR.py:17-63``, ``…_BTI_ICA_NoMirroring.py:17-63``) and its two TI trainers differ only in
their interaction lists; the rest — deep-supervision weights 1/2^i with the last one zero (:23-27),
connectivity 26 / lambda 1e-6 in 3-D and 8 / 1e-4 in 2-D (:34-39), Dice(batch_dice, smooth 1e-5, no
background, ddp) + CE + lambda * (B)TI, the log lines (:53-59) and the DeepSupervisionWrapper (:63) —
is this one method.
"""
from __future__ import annotations

import numpy as np
import torch

from ..loss.compound_bti_loss import DC_and_CE_and_BTI_Loss, DC_and_CE_and_TI_Loss
from ..loss.nnunet_losses import DeepSupervisionWrapper, MemoryEfficientSoftDiceLoss
from .nnUNetTrainer_NexToU import nnUNetTrainer_NexToU


class _TopologicalInteractionTrainer(nnUNetTrainer_NexToU):
    inclusion_list: list = []
    exclusion_list: list = []
    compound_loss = DC_and_CE_and_BTI_Loss

    def make_tensors(self, lists, device):
        """nested int lists -> nested tensors on ``device`` (reference :9-15)."""
        if not lists:
            return lists
        if isinstance(lists[0], list):
            return [self.make_tensors(sub, device) for sub in lists]
        return torch.tensor(lists).to(device)

    def _build_loss(self):
        scales = self._get_deep_supervision_scales()
        weights = np.array([1 / (2 ** i) for i in range(len(scales))])
        weights[-1] = 0
        weights = weights / weights.sum()

        dim = len(self.configuration_manager.patch_size)
        # ECCV 2022 'Learning Topological Interactions …': lambda 1e-4 in 2-D, 1e-6 in 3-D
        connectivity, lambda_ti = (26, 1e-6) if dim == 3 else (8, 1e-4)
        inclusion = self.make_tensors(self.inclusion_list, self.device)
        exclusion = self.make_tensors(self.exclusion_list, self.device)
        loss = self.compound_loss(
            {'batch_dice': self.configuration_manager.batch_dice, 'smooth': 1e-5, 'do_bg': False, 'ddp': self.is_ddp},
            {},
            {'dim': dim, 'connectivity': connectivity, 'inclusion': inclusion, 'exclusion': exclusion, 'min_thick': 1},
            weight_ce=1, weight_dice=1, weight_ti=lambda_ti, ignore_label=self.label_manager.ignore_label,
            dice_class=MemoryEfficientSoftDiceLoss)
        self.print_to_log_file("dim: %s" % str(dim))
        self.print_to_log_file("connectivity: %s" % str(connectivity))
        self.print_to_log_file("lambda_ti: %s" % str(lambda_ti))
        self.print_to_log_file("inclusion_list: %s" % str(inclusion))
        self.print_to_log_file("exclusion_list_len: %s" % str(len(exclusion)))
        self.print_to_log_file("exclusion_list: %s" % str(exclusion))
        return DeepSupervisionWrapper(loss, weights)


class _NoMirroringMixin:
    """Disables mirror augmentation and mirror TTA (reference nnUNetTrainer_NexToU_NoMirroring.py:4-10)."""

    def configure_rotation_dummyDA_mirroring_and_inital_patch_size(self):
        rotation_for_DA, do_dummy_2d_data_aug, initial_patch_size, mirror_axes = \
            super().configure_rotation_dummyDA_mirroring_and_inital_patch_size()
        mirror_axes = None
        self.inference_allowed_mirroring_axes = None
        return rotation_for_DA, do_dummy_2d_data_aug, initial_patch_size, mirror_axes
