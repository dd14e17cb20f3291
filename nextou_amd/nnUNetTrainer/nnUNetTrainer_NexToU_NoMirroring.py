"""NexToU trainer without mirror augmentation / TTA: reference nnUNetTrainer_NexToU_NoMirroring.py:4-10."""
from ._bti_base import _NoMirroringMixin
from .nnUNetTrainer_NexToU import nnUNetTrainer_NexToU


class nnUNetTrainer_NexToU_NoMirroring(_NoMirroringMixin, nnUNetTrainer_NexToU):
    pass
