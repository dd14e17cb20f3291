"""All-pairs TI trainer without mirroring: reference nnUNetTrainer_NexToU_TI_NoMirroring.py."""
from ._bti_base import _NoMirroringMixin
from .nnUNetTrainer_NexToU_TI import nnUNetTrainer_NexToU_TI


class nnUNetTrainer_NexToU_TI_NoMirroring(_NoMirroringMixin, nnUNetTrainer_NexToU_TI):
    pass
