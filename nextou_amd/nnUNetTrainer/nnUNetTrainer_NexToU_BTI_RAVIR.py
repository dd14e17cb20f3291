"""BTI trainer for RAVIR (artery / vein): reference nnUNetTrainer_NexToU_BTI_RAVIR.py:8-63."""
from ._bti_base import _TopologicalInteractionTrainer


class nnUNetTrainer_NexToU_BTI_RAVIR(_TopologicalInteractionTrainer):
    inclusion_list = []
    exclusion_list = [[1, 2]]  # reference :43
