"""BTI trainer for Synapse / BTCV (13 organs): reference nnUNetTrainer_NexToU_BTI_Synapse.py:8-64."""
from ._bti_base import _TopologicalInteractionTrainer


class nnUNetTrainer_NexToU_BTI_Synapse(_TopologicalInteractionTrainer):
    # binary-tree exclusion list over the 13 BTCV organs (reference :43-44)
    inclusion_list = []
    exclusion_list = [[[1, 3, 5, 7, 8, 11, 13], [2, 4, 6, 9, 10, 12]], [[1, 3, 11, 13], [5, 7, 8]], [[1, 3], [11, 13]],
                      [1, 3], [11, 13], [[5, 8], [7]], [5, 8], [[4, 6, 10], [2, 9, 12]], [[4, 6], [10]], [4, 6],
                      [[9, 12], [2]], [9, 12]]
