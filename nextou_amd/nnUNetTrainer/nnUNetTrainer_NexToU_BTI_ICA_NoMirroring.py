"""BTI trainer for the intracranial-artery dataset, no mirroring:
reference nnUNetTrainer_NexToU_BTI_ICA_NoMirroring.py:8-63."""
from ._bti_base import _NoMirroringMixin, _TopologicalInteractionTrainer


class nnUNetTrainer_NexToU_BTI_ICA_NoMirroring(_NoMirroringMixin, _TopologicalInteractionTrainer):
    inclusion_list = []
    exclusion_list = [[[7, 9, 11, 12, 14, 15, 16, 17, 18], [1, 2, 3, 4, 5, 6, 8, 10, 13]],
                      [[7, 9, 11, 12], [14, 15, 16, 17, 18]], [[7, 9], [11, 12]], [7, 9], [11, 12],
                      [[14, 15], [16, 17, 18]], [14, 15], [[16, 17], [18]], [16, 17],
                      [[3, 8, 10, 13], [1, 2, 4, 5, 6]], [[3, 10], [8, 13]], [3, 10], [8, 13],
                      [[1, 6], [2, 4, 5]], [1, 6], [[2, 4], [5]], [2, 4]]  # reference :43
