"""All-pairs TI trainer: reference nnUNetTrainer_NexToU_TI.py:9-72 (every pair of foreground classes
excludes each other, generated from ``dataset_json['labels']``, :10-13,48)."""
from itertools import combinations

from ..loss.compound_bti_loss import DC_and_CE_and_TI_Loss
from ._bti_base import _TopologicalInteractionTrainer


class nnUNetTrainer_NexToU_TI(_TopologicalInteractionTrainer):
    compound_loss = DC_and_CE_and_TI_Loss
    inclusion_list = []

    def generate_combinations(self, n):
        return [list(pair) for pair in combinations(range(1, n + 1), 2)]

    @property
    def exclusion_list(self):
        return self.generate_combinations(max(self.dataset_json["labels"].values()))
