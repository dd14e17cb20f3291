"""Binary topological interaction (BTI) loss on the MI355X critical-voxel kernel.

Interface mirror of the reference's ``loss/bti_loss.py:8-145`` (``BTI_Loss(dim, connectivity,
inclusion, exclusion, min_thick).forward(x, y) -> 0-dim float64``), and — because a scalar label is a
one-element set — of ``loss/ti_loss.py`` as well (``TI_Loss`` below).

The reference computes the critical-voxel map with, per interaction, two ``isin`` masks, two
float64 ``conv3d`` used as binary dilations and three ``where`` thresholds (:83-117).  That loop is
pure bit logic: give every interaction one bit, look the two bit masks of a voxel's label up in a
table, OR them over the neighbourhood and test ``(OR_c & a) | (OR_a & c)``.  Here the label map
comes from ``graph_ops.argmax_labels`` and the map from ``graph_ops.bti_critical_map`` (one HIP pass
over the uint8 volume); only the float64 cross-entropy of :141-143, through which all gradient
flows, is a second fused kernel (``graph_ops.critical_cross_entropy``).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .. import graph_ops

_BITS = 32  # interactions per kernel pass (one uint32 bit each)


def _label_set(spec) -> List[int]:
    """labels of one side of an interaction: 0-dim / 1-D tensor, int, or list of ints."""
    if isinstance(spec, torch.Tensor):
        return [int(v) for v in spec.detach().reshape(-1).tolist()]
    if isinstance(spec, (list, tuple, np.ndarray)):
        return [int(v) for v in np.asarray(spec).reshape(-1).tolist()]
    return [int(spec)]


class BTI_Loss(torch.nn.Module):
    def __init__(self, dim=3, connectivity=26, inclusion=[], exclusion=[], min_thick=1):
        """
        :param dim: 2 or 3
        :param connectivity: 4 or 8 in 2-D; 6 or 26 in 3-D
        :param inclusion: list of [A, B] label sets, A completely surrounded by B
        :param exclusion: list of [A, C] label sets that must not touch
        :param min_thick: minimum separation; only used for connectivity 8 / 26
        """
        super().__init__()
        if dim not in (2, 3):
            raise ValueError("dim must be 2 or 3, got %r" % (dim,))
        allowed = (4, 8) if dim == 2 else (6, 26)
        if connectivity not in allowed:
            raise ValueError("connectivity %r is not valid for dim=%d (use one of %s)" % (connectivity, dim, allowed))
        self.dim, self.connectivity, self.min_thick = dim, connectivity, min_thick
        self.sum_dim_list = [1, 2, 3] if dim == 2 else [1, 2, 3, 4]
        # same bookkeeping as the reference (:38-49): [is_inclusion, A, C]
        self.interaction_list = [[True, inc[0], inc[1]] for inc in inclusion] + \
                                [[False, exc[0], exc[1]] for exc in exclusion]
        for _, spec_a, spec_c in self.interaction_list:
            for label in _label_set(spec_a) + _label_set(spec_c):
                if not 0 <= label < 256:   # the label map is uint8; the reference (float64 compare) has no such limit
                    raise ValueError("BTI_Loss: interaction label %d outside [0, 256) is not representable in the "
                                     "uint8 label map of the HIP critical-voxel kernel" % label)
        self._luts = self._build_luts(self.interaction_list)
        self._device_luts = {}
        # True: one host sync per call, raises IndexError as the reference's CrossEntropyLoss does for such targets
        # (bti_loss.py:141).  "deferred": no host sync — the check runs on the device, a bad target turns the loss NaN and is
        # counted in a device counter that :meth:`check_targets` reads (and raises from) later; what a captured hipGraph step
        # uses (harness.GraphedTrainStep).  False: no check (targets known to be clean).
        self.validate_targets = True
        self._bad_targets = None

    @staticmethod
    def _build_luts(interactions: Sequence) -> List[np.ndarray]:
        """One (lut_a, lut_c) uint32[256] pair per group of 32 interactions (reference :90-98)."""
        groups = []
        for g0 in range(0, len(interactions), _BITS):
            lut_a = np.zeros(256, dtype=np.uint32)
            lut_c = np.zeros(256, dtype=np.uint32)
            for bit, (is_inclusion, spec_a, spec_c) in enumerate(interactions[g0:g0 + _BITS]):
                set_a = [l for l in _label_set(spec_a) if 0 <= l < 256]
                set_c = [l for l in _label_set(spec_c) if 0 <= l < 256]
                lut_a[set_a] |= np.uint32(1 << bit)
                if is_inclusion:  # C := not (C or A)
                    member = np.zeros(256, dtype=bool)
                    member[set_a] = True
                    member[set_c] = True
                    lut_c[~member] |= np.uint32(1 << bit)
                else:
                    lut_c[set_c] |= np.uint32(1 << bit)
            groups.append((lut_a, lut_c))
        return groups

    def _luts_on(self, device):
        key = str(device)
        if key not in self._device_luts:
            self._device_luts[key] = [
                (torch.from_numpy(a.view(np.int32).copy()).to(device), torch.from_numpy(c.view(np.int32).copy()).to(device))
                for a, c in self._luts]
        return self._device_luts[key]

    @torch.no_grad()
    def critical_voxels_from_labels(self, labels: torch.Tensor) -> torch.Tensor:
        """uint8 label map (B,*spatial) -> uint8 critical map of the same shape."""
        critical = None
        for lut_a, lut_c in self._luts_on(labels.device):
            part = graph_ops.bti_critical_map(labels, lut_a, lut_c, self.connectivity, self.min_thick)
            critical = part if critical is None else critical | part
        if critical is None:  # no interactions: the reference would fail on an unbound name (:117)
            raise UnboundLocalError("BTI_Loss needs at least one inclusion or exclusion interaction")
        return critical

    def binary_topological_interaction_module(self, P):
        """Reference signature (:76-117): float label map (B,1,*spatial) -> float64 {0,1} map."""
        labels = P.squeeze(1).to(torch.uint8).contiguous()
        return self.critical_voxels_from_labels(labels).unsqueeze(1).double()

    def forward(self, x, y):
        """x: logits (B,L,*spatial); y: labels (B,1,*spatial) in [0,L) -> 0-dim float64 loss."""
        n_classes = x.shape[1]
        if n_classes > 256:
            raise ValueError("BTI_Loss: %d classes do not fit the uint8 label map (at most 256)" % n_classes)
        bad, y8 = None, None
        if self.validate_targets == "deferred":
            # no host read and no ATen reduction: the uint8 label map the CE kernel needs anyway is produced by one kernel that also
            # ORs an out-of-range flag on the device
            flag = torch.zeros(1, dtype=torch.int32, device=y.device)
            y8 = graph_ops.checked_label_map(y[:, 0], n_classes, flag)
            bad = flag[0] > 0
            if self._bad_targets is None or self._bad_targets.device != y.device:
                self._bad_targets = torch.zeros((), dtype=torch.int64, device=y.device)
            self._bad_targets += flag[0]
        elif self.validate_targets:
            lo, hi = (float(v) for v in torch.stack(torch.aminmax(y.detach())).tolist())
            if lo < 0 or hi >= n_classes:
                raise IndexError("Target %d is out of bounds." % int(hi if hi >= n_classes else lo))
        labels = graph_ops.argmax_labels(x)                      # argmax(softmax(x,1),1), :132-134
        critical = self.critical_voxels_from_labels(labels)
        # :141-143  CE(x.double(), y, 'none') * critical, summed over voxels, mean over the batch —
        # one fused float64 kernel that only touches critical voxels
        per_sample = graph_ops.critical_cross_entropy(x, y[:, 0].to(torch.uint8) if y8 is None else y8, critical)
        loss = per_sample.mean()
        if bad is not None:
            loss = torch.where(bad, torch.full_like(loss, float("nan")), loss)
        return loss

    def check_targets(self) -> None:
        """With ``validate_targets = "deferred"``: raise (one host sync) if any call since the last check saw a target outside
        [0, n_classes) — the IndexError the eager path raises at once."""
        if self._bad_targets is not None and int(self._bad_targets) > 0:
            n = int(self._bad_targets)
            self._bad_targets.zero_()
            raise IndexError("Target out of bounds in %d BTI_Loss call(s) (deferred validation)." % n)


class TI_Loss(BTI_Loss):
    """All-pairs topological interaction loss (reference loss/ti_loss.py:8-145): the same module with
    scalar labels; more than 32 interactions run as several kernel passes OR-ed together."""

    def topological_interaction_module(self, P):
        """Reference name of the critical-voxel module in loss/ti_loss.py:76-117."""
        return self.binary_topological_interaction_module(P)
