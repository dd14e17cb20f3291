"""Dense (dilated) kNN graph construction on the MI355X kernel.

Interface mirror of the reference's ``network_architecture/torch_edge.py``: the distance helpers
(:12-55), ``dense_knn_matrix`` / ``xy_dense_knn_matrix`` (:58-110), ``DenseDilated`` (:113-136)
and ``DenseDilatedKnnGraph`` (:139-163) keep their names, arguments and output layout
(``edge_index`` = int64 ``(2, B, N, k)``, ``[0]`` neighbour ids, ``[1]`` centre ids).  All of them
call the fused HIP kernels of :mod:`nextou_amd.graph_ops`; the (B,N,M) distance matrix is only
materialised when a ``*_pairwise_distance`` helper is called directly.

Neighbour order is the canonical one of SURVEY.md §7 hard part 1: ascending (distance, index).
``torch.topk`` leaves the order of exact ties unspecified, so that is the only place the two can
differ; the > 10 000-point row chunking of the reference (:70-82) is a memory workaround with no
effect on results and has no counterpart here.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import graph_ops


def _points_major_to_channel_major(x):
    # the helpers below take (B, N, C) like the reference; the kernels want (B, C, N)
    return x.detach().transpose(2, 1).contiguous()


def pairwise_distance(x):
    """x (B,N,C) -> squared distances (B,N,N)."""
    return graph_ops.pairwise_sq_distance(_points_major_to_channel_major(x))


def part_pairwise_distance(x, start_idx=0, end_idx=1):
    """Rows [start_idx, end_idx) of :func:`pairwise_distance`."""
    return graph_ops.pairwise_sq_distance(_points_major_to_channel_major(x), None, start_idx, end_idx)


def xy_pairwise_distance(x, y):
    """x (B,N,C), y (B,M,C) -> squared distances (B,N,M)."""
    return graph_ops.pairwise_sq_distance(_points_major_to_channel_major(x),
                                          _points_major_to_channel_major(y))


def dense_knn_matrix(x, k=16, relative_pos=None):
    """x (B,C,N,1), taken as is (no normalisation) -> edge_index (2,B,N,k) int64."""
    nn_idx = graph_ops.knn_graph(x, None, relative_pos, k, normalize=False)
    return graph_ops.edge_index_from_nn_idx(nn_idx, 1)


def xy_dense_knn_matrix(x, y, k=16, relative_pos=None):
    """x (B,C,N,1), y (B,C,M,1), taken as is -> edge_index (2,B,N,k) int64, ids into y."""
    nn_idx = graph_ops.knn_graph(x, y, relative_pos, k, normalize=False)
    return graph_ops.edge_index_from_nn_idx(nn_idx, 1)


def reference_rng_stream() -> bool:
    """``NEXTOU_REFERENCE_RNG=1``: consume the global CPU generator exactly as the reference does — one ``torch.rand(1)`` per
    graph construction whenever ``stochastic`` is set, train or eval, whatever the dilation (reference :128; SURVEY.md §A.4
    lists the draw as a side effect not worth reproducing, VERDICT r2 asks for seeded runs to be able to follow the
    reference's stream).  Off by default: the draw is a host-side op in the middle of the forward."""
    import os
    return os.environ.get("NEXTOU_REFERENCE_RNG", "0") == "1"


def _dilate(index, k, dilation, stochastic, epsilon, training):
    """Pick k of the k*dilation neighbours along the last axis (reference :126-136).

    Regular: every ``dilation``-th.  Stochastic + training: with probability ``epsilon`` a
    random subset instead.  The reference draws ``torch.rand(1)`` on every call, train or eval
    (SURVEY.md §A.4 'N'); here the RNG is only touched when the draw can matter, unless
    :func:`reference_rng_stream` asks for the reference's stream.
    """
    if stochastic and (reference_rng_stream() or (training and epsilon > 0)):
        if float(torch.rand(1)) < epsilon and training:
            pick = torch.randperm(k * dilation)[:k].to(index.device)
            return index.index_select(index.dim() - 1, pick)
    return index[..., ::dilation]


class DenseDilated(nn.Module):
    """edge_index (2,B,N,k*d) -> (2,B,N,k)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k

    def forward(self, edge_index):
        return _dilate(edge_index, self.k, self.dilation, self.stochastic, self.epsilon, self.training)


class DenseDilatedKnnGraph(nn.Module):
    """Normalise over channels, find k*d nearest candidates, dilate."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)

    def neighbor_ids(self, x, y=None, relative_pos=None):
        """int32 (B,N,k) neighbour ids — what the fused MRConv path consumes (no int64 expansion)."""
        nn_idx = graph_ops.knn_graph(x, y, relative_pos, self.k * self.dilation)
        if self.dilation == 1 and not (self.stochastic and reference_rng_stream()):
            return nn_idx       # any choice of all k neighbours is the same set
        return _dilate(nn_idx, self.k, self.dilation, self.stochastic, self.epsilon, self.training).contiguous()

    def forward(self, x, y=None, relative_pos=None):
        nn_idx = graph_ops.knn_graph(x, y, relative_pos, self.k * self.dilation)
        edge_index = graph_ops.edge_index_from_nn_idx(nn_idx, 1)
        return self._dilated(edge_index)
