"""NexToU encoder / decoder and the Pool-GNN / Swin-GNN blocks, on the MI355X graph kernels.

Drop-in mirror of the reference's ``network_architecture/NexToU_Encoder_Decoder.py``: the same
class names, constructor arguments, forward results and ``state_dict`` key grammar
(SURVEY.md §A.3), so checkpoints and the nnU-Net trainer plug-ins work unchanged.  What differs is
how a graph block runs:

* reference ``DyGraphConv`` / ``PoolDyGraphConv`` -> ``DenseDilatedKnnGraph`` -> ``MRConv``
  (:454-474, :516-551, :401-418) executes ~25 ATen ops and materialises the (B',N,M) distance
  matrix and two (B',C,N,k) gathers;
* here the same modules call two fused HIP operators (``graph_ops.knn_graph``,
  ``graph_ops.mr_aggregate``) on (B',C,N) channel-major tensors; neighbour ids stay int32 and
  the int64 ``edge_index`` only exists when the public ``MRConv.forward(x, edge_index, y)`` /
  ``DenseDilatedKnnGraph.forward`` signatures are used directly.

The dense convolution stages (StackedConvBlocks, transposed convs, 1x1 convs, norms) stay on
PyTorch-ROCm, as BASELINE.json's north_star prescribes.

Dead code of the reference is not reproduced (``Grapher`` :553-632, the unreachable Swin-only
stage branch :128-133/:289-295, the ``pool`` argument's pooling branch :114-118), see SURVEY §A.4.
"""
from __future__ import annotations

import functools
from typing import List, Sequence, Tuple, Type, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.modules.conv import _ConvNd
from torch.nn.modules.dropout import _DropoutNd

from .. import graph_ops
from .conv_blocks import StackedConvBlocks, get_matching_convtransp, maybe_convert_scalar_to_list
from .layout import is_channels_last_volume, set_stage_layout
from .norm_act import up_conv_cat
from .pos_embed import get_2d_relative_pos_embed, get_3d_relative_pos_embed  # noqa: F401 (re-export)
from .pos_embed import get_nd_relative_pos_embed
from .torch_edge import DenseDilatedKnnGraph
from .torch_nn import BasicConv, act_layer, batched_index_select  # noqa: F401 (re-export)


def _conv_dim(conv_op) -> int:
    if conv_op == nn.Conv2d:
        return 2
    if conv_op == nn.Conv3d:
        return 3
    raise NotImplementedError('conv operation [%s] is not found' % conv_op)


class OptInit:
    """Hard-coded GNN hyper-parameters (reference :17-32)."""

    def __init__(self, drop_path_rate=0., pool_op_kernel_sizes_len=4):
        self.pool_op_kernel_sizes_len = pool_op_kernel_sizes_len
        self.conv = 'mr'
        self.act = 'leakyrelu'
        self.norm = 'instance'
        self.bias = True
        self.dropout = 0.0
        self.use_dilation = True
        self.epsilon = 0.2
        self.use_stochastic = True
        self.drop_path = drop_path_rate
        self.blocks = [1] * pool_op_kernel_sizes_len
        self.reduce_ratios = [16, 8, 4, 2] + [1] * (pool_op_kernel_sizes_len - 4)


def _stage_shapes(conv_op, patch_size, strides):
    """Spatial shape and point count of every resolution stage (reference :70-99, :223-252)."""
    try:
        dim = _conv_dim(conv_op)
    except NotImplementedError:
        raise ValueError("unknown convolution dimensionality, conv op: %s" % str(conv_op))
    shape = [int(s) for s in patch_size[:dim]]
    shapes, sizes = [tuple(shape)], [int(np.prod(shape))]
    for pool in strides[1:]:
        shape = [s // int(p) for s, p in zip(shape, pool)]
        shapes.append(tuple(shape))
        sizes.append(int(np.prod(shape)))
    return shapes, sizes


def gnn_stage_hyperparameters(conv_op, img_min_shape, n_levels):
    """``(k_list, max_dilation, window_size)`` shared by every GNN block (reference :960-987,
    :1040-1067; table in SURVEY.md §A.1)."""
    dim = _conv_dim(conv_op)
    n_min = int(np.prod(img_min_shape))
    max_num = int(n_min // dim)
    max_k = min([2, 4, 8, 16, 32], key=lambda c: abs(c - max_num))
    min_k = max_num // (2 ** dim)
    k_list = [min(min_k * f, max_k) for f in (1, 2, 2, 4, 8)]
    if n_levels >= 5:
        k_list += [min(min_k * 16, max_k)] * (n_levels - 5)
    else:
        k_list = k_list[:n_levels]
    max_dilation = n_min // max(k_list)
    return k_list, max_dilation, tuple(img_min_shape)


def _query_pool_size(img_shape, img_min_shape):
    """Max-pool the queries by 2 on even axes iff the stage has more than 4^dim * N_min points
    (reference :490-503, :845-858)."""
    n = int(np.prod(img_shape))
    n_small = int(np.prod([h * 4 for h in img_min_shape]))
    if n > n_small:
        return [2 if h % 2 == 0 else 1 for h in img_shape]
    return [1 for _ in img_shape]


def _bicubic_taps(n_in: int, n_out: int):
    """Source rows and weights of torch's 1-D bicubic resize (align_corners=False, A = -0.75, border
    clamped): ``out[o] = sum_k w[k][o] * in[idx[k][o]]``.  The coordinates and the cubic weights are
    evaluated in float32 exactly as ATen's upsample_bicubic2d does for a float32 tensor — at 10 752
    output rows the float32 coordinate carries ~1e-3 of rounding, which is part of the reference's
    table (computing the taps in float64 would move the table by 2e-4)."""
    f = np.float32
    a = f(-0.75)
    scale = f(n_in) / f(n_out)
    src = scale * (np.arange(n_out, dtype=np.float32) + f(0.5)) - f(0.5)
    i0 = np.floor(src)
    t = (src - i0).astype(np.float32)
    i0 = i0.astype(np.int64)

    def inner(x):
        return ((a + f(2)) * x - (a + f(3))) * x * x + f(1)

    def outer(x):
        return ((a * x - f(5) * a) * x + f(8) * a) * x - f(4) * a

    weights = [outer(t + f(1)), inner(t), inner(f(1) - t), outer((f(1) - t) + f(1))]
    index = [np.clip(i0 - 1 + k, 0, n_in - 1) for k in range(4)]
    return index, [w.astype(np.float64) for w in weights]


def _resample_rows(table: np.ndarray, n_out: int) -> np.ndarray:
    index, weights = _bicubic_taps(table.shape[0], n_out)
    return sum(w[:, None] * table[i] for i, w in zip(index, weights))


@functools.lru_cache(maxsize=None)
def _relative_pos_table(dim: int, channels: int, n: int, r: int, exact: bool = True) -> torch.Tensor:
    """Frozen position bias of a graph block, shape (1, n, n // r**dim), already negated.

    Reference (:728-742, :867-880): float64 sin-cos table P (g**dim x C, ``g = int(n ** (1/dim))`` — the
    reference's floating root) -> ``S = 2 P P^T / C`` -> float32 -> bicubic resize to
    (n, n // r**dim) -> negate.  ``exact=True`` (default) does literally that, bit-identical to the
    reference's tables: a 10 648^2 float64 matrix per stage-2/3 Pool block of cfg 2 (~4 s each on the
    host), 24 389^2 = 4.8 GB for cfg 5.

    ``exact=False`` (or NEXTOU_FAST_RELPOS=1) uses that the resize is linear and separable,
    ``R S C^T = (2/C) (R P)(C P)^T``: two 4-tap row resamplings of P and one (n x C x m) product in
    float64 — no g**dim-squared matrix.  It agrees with the literal order to 2e-5 (the float32
    rounding of S and of the resize accumulations); since the kNN selection is discontinuous in the
    bias, models built from scratch for parity use the exact path, and checkpoints carry the table
    as a parameter anyway.  Cached: encoder and decoder blocks of one resolution share (C, n, r).
    """
    import os
    if os.environ.get("NEXTOU_FAST_RELPOS") == "1":
        exact = False
    grid = int(n ** (1 / dim))
    m = n // (r ** dim)
    if exact:
        table = torch.from_numpy(np.float32(get_nd_relative_pos_embed(channels, grid, dim)))
        table = F.interpolate(table[None, None], size=(n, m), mode='bicubic', align_corners=False)
        return -table.squeeze(1)
    from .pos_embed import _nd_sincos
    p = _nd_sincos(channels, grid, dim)                       # (g**dim, C) float64
    rows, cols = _resample_rows(p, n), _resample_rows(p, m)
    table = (2.0 / p.shape[1]) * (rows @ cols.T)
    return -torch.from_numpy(np.float32(table)).unsqueeze(0)


class _GrapherBase(nn.Module):
    """State shared by PoolGrapher and SwinGrapher: fc1 -> graph_conv -> fc2 -> + shortcut."""

    def _build_fc(self, conv_op, norm_op, norm_op_kwargs, c_in, c_out):
        return nn.Sequential(conv_op(c_in, c_out, 1, stride=1, padding=0), norm_op(c_out, **norm_op_kwargs))

    def _init_relative_pos(self, use, conv_op, channels, n, r):
        self.relative_pos = None
        if use:
            table = _relative_pos_table(_conv_dim(conv_op), channels, n, r)
            self.relative_pos = nn.Parameter(table.clone(), requires_grad=False)

    def _get_relative_pos(self, relative_pos, size_tuple):
        """Bias resized to another point count (reference :744-763, :882-901); identity at the
        construction size, which is the only size the assembled network ever uses."""
        n = int(np.prod(size_tuple))
        if relative_pos is None or n == self.n:
            return relative_pos
        n_reduced = n // (self.r ** len(size_tuple))
        return F.interpolate(relative_pos.unsqueeze(0), size=(n, n_reduced), mode="bicubic").squeeze(0)


class FFN(nn.Module):
    """1x1 conv + norm -> act -> 1x1 conv + norm -> + shortcut (reference :368-390)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act='relu', drop_path=0.0,
                 conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        norm_op_kwargs = norm_op_kwargs or {}
        self.fc1 = nn.Sequential(conv_op(in_features, hidden_features, 1, stride=1, padding=0),
                                 norm_op(hidden_features, **norm_op_kwargs))
        self.act = act_layer(act)
        self.fc2 = nn.Sequential(conv_op(hidden_features, out_features, 1, stride=1, padding=0),
                                 norm_op(out_features, **norm_op_kwargs))
        self.drop_path = _drop_path(drop_path)

    def forward(self, x):
        if isinstance(self.act, nn.Identity) and isinstance(self.drop_path, nn.Identity) and len(self.fc1) == 2 and len(self.fc2) == 2:
            # channels-last fp32 volume on the GPU: conv -> norm -> act -> conv -> norm -> + x as one K7 / K6 pipeline
            y = graph_ops.pointwise_chain(x, x, self.fc1[0], self.fc1[1], self.fc2[0], self.fc2[1])
            if y is not None:
                return y
        return self.drop_path(self.fc2(self.act(self.fc1(x)))) + x

    def _absorb_activation(self):
        """norm_act.fuse_norm_act hook: fold the LeakyReLU between fc1 and fc2 into fc1's fused norm."""
        norm = self.fc1[-1]
        if type(self.act) is nn.LeakyReLU and getattr(norm, 'negative_slope', None) == 1.0:
            norm.negative_slope = float(self.act.negative_slope)
            self.act = nn.Identity()


def _drop_path(rate):
    # drop_path is 0 in every constructible configuration (OptInit), i.e. always Identity
    if rate > 0.:
        from timm.models.layers import DropPath  # only needed for a non-default drop path rate
        return DropPath(rate)
    return nn.Identity()


class MRConv(nn.Module):
    """Max-relative graph convolution (reference :392-418): [x, max_j(x_j - x_i)] interleaved over
    channels -> grouped 1x1 conv -> norm -> act."""

    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True, conv_op=nn.Conv3d,
                 dropout_op=nn.Dropout3d):
        super().__init__()
        self.conv_op = conv_op
        self.nn = BasicConv([in_channels * 2, out_channels], act=act, norm=norm, bias=bias, drop=0.,
                            conv_op=conv_op, dropout_op=dropout_op)

    def _pointwise(self, feat):
        """(B,2C,N) -> BasicConv on the reference's (B,2C,N,1[,1]) view."""
        dim = _conv_dim(self.conv_op)
        return self.nn(feat.reshape(feat.shape[0], feat.shape[1], feat.shape[2], *([1] * (dim - 1))))

    def aggregate(self, x, nn_idx, y=None):
        """Fused path: x (B,C,N), y (B,C,M)|None, nn_idx int32 (B,N,k) -> (B,2C,N,1[,1])."""
        basic = self.nn
        if len(basic) >= 2 and isinstance(basic[0], (nn.Conv2d, nn.Conv3d)) and all(isinstance(m, nn.Identity) for m in list(basic)[2:]):
            # K2 + K7 (channel-major): aggregation, grouped 1x1 convolution and the InstanceNorm statistics in one launch, then K6's apply
            out = graph_ops.mr_grouped_cm_block(x, nn_idx, y, basic[0], basic[1])
            if out is not None:
                return out.reshape(out.shape[0], out.shape[1], out.shape[2], *([1] * (_conv_dim(self.conv_op) - 1)))
        return self._pointwise(graph_ops.mr_aggregate(x, nn_idx, y))

    def forward(self, x, edge_index, y=None):
        """Reference signature: x (B,C,N,1), edge_index int64 (2,B,N,k), y (B,C,M,1)|None."""
        b, c, n = x.shape[:3]
        src = None if y is None else y.reshape(y.shape[0], y.shape[1], -1)
        feat = graph_ops.mr_aggregate(x.reshape(b, c, n), edge_index[0], src, center_idx=edge_index[1])
        return self._pointwise(feat)


class GraphConv(nn.Module):
    """Static graph convolution layer (reference :420-432)."""

    def __init__(self, in_channels, out_channels, conv='edge', act='relu', norm=None, bias=True,
                 conv_op=nn.Conv3d, dropout_op=nn.Dropout3d):
        super().__init__()
        if conv != 'mr':
            raise NotImplementedError('conv:{} is not supported'.format(conv))
        self.gconv = MRConv(in_channels, out_channels, act, norm, bias, conv_op, dropout_op)

    def forward(self, x, edge_index, y=None):
        return self.gconv(x, edge_index, y)


class DyGraphConv(GraphConv):
    """Dynamic graph convolution: kNN graph on the current features, then MRConv
    (reference :434-474).  ``r > 1`` average-pools the candidate set."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv='edge', act='relu',
                 norm=None, bias=True, stochastic=False, epsilon=0.0, r=1, conv_op=nn.Conv3d,
                 dropout_op=nn.Dropout3d):
        super().__init__(in_channels, out_channels, conv, act, norm, bias, conv_op, dropout_op)
        self.k, self.d, self.r = kernel_size, dilation, r
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)
        self.conv_op, self.dropout_op = conv_op, dropout_op
        self.avg_pool = F.avg_pool2d if _conv_dim(conv_op) == 2 else F.avg_pool3d

    def _graph_forward(self, x, relative_pos):
        """x (B,C,*sp) -> (B,2C,N,1[,1]); shared with PoolDyGraphConv."""
        b, c = x.shape[:2]
        y = None
        if self.r > 1:
            y = self.avg_pool(x, self.r, self.r).reshape(b, c, -1)
        x = x.reshape(b, c, -1)
        nn_idx = self.dilated_knn_graph.neighbor_ids(x, y, relative_pos)
        return self.gconv.aggregate(x, nn_idx, y)

    def forward(self, x, relative_pos=None):
        _conv_dim(self.conv_op)
        spatial = x.shape[2:]
        out = self._graph_forward(x, relative_pos)
        return out.reshape(x.shape[0], -1, *spatial)


class PoolDyGraphConv(DyGraphConv):
    """Max-pool the queries, graph convolution, max-unpool (reference :476-551).

    The unpool reuses the arg-max locations of input channel ``c mod C`` for output channel ``c``
    (:536).  A stage whose pool size is all ones skips the identity pool / unpool pair.
    """

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv='edge', act='relu',
                 norm=None, bias=True, stochastic=False, epsilon=0.0, r=1, conv_op=nn.Conv3d,
                 dropout_op=nn.Dropout3d, img_shape=None, img_min_shape=None):
        super().__init__(in_channels, out_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                         epsilon, r, conv_op, dropout_op)
        self.pool_size = _query_pool_size(img_shape, img_min_shape)
        if _conv_dim(conv_op) == 2:
            self.max_pool_input = nn.MaxPool2d(self.pool_size, stride=self.pool_size, return_indices=True)
            self.max_unpool_output = nn.MaxUnpool2d(self.pool_size, stride=self.pool_size)
        else:
            self.max_pool_input = nn.MaxPool3d(self.pool_size, stride=self.pool_size, return_indices=True)
            self.max_unpool_output = nn.MaxUnpool3d(self.pool_size, stride=self.pool_size)

    def _forward_channels_last(self, x, relative_pos):
        """x: dense channels-last device tensor.  The query max-pool reads the volume's rows and emits channel-major
        values plus the winning cell of every pooling window (uint8); the unpool writes the channels-last output once,
        zeros included (graph_ops.pool_rows / cell_scatter) — no int64 indices, no cat, no memset, no layout copies."""
        b, c = x.shape[:2]
        full_spatial = tuple(x.shape[2:])
        pooled = any(p != 1 for p in self.pool_size)
        # a stage that does not pool (pool size all ones) takes the same two kernels: with one-voxel cells they are the
        # plain channels-last <-> channel-major tile transposes
        values, cell = graph_ops.pool_rows(x, self.pool_size)
        if pooled and graph_ops.tape_active():          # test hook: record / replay the reference's MaxPool indices
            flat = graph_ops.taped(lambda: graph_ops.cells_to_flat_indices(cell, full_spatial, self.pool_size), x.device)
            if graph_ops.tape_replaying():
                values, cell = graph_ops.pool_rows(
                    x, self.pool_size, forced_cell=graph_ops.flat_indices_to_cells(flat, full_spatial, self.pool_size))
        pooled_spatial = tuple(s // p for s, p in zip(full_spatial, self.pool_size))
        out = self._graph_forward(values.view(b, c, *pooled_spatial), relative_pos)
        return graph_ops.cell_scatter(out.reshape(b, out.shape[1], -1), cell, full_spatial, self.pool_size)

    def forward(self, x, relative_pos=None):
        _conv_dim(self.conv_op)
        if is_channels_last_volume(x):
            return self._forward_channels_last(x, relative_pos)
        pooled = any(p != 1 for p in self.pool_size)
        if pooled:
            full_spatial = x.shape[2:]
            values, computed = self.max_pool_input(x)
            indices = graph_ops.taped(lambda: computed, x.device)
            if indices is not computed:  # teacher-forced arg-max locations (test hook)
                values = x.flatten(2).gather(2, indices.flatten(2)).reshape(indices.shape)
            x = values
        spatial = x.shape[2:]
        out = self._graph_forward(x, relative_pos).reshape(x.shape[0], -1, *spatial)
        if pooled:
            out = self.max_unpool_output(out, torch.cat((indices, indices), 1), output_size=full_spatial)
        return out


class PoolGrapher(_GrapherBase):
    """Pool-GNN block (reference :820-933): fc1 -> PoolDyGraphConv -> fc2 -> + shortcut."""

    def __init__(self, in_channels, img_shape, kernel_size=9, dilation=1, conv='edge', act='relu', norm=None,
                 bias=True, stochastic=False, epsilon=0.0, r=1, n=196, drop_path=0.0, relative_pos=False,
                 conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs=None, dropout_op=nn.Dropout3d,
                 img_min_shape=None):
        super().__init__()
        norm_op_kwargs = norm_op_kwargs or {}
        self.channels, self.r, self.conv_op, self.img_shape = in_channels, r, conv_op, img_shape
        self.fc1 = self._build_fc(conv_op, norm_op, norm_op_kwargs, in_channels, in_channels)
        self.graph_conv = PoolDyGraphConv(in_channels, in_channels * 2, kernel_size, dilation, conv, act, norm,
                                          bias, stochastic, epsilon, r, conv_op, dropout_op,
                                          img_shape=img_shape, img_min_shape=img_min_shape)
        self.fc2 = self._build_fc(conv_op, norm_op, norm_op_kwargs, in_channels * 2, in_channels)
        self.drop_path = _drop_path(drop_path)
        self.pool_size = _query_pool_size(img_shape, img_min_shape)
        self.n = int(np.prod(img_shape)) // int(np.prod(self.pool_size))
        self._init_relative_pos(relative_pos, conv_op, in_channels, self.n, r)

    def forward(self, x):
        _conv_dim(self.conv_op)
        shortcut = x
        x = _conv_norm(self.fc1, x)
        pooled_shape = tuple(s // p for s, p in zip(x.shape[2:], self.pool_size))
        x = self.graph_conv(x, self._get_relative_pos(self.relative_pos, pooled_shape))
        if isinstance(self.drop_path, nn.Identity):
            return _conv_norm(self.fc2, x, shortcut)
        return self.drop_path(self.fc2(x)) + shortcut


def _conv_norm(seq, x, residual=None):
    """``seq`` = Sequential(1x1 conv, norm): the fused statistics-epilogue GEMM + apply (+ residual) when it qualifies
    (graph_ops.pointwise_chain), else the modules one by one."""
    if len(seq) == 2:
        y = graph_ops.pointwise_chain(x, residual, seq[0], seq[1])
        if y is not None:
            return y
    y = seq(x)
    return y if residual is None else y + residual


def window_partition(x, window_size):
    """(B,C,*spatial) -> (B * n_windows, C, *window_size); windows enumerate row-major
    (reference :634-660)."""
    if x.dim() not in (4, 5):
        raise NotImplementedError('len(x.shape) [%d] is equal to 4 or 5' % x.dim())
    dim = x.dim() - 2
    b, c = x.shape[:2]
    counts = [s // w for s, w in zip(x.shape[2:], window_size)]
    split = [b, c]
    for n_w, w in zip(counts, window_size):
        split += [n_w, w]
    x = x.reshape(split)
    count_axes = [2 + 2 * i for i in range(dim)]
    inner_axes = [3 + 2 * i for i in range(dim)]
    x = x.permute(0, *count_axes, 1, *inner_axes)
    return x.reshape(b * int(np.prod(counts)), c, *window_size)


def window_reverse(windows, window_size, size_tuple):
    """Inverse of :func:`window_partition` (reference :662-693)."""
    if windows.dim() not in (4, 5):
        raise NotImplementedError('len(x.shape) [%d] is equal to 4 or 5' % windows.dim())
    dim = windows.dim() - 2
    counts = [s // w for s, w in zip(size_tuple, window_size)]
    b = int(windows.shape[0] / int(np.prod(counts)))
    c = windows.shape[1]
    x = windows.reshape(b, *counts, c, *window_size)
    order = [0, 1 + dim]
    for i in range(dim):
        order += [1 + i, 2 + dim + i]
    return x.permute(order).reshape(b, c, *size_tuple)


class SwinGrapher(_GrapherBase):
    """Swin-GNN block (reference :695-818): cyclic shift -> windows -> fc1 -> DyGraphConv (self kNN
    inside each window) -> fc2 -> reverse -> unshift -> + shortcut.  No mask for wrapped windows."""

    def __init__(self, in_channels, img_shape, kernel_size=9, dilation=1, conv='edge', act='relu', norm=None,
                 bias=True, stochastic=False, epsilon=0.0, r=1, n=196, drop_path=0.0, relative_pos=False,
                 conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs=None, dropout_op=nn.Dropout3d,
                 window_size=[3, 6, 6], shift_size=[0, 0, 0]):
        super().__init__()
        norm_op_kwargs = norm_op_kwargs or {}
        self.channels, self.r, self.conv_op, self.img_shape = in_channels, r, conv_op, img_shape
        self.window_size, self.shift_size = window_size, shift_size
        self.fc1 = self._build_fc(conv_op, norm_op, norm_op_kwargs, in_channels, in_channels)
        # the graph conv inside windows normalises with BatchNorm whatever `norm` says (:714)
        self.graph_conv = DyGraphConv(in_channels, in_channels * 2, kernel_size, dilation, conv, act, 'batch',
                                      bias, stochastic, epsilon, r, conv_op, dropout_op)
        self.fc2 = self._build_fc(conv_op, norm_op, norm_op_kwargs, in_channels * 2, in_channels)
        self.drop_path = _drop_path(drop_path)
        self.n = int(np.prod(window_size))
        self._init_relative_pos(relative_pos, conv_op, in_channels, self.n, r)

    def forward(self, x):
        dim = _conv_dim(self.conv_op)
        shortcut = x
        size_tuple = tuple(x.shape[2:])
        assert size_tuple == tuple(self.img_shape), "input features has wrong size"
        if is_channels_last_volume(x) and isinstance(self.drop_path, nn.Identity) and self.graph_conv.r == 1 and \
                self._pointwise_ops_commute_with_windows():
            return self._forward_channels_last(x, size_tuple, dim)
        axes = tuple(range(2, 2 + dim))
        shifted = max(self.shift_size) > 0
        if shifted:
            x = torch.roll(x, shifts=tuple(-s for s in self.shift_size), dims=axes)
        x = self.fc1(window_partition(x, self.window_size))
        x = self.graph_conv(x, self._get_relative_pos(self.relative_pos, tuple(x.shape[2:])))
        x = window_reverse(self.fc2(x), self.window_size, size_tuple)
        if shifted:
            x = torch.roll(x, shifts=tuple(self.shift_size), dims=axes)
        return self.drop_path(x) + shortcut


def _swin_pointwise_ops_commute(self) -> bool:
    """fc1 / fc2 / the MRConv's norm may run on the whole volume instead of the (B * nW, C, window) tensors only when their
    statistics are sums over ALL points of the batch (BatchNorm, or running statistics): a per-sample norm (InstanceNorm as
    ``norm_op``) normalises per WINDOW in the reference, which the volume path would silently turn into per volume
    (ADVICE r2).  Anything else takes the window-major path."""
    batch_norm = nn.modules.batchnorm._BatchNorm
    return all(isinstance(m, batch_norm) for m in (self.fc1[1], self.fc2[1], self.graph_conv.gconv.nn[1]))


def _swin_forward_channels_last(self, x, size_tuple, dim):
    """SwinGrapher on a channels-last volume.  Only the two graph kernels need the (windows, C, points) rows; fc1, the
    MRConv's grouped 1x1 conv + norm + activation and fc2 are point-wise (and their batch statistics are sums over all
    points), so they commute with the window permutation and run on the NDHWC volume — where MIOpen's CK kernels need
    no layout transposes (they cost 1.34 ms of a 4.3 ms stage-2 block on window-major NCDHW tensors,
    profiles/r02_gnn_stage_profile.md).  The shift + partition and the reverse + un-shift are index math inside
    graph_ops.window_gather / window_scatter.  Same function as reference :766-818 up to the summation order of the batch
    statistics."""
    shift = tuple(self.shift_size) if max(self.shift_size) > 0 else (0,) * dim
    gc = self.graph_conv
    h = _conv_norm(self.fc1, x)
    windows = graph_ops.window_gather(h, self.window_size, shift)                      # (B * nW, C, Nw)
    nn_idx = gc.dilated_knn_graph.neighbor_ids(windows, None, self._get_relative_pos(self.relative_pos, tuple(self.window_size)))
    basic = gc.gconv.nn                                                                # grouped 1x1 conv -> norm (-> absorbed LeakyReLU)
    chain = len(basic) == 3 and isinstance(basic[2], nn.Identity) and len(self.fc2) == 2
    if chain:       # K2 + K7: aggregation, window reverse and the grouped convolution in one kernel, then the fused chain
        y = graph_ops.mr_grouped_chain(windows, nn_idx, x, basic[0], basic[1], self.fc2[0], self.fc2[1], size_tuple, self.window_size, shift)
        if y is not None:
            return y
    if len(basic) >= 2 and isinstance(basic[0], (nn.Conv2d, nn.Conv3d)):
        # below the chain's point threshold: still one launch for aggregate + window reverse + grouped convolution, then the norm module
        h = graph_ops.mr_grouped_conv(windows, nn_idx, basic[0], basic[1], x.shape[0], size_tuple, self.window_size, shift)
        if h is not None:
            for mod in list(basic)[1:]:         # the norm (with its absorbed activation) and whatever follows it in BasicConv
                h = mod(h)
            return self.fc2(h) + x
    agg = graph_ops.mr_aggregate(windows, nn_idx)                                      # (B * nW, 2C, Nw)
    vol = graph_ops.window_scatter(agg, size_tuple, self.window_size, shift)           # NDHWC (B, 2C, *size)
    if chain:
        y = graph_ops.pointwise_chain(vol, x, basic[0], basic[1], self.fc2[0], self.fc2[1])
        if y is not None:
            return y
    return self.fc2(basic(vol)) + x


SwinGrapher._forward_channels_last = _swin_forward_channels_last
SwinGrapher._pointwise_ops_commute_with_windows = _swin_pointwise_ops_commute


class _GNNBlocks(nn.Module):
    """One (Grapher, FFN) pair per entry of ``opt.blocks[index]`` (always 1)."""

    def _common(self, opt, conv_op, index):
        levels = opt.pool_op_kernel_sizes_len
        k_list, max_dilation, window = gnn_stage_hyperparameters(conv_op, opt.img_min_shape, levels)
        self.n_blocks = sum(opt.blocks)
        dpr = [v.item() for v in torch.linspace(0, opt.drop_path, self.n_blocks)]
        first = sum(opt.blocks[0:index])
        block_ids = [first + j for j in range(opt.blocks[index])]
        return k_list, max_dilation, window, dpr, block_ids

    def forward(self, x):
        return self.blocks(x)


class SwinGNNBlocks(_GNNBlocks):
    """Reference :935-1013: window = bottleneck shape, shift = window // 2, r = 1, k = k_list[index]."""

    def __init__(self, channels, img_shape, index, opt=None, conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d,
                 norm_op_kwargs=None, dropout_op=nn.Dropout3d, **kwargs):
        super().__init__()
        k_list, max_dilation, window, dpr, block_ids = self._common(opt, conv_op, index)
        shift = [w // 2 for w in window]
        blocks = []
        for idx in block_ids:
            blocks.append(nn.Sequential(
                SwinGrapher(channels, img_shape, k_list[index], min(idx // 4 + 1, max_dilation), opt.conv,
                            opt.act, opt.norm, opt.bias, opt.use_stochastic, opt.epsilon, 1,
                            n=int(np.prod(window)), drop_path=dpr[idx], relative_pos=True, conv_op=conv_op,
                            norm_op=norm_op, norm_op_kwargs=norm_op_kwargs, dropout_op=dropout_op,
                            window_size=window, shift_size=shift),
                FFN(channels, channels * 4, act=opt.act, drop_path=dpr[idx], conv_op=conv_op, norm_op=norm_op,
                    norm_op_kwargs=norm_op_kwargs)))
        self.blocks = nn.Sequential(*blocks)


class PoolGNNBlocks(_GNNBlocks):
    """Reference :1015-1092: k = k_list[index + stage_num], r = reduce_ratios[index + stage_num]."""

    def __init__(self, channels, img_shape, index, stage_num, opt=None, conv_op=nn.Conv3d,
                 norm_op=nn.BatchNorm3d, norm_op_kwargs=None, dropout_op=nn.Dropout3d, **kwargs):
        super().__init__()
        k_list, max_dilation, _window, dpr, block_ids = self._common(opt, conv_op, index)
        level = index + stage_num
        blocks = []
        for idx in block_ids:
            blocks.append(nn.Sequential(
                PoolGrapher(channels, img_shape, k_list[level], min(idx // 4 + 1, max_dilation), opt.conv,
                            opt.act, opt.norm, opt.bias, opt.use_stochastic, opt.epsilon,
                            opt.reduce_ratios[level], n=opt.n_size_list[level], drop_path=dpr[idx],
                            relative_pos=True, conv_op=conv_op, norm_op=norm_op, norm_op_kwargs=norm_op_kwargs,
                            dropout_op=dropout_op, img_min_shape=opt.img_min_shape),
                FFN(channels, channels * 4, act=opt.act, drop_path=dpr[idx], conv_op=conv_op, norm_op=norm_op,
                    norm_op_kwargs=norm_op_kwargs)))
        self.blocks = nn.Sequential(*blocks)


def _as_list(value, n):
    return [value] * n if isinstance(value, int) else value


class NexToU_Encoder(nn.Module):
    """Reference :34-173.  Stages 0 .. n-5 are plain conv stages; the last four are
    [1 conv block, PoolGNNBlocks, SwinGNNBlocks]."""

    def __init__(self,
                 input_channels: int,
                 patch_size: List[int],
                 n_stages: int,
                 features_per_stage: Union[int, List[int], Tuple[int, ...]],
                 conv_op: Type[_ConvNd],
                 kernel_sizes: Union[int, List[int], Tuple[int, ...]],
                 strides: Union[int, List[int], Tuple[int, ...]],
                 n_conv_per_stage: Union[int, List[int], Tuple[int, ...]],
                 conv_bias: bool = False,
                 norm_op: Union[None, Type[nn.Module]] = None,
                 norm_op_kwargs: dict = None,
                 dropout_op: Union[None, Type[_DropoutNd]] = None,
                 dropout_op_kwargs: dict = None,
                 nonlin: Union[None, Type[torch.nn.Module]] = None,
                 nonlin_kwargs: dict = None,
                 return_skips: bool = False,
                 nonlin_first: bool = False,
                 pool: str = 'conv'):
        super().__init__()
        kernel_sizes = _as_list(kernel_sizes, n_stages)
        features_per_stage = _as_list(features_per_stage, n_stages)
        n_conv_per_stage = _as_list(n_conv_per_stage, n_stages)
        strides = _as_list(strides, n_stages)
        assert len(kernel_sizes) == n_stages, "kernel_sizes must have as many entries as we have resolution stages (n_stages)"
        assert len(n_conv_per_stage) == n_stages, "n_conv_per_stage must have as many entries as we have resolution stages (n_stages)"
        assert len(features_per_stage) == n_stages, "features_per_stage must have as many entries as we have resolution stages (n_stages)"
        assert len(strides) == n_stages, "strides must have as many entries as we have resolution stages (n_stages). " \
                                         "Important: first entry is recommended to be 1, else we run strided conv drectly on the input"
        if pool != 'conv':
            raise RuntimeError("only strided-convolution downsampling (pool='conv') is supported")

        img_shape_list, n_size_list = _stage_shapes(conv_op, patch_size, strides)
        self.opt = OptInit(pool_op_kernel_sizes_len=len(strides))
        self.opt.img_min_shape = img_shape_list[-1]
        self.opt.n_size_list = n_size_list
        self.n_swin_gnn_stages = 0
        self.no_pool_gnn_stage_num = n_stages - 4
        self.n_conv_stages = self.no_pool_gnn_stage_num - self.n_swin_gnn_stages

        conv_args = (conv_bias, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                     nonlin_first)
        gnn_args = dict(opt=self.opt, conv_op=conv_op, norm_op=norm_op, norm_op_kwargs=norm_op_kwargs,
                        dropout_op=dropout_op)
        stages = []
        for s in range(n_stages):
            if s < self.n_conv_stages:
                body = StackedConvBlocks(n_conv_per_stage[s], conv_op, input_channels, features_per_stage[s],
                                         kernel_sizes[s], strides[s], *conv_args)
            else:
                body = nn.Sequential(
                    StackedConvBlocks(n_conv_per_stage[s] - 1, conv_op, input_channels, features_per_stage[s],
                                      kernel_sizes[s], strides[s], *conv_args),
                    PoolGNNBlocks(features_per_stage[s], img_shape_list[s], s - self.no_pool_gnn_stage_num,
                                  self.no_pool_gnn_stage_num, **gnn_args),
                    SwinGNNBlocks(features_per_stage[s], img_shape_list[s], s - self.n_conv_stages, **gnn_args))
            stages.append(nn.Sequential(body))
            input_channels = features_per_stage[s]

        self.stages = nn.Sequential(*stages)
        self.output_channels = features_per_stage
        self.strides = [maybe_convert_scalar_to_list(conv_op, i) for i in strides]
        self.return_skips = return_skips
        # what a decoder needs to know
        self.conv_op = conv_op
        self.norm_op, self.norm_op_kwargs = norm_op, norm_op_kwargs
        self.nonlin, self.nonlin_kwargs = nonlin, nonlin_kwargs
        self.dropout_op, self.dropout_op_kwargs = dropout_op, dropout_op_kwargs
        self.conv_bias = conv_bias
        self.kernel_sizes = kernel_sizes

    channels_last_stages = frozenset()   # set by NexToU.__init__ (network_architecture/layout.py)
    reduced_precision_layout_ok = None   # ... as is this: True iff the plain stages really run multiple-of-8 channel counts

    def forward(self, x):
        skips = []
        last = len(self.stages) - 1
        for s, stage in enumerate(self.stages):
            x = stage(set_stage_layout(x, s in self.channels_last_stages, self.reduced_precision_layout_ok))
            if self.return_skips and s < last:
                # two consumers (the next stage, the decoder's concatenation): where the stage ends in a fused norm its backward takes the
                # two gradients unsummed (graph_ops.skip_fork); elsewhere this is (x, x)
                x, skip = graph_ops.skip_fork(x)
                skips.append(skip)
            else:
                skips.append(x)
        return skips if self.return_skips else skips[-1]

    def compute_conv_feature_map_size(self, input_size):
        """Planner helper: number of conv feature-map values (reference :175-185)."""
        total = np.int64(0)
        for s, stage in enumerate(self.stages):
            input_size = [i // j for i, j in zip(input_size, self.strides[s])]
            total += np.prod([self.output_channels[s], *input_size], dtype=np.int64) * _count_convs(stage)
        return total


def _count_convs(module) -> int:
    return sum(1 for m in module.modules() if isinstance(m, (nn.Conv2d, nn.Conv3d)))


class NexToU_Decoder(nn.Module):
    """Reference :187-337.  Decoder stage s (lowest resolution first) mirrors encoder stage
    n-1-s; the first three carry Pool-GNN + Swin-GNN blocks; every stage owns a 1x1 seg head."""

    def __init__(self,
                 encoder: NexToU_Encoder,
                 patch_size: List[int],
                 strides: Union[int, List[int], Tuple[int, ...]],
                 num_classes: int,
                 n_conv_per_stage: Union[int, Tuple[int, ...], List[int]],
                 deep_supervision, nonlin_first: bool = False):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder
        self.num_classes = num_classes
        n_enc = len(encoder.output_channels)
        n_conv_per_stage = _as_list(n_conv_per_stage, n_enc - 1)
        assert len(n_conv_per_stage) == n_enc - 1, "n_conv_per_stage must have as many entries as we have " \
                                                   "resolution stages - 1 (n_stages in encoder - 1), " \
                                                   "here: %d" % n_enc
        conv_op = encoder.conv_op
        transpconv_op = get_matching_convtransp(conv_op=conv_op)
        img_shape_list, n_size_list = _stage_shapes(conv_op, patch_size, strides)
        self.opt = OptInit(pool_op_kernel_sizes_len=len(strides))
        self.opt.img_min_shape = img_shape_list[-1]
        self.opt.n_size_list = n_size_list
        self.n_swin_gnn_stages = 0
        self.no_pool_gnn_stage_num = n_enc - 4
        self.n_conv_stages = self.no_pool_gnn_stage_num - self.n_swin_gnn_stages

        conv_args = (encoder.conv_bias, encoder.norm_op, encoder.norm_op_kwargs, encoder.dropout_op,
                     encoder.dropout_op_kwargs, encoder.nonlin, encoder.nonlin_kwargs, nonlin_first)
        gnn_args = dict(opt=self.opt, conv_op=conv_op, norm_op=encoder.norm_op,
                        norm_op_kwargs=encoder.norm_op_kwargs, dropout_op=encoder.dropout_op)
        stages, transpconvs, seg_layers = [], [], []
        for s in range(1, n_enc):
            below, skip = encoder.output_channels[-s], encoder.output_channels[-(s + 1)]
            up = encoder.strides[-s]
            transpconvs.append(transpconv_op(below, skip, up, up, bias=encoder.conv_bias))
            level = n_enc - (s + 1)  # encoder stage this decoder stage mirrors
            kernel = encoder.kernel_sizes[level]
            if s < n_enc - self.no_pool_gnn_stage_num:
                stages.append(nn.Sequential(
                    StackedConvBlocks(n_conv_per_stage[s - 1] - 1, conv_op, 2 * skip, skip, kernel, 1, *conv_args),
                    PoolGNNBlocks(skip, img_shape_list[level], level - self.no_pool_gnn_stage_num,
                                  self.no_pool_gnn_stage_num, **gnn_args),
                    SwinGNNBlocks(skip, img_shape_list[level], level - self.n_conv_stages, **gnn_args)))
            else:
                stages.append(StackedConvBlocks(n_conv_per_stage[s - 1], conv_op, 2 * skip, skip, kernel, 1,
                                                *conv_args))
            # heads are always built so that deep-supervision checkpoints load for inference
            seg_layers.append(conv_op(skip, num_classes, 1, 1, 0, bias=True))

        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(transpconvs)
        self.seg_layers = nn.ModuleList(seg_layers)

    def forward(self, skips):
        """skips in encoder order (bottleneck last) -> list of logits, highest resolution first,
        or the single full-resolution tensor when ``deep_supervision`` is off."""
        x = skips[-1]
        outputs = []
        last = len(self.stages) - 1
        for s, stage in enumerate(self.stages):
            # decoder stage s works at the resolution (and in the memory layout) of encoder stage last - s
            x = set_stage_layout(x, (last - s) in self.encoder.channels_last_stages, self.encoder.reduced_precision_layout_ok)
            x = stage(up_conv_cat(self.transpconvs[s], x, skips[-(s + 2)]))     # torch.cat((up-convolution(x), skip), 1)
            if self.deep_supervision:
                outputs.append(self.seg_layers[s](x))
            elif s == last:
                outputs.append(self.seg_layers[-1](x))
        outputs = outputs[::-1]
        return outputs if self.deep_supervision else outputs[0]

    def compute_conv_feature_map_size(self, input_size):
        """Planner helper (the reference's version raises AttributeError on the GNN stages,
        :360 — SURVEY §A.4 'N'; this one counts conv, transposed-conv and head outputs)."""
        skip_sizes = []
        for s in range(len(self.encoder.strides) - 1):
            skip_sizes.append([i // j for i, j in zip(input_size, self.encoder.strides[s])])
            input_size = skip_sizes[-1]
        assert len(skip_sizes) == len(self.stages)
        total = np.int64(0)
        for s, stage in enumerate(self.stages):
            size, feats = skip_sizes[-(s + 1)], self.encoder.output_channels[-(s + 2)]
            total += np.prod([feats, *size], dtype=np.int64) * (_count_convs(stage) + 1)
            if self.deep_supervision or s == len(self.stages) - 1:
                total += np.prod([self.num_classes, *size], dtype=np.int64)
        return total
