"""Small building blocks of the graph convolution: activation / norm factories, the grouped
1x1 ``BasicConv`` and ``batched_index_select``.

Interface mirror of the reference's ``network_architecture/torch_nn.py`` (act_layer :13-29,
norm_layer :32-51, BasicConv :66-92, batched_index_select :94-115): same names, argument
meaning, module ordering (conv -> norm -> act, hence the same ``state_dict`` keys ``0.*``,
``1.*``) and error behaviour.  ``batched_index_select`` runs on the HIP gather kernel instead of
the reference's transpose / flat-index / permute copies.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import graph_ops

_ACTIVATIONS = {
    "relu": lambda inplace, slope, n: nn.ReLU(inplace),
    "leakyrelu": lambda inplace, slope, n: nn.LeakyReLU(slope, inplace),
    "prelu": lambda inplace, slope, n: nn.PReLU(num_parameters=n, init=slope),
    "gelu": lambda inplace, slope, n: nn.GELU(),
    "hswish": lambda inplace, slope, n: nn.Hardswish(inplace),
}

_NORMS = {
    ("batch", nn.Conv2d): nn.BatchNorm2d,
    ("batch", nn.Conv3d): nn.BatchNorm3d,
    ("instance", nn.Conv2d): nn.InstanceNorm2d,
    ("instance", nn.Conv3d): nn.InstanceNorm3d,
}

# groups of the 1x1 convolution inside MRConv (reference torch_nn.py:73,77)
GROUPS_BY_CONV = {nn.Conv2d: 4, nn.Conv3d: 6}


def act_layer(act, inplace=True, neg_slope=1e-2, n_prelu=1):
    make = _ACTIVATIONS.get(act.lower())
    if make is None:
        raise NotImplementedError('activation layer [%s] is not found' % act)
    return make(inplace, neg_slope, n_prelu)


def norm_layer(norm, nc, conv_op):
    kind = norm.lower()
    if kind not in ("batch", "instance"):
        raise NotImplementedError('normalization layer [%s] is not found' % norm)
    cls = _NORMS.get((kind, conv_op))
    if cls is None:
        raise NotImplementedError('conv operation [%s] is not found' % conv_op)
    return cls(nc, affine=True)


class BasicConv(nn.Sequential):
    """[grouped 1x1 conv -> norm -> activation] per consecutive channel pair of ``channels``.

    As in the reference the norm layer is always sized by ``channels[-1]`` (:87-88).
    """

    def __init__(self, channels, act='relu', norm=None, bias=True, drop=0., conv_op=nn.Conv3d,
                 dropout_op=None):
        if conv_op not in GROUPS_BY_CONV:
            raise NotImplementedError('conv operation [%s] is not found' % conv_op)
        self.conv_op = conv_op
        self.groups_num = GROUPS_BY_CONV[conv_op]
        layers = []
        for c_in, c_out in zip(channels[:-1], channels[1:]):
            layers.append(conv_op(c_in, c_out, 1, bias=bias, groups=self.groups_num))
            if norm is not None and norm.lower() != 'none':
                layers.append(norm_layer(norm, channels[-1], conv_op))
            if act is not None and act.lower() != 'none':
                layers.append(act_layer(act))
        super().__init__(*layers)


def batched_index_select(x, idx):
    """``out[b,c,n,j] = x[b,c,idx[b,n,j]]`` for x (B,C,M,1), idx (B,N,k) -> (B,C,N,k)."""
    b, c, m = x.shape[:3]
    return graph_ops.gather_neighbors(x.reshape(b, c, m), idx)
