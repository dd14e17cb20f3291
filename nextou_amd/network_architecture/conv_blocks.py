"""Plain convolution stages and helpers the NexToU encoder/decoder is assembled from.

The reference takes these from the un-vendored third-party package
``dynamic_network_architectures`` (version contemporaneous with nnU-Net v2.0; call sites
``NexToU_Encoder_Decoder.py:7-8, 125, 130, 136, 281, 291, 298`` and
``nnUNetTrainer_NexToU.py:10-11, 88``).  It is absent from ``/root/reference`` and from this
image, so this is a restatement of its published behaviour — **parity unpinned** by any
reference test — keeping the attribute names (``convs.N.conv / norm / nonlin / all_modules``)
that define the ``state_dict`` keys of published NexToU checkpoints.  When the
real package is importable it is used instead, so inside an nnU-Net installation nothing changes.

These stages are dense 3-D convolutions: they stay on PyTorch-ROCm (MIOpen / hipBLASLt, MFMA),
as BASELINE.json's north_star prescribes.
"""
from __future__ import annotations

from typing import List, Tuple, Type, Union

import numpy as np
import torch
from torch import nn
from torch.nn.modules.conv import _ConvNd
from torch.nn.modules.dropout import _DropoutNd

try:  # inside an nnU-Net v2 environment defer to the real building blocks
    from dynamic_network_architectures.building_blocks.simple_conv_blocks import StackedConvBlocks  # type: ignore
    from dynamic_network_architectures.building_blocks.helper import (  # type: ignore
        convert_conv_op_to_dim, convert_dim_to_conv_op, get_matching_batchnorm, get_matching_convtransp,
        maybe_convert_scalar_to_list)
    from dynamic_network_architectures.initialization.weight_init import InitWeights_He  # type: ignore
    HAVE_DYNAMIC_NETWORK_ARCHITECTURES = True
except ImportError:
    HAVE_DYNAMIC_NETWORK_ARCHITECTURES = False

    _DIM_OF_CONV = {nn.Conv1d: 1, nn.Conv2d: 2, nn.Conv3d: 3}
    _CONV_OF_DIM = {v: k for k, v in _DIM_OF_CONV.items()}
    _TRANSP_OF_DIM = {1: nn.ConvTranspose1d, 2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
    _BN_OF_DIM = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}

    def convert_conv_op_to_dim(conv_op: Type[_ConvNd]) -> int:
        if conv_op not in _DIM_OF_CONV:
            raise ValueError("Unknown dimension. Only 1d 2d and 3d conv are supported. got %s" % str(conv_op))
        return _DIM_OF_CONV[conv_op]

    def convert_dim_to_conv_op(dimension: int) -> Type[_ConvNd]:
        if dimension not in _CONV_OF_DIM:
            raise ValueError("Unknown dimension. Only 1, 2 and 3 are supported")
        return _CONV_OF_DIM[dimension]

    def get_matching_convtransp(conv_op: Type[_ConvNd] = None, dimension: int = None):
        assert (conv_op is None) != (dimension is None), "give exactly one of conv_op / dimension"
        return _TRANSP_OF_DIM[dimension if dimension is not None else convert_conv_op_to_dim(conv_op)]

    def get_matching_batchnorm(conv_op: Type[_ConvNd] = None, dimension: int = None):
        assert (conv_op is None) != (dimension is None), "give exactly one of conv_op / dimension"
        return _BN_OF_DIM[dimension if dimension is not None else convert_conv_op_to_dim(conv_op)]

    def maybe_convert_scalar_to_list(conv_op, scalar):
        if isinstance(scalar, (tuple, list, np.ndarray)):
            return scalar
        return [scalar] * convert_conv_op_to_dim(conv_op)

    class ConvDropoutNormReLU(nn.Module):
        """conv -> [dropout] -> norm -> nonlin (or conv -> [dropout] -> nonlin -> norm)."""

        def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias=False,
                     norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                     nonlin=None, nonlin_kwargs=None, nonlin_first=False):
            super().__init__()
            self.input_channels, self.output_channels = input_channels, output_channels
            stride = maybe_convert_scalar_to_list(conv_op, stride)
            self.stride = stride
            kernel_size = maybe_convert_scalar_to_list(conv_op, kernel_size)
            norm_op_kwargs = norm_op_kwargs or {}
            nonlin_kwargs = nonlin_kwargs or {}
            parts = []
            self.conv = conv_op(input_channels, output_channels, kernel_size, stride,
                                padding=[(i - 1) // 2 for i in kernel_size], dilation=1, bias=conv_bias)
            parts.append(self.conv)
            if dropout_op is not None:
                self.dropout = dropout_op(**(dropout_op_kwargs or {}))
                parts.append(self.dropout)
            if norm_op is not None:
                self.norm = norm_op(output_channels, **norm_op_kwargs)
                parts.append(self.norm)
            if nonlin is not None:
                self.nonlin = nonlin(**nonlin_kwargs)
                parts.append(self.nonlin)
            if nonlin_first and norm_op is not None and nonlin is not None:
                parts[-1], parts[-2] = parts[-2], parts[-1]
            self.all_modules = nn.Sequential(*parts)

        def forward(self, x):
            return self.all_modules(x)

    class StackedConvBlocks(nn.Module):
        """``num_convs`` ConvDropoutNormReLU blocks; the first one carries ``initial_stride``."""

        def __init__(self, num_convs, conv_op, input_channels, output_channels, kernel_size, initial_stride,
                     conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None,
                     dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None, nonlin_first=False):
            super().__init__()
            if not isinstance(output_channels, (tuple, list)):
                output_channels = [output_channels] * num_convs
            common = (conv_bias, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                      nonlin_first)
            blocks = [ConvDropoutNormReLU(conv_op, input_channels, output_channels[0], kernel_size,
                                          initial_stride, *common)]
            for i in range(1, num_convs):
                blocks.append(ConvDropoutNormReLU(conv_op, output_channels[i - 1], output_channels[i],
                                                  kernel_size, 1, *common))
            self.convs = nn.Sequential(*blocks)
            self.output_channels = output_channels[-1]
            self.initial_stride = maybe_convert_scalar_to_list(conv_op, initial_stride)

        def forward(self, x):
            return self.convs(x)

    class InitWeights_He(object):
        """kaiming_normal_(a=neg_slope) on every (transposed) conv weight, zero bias."""

        def __init__(self, neg_slope: float = 1e-2):
            self.neg_slope = neg_slope

        def __call__(self, module):
            if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
                module.weight = nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
                if module.bias is not None:
                    module.bias = nn.init.constant_(module.bias, 0)
