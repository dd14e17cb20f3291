"""``NexToU`` — the nn.Module nnU-Net instantiates (drop-in boundary b of SURVEY.md §8).

Constructor signature, attributes (``encoder``, ``decoder``, ``decoder.deep_supervision``), forward
results and ``state_dict`` keys follow the reference's ``network_architecture/NexToU.py:11-63``.
"""
from __future__ import annotations

import os
from typing import List, Tuple, Type, Union

import torch
from torch import nn
from torch.nn.modules.conv import _ConvNd
from torch.nn.modules.dropout import _DropoutNd

from .NexToU_Encoder_Decoder import NexToU_Decoder, NexToU_Encoder
from .channel_pad import pad_multiple, pad_plain_stage_channels
from .conv_blocks import convert_conv_op_to_dim
from .. import graph_ops
from .layout import channels_last_stages, filters_to_channels_last
from .norm_act import attach_deferred_counters, fuse_norm_act, fuse_stem_block, fusion_enabled


class NexToU(nn.Module):
    def __init__(self,
                 input_channels: int,
                 patch_size: List[int],
                 n_stages: int,
                 features_per_stage: Union[int, List[int], Tuple[int, ...]],
                 conv_op: Type[_ConvNd],
                 kernel_sizes: Union[int, List[int], Tuple[int, ...]],
                 strides: Union[int, List[int], Tuple[int, ...]],
                 n_conv_per_stage: Union[int, List[int], Tuple[int, ...]],
                 num_classes: int,
                 n_conv_per_stage_decoder: Union[int, Tuple[int, ...], List[int]],
                 conv_bias: bool = False,
                 norm_op: Union[None, Type[nn.Module]] = None,
                 norm_op_kwargs: dict = None,
                 dropout_op: Union[None, Type[_DropoutNd]] = None,
                 dropout_op_kwargs: dict = None,
                 nonlin: Union[None, Type[torch.nn.Module]] = None,
                 nonlin_kwargs: dict = None,
                 deep_supervision: bool = False,
                 nonlin_first: bool = False):
        """nonlin_first: conv -> nonlin -> norm instead of conv -> norm -> nonlin."""
        super().__init__()
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * n_stages
        if isinstance(n_conv_per_stage_decoder, int):
            n_conv_per_stage_decoder = [n_conv_per_stage_decoder] * (n_stages - 1)
        assert len(n_conv_per_stage) == n_stages, \
            f"n_conv_per_stage must have as many entries as we have resolution stages. here: {n_stages}. " \
            f"n_conv_per_stage: {n_conv_per_stage}"
        assert len(n_conv_per_stage_decoder) == (n_stages - 1), \
            f"n_conv_per_stage_decoder must have one less entries as we have resolution stages. here: " \
            f"{n_stages} stages, so it should have {n_stages - 1} entries. " \
            f"n_conv_per_stage_decoder: {n_conv_per_stage_decoder}"
        self.encoder = NexToU_Encoder(input_channels, patch_size, n_stages, features_per_stage, conv_op,
                                      kernel_sizes, strides, n_conv_per_stage, conv_bias, norm_op,
                                      norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                                      return_skips=True, nonlin_first=nonlin_first)
        self.decoder = NexToU_Decoder(self.encoder, patch_size, strides, num_classes, n_conv_per_stage_decoder,
                                      deep_supervision, nonlin_first=nonlin_first)
        if fusion_enabled():  # (norm -> LeakyReLU) pairs become one K6 launch; state_dict unchanged
            fuse_norm_act(self)
            self.__dict__["_batch_counters"] = attach_deferred_counters(self)      # plain attribute: no module, no state
            # the plain conv stages run channels-last on the GPU (layout.py); needs K6's NDHWC kernels, hence here
            self.encoder.channels_last_stages = channels_last_stages(conv_op, self.encoder.n_conv_stages, n_stages)
            # 33 -> 40 / 66 -> 72 channels inside the plain conv stages (channel_pad.py): parameters keep their shapes
            self.padded_modules = pad_plain_stage_channels(self, pad_multiple())
            self.stem_block_fused = fuse_stem_block(self)       # K9 (round 6): first conv -> norm -> act block without the conv output in HBM
            if self.encoder.channels_last_stages:      # the filters of channels-last convolutions are stored channels-last as well
                filters_to_channels_last(self)
            # reduced-precision autocast keeps NDHWC only when the plain stages really run multiple-of-8 channel counts
            plain = list(self.encoder.output_channels)[:self.encoder.n_conv_stages]
            # (the hardware requirement is fixed — 16-byte bf16 / fp16 rows = multiples of 8 channels — whatever NEXTOU_PAD_CHANNELS says:
            # with padding switched off, or to a multiple of 4, the 33 / 66-channel stages must NOT claim it)
            runs_padded_to_8 = bool(self.padded_modules) and pad_multiple() % 8 == 0
            self.encoder.reduced_precision_layout_ok = runs_padded_to_8 or all(f % 8 == 0 for f in plain)

    def forward(self, x):
        counters = self.__dict__.get("_batch_counters")
        if counters is None or not self.training or os.environ.get("NEXTOU_STEP_GLUE", "1") == "0":     # ("0": per-norm launches, A/B)
            return self.decoder(self.encoder(x))
        # every batch norm's num_batches_tracked advances in one multi-tensor launch on the way out; the (exactly zero) gradients of the
        # convolution biases folded into statistics norms come from one zero-filled buffer (graph_ops.ZeroGradScope)
        with counters, graph_ops.ZERO_GRADS as riders:
            return riders.attach(self.decoder(self.encoder(x)))

    def compute_conv_feature_map_size(self, input_size):
        assert len(input_size) == convert_conv_op_to_dim(self.encoder.conv_op), \
            "just give the image size without color/feature channels or batch channel. Do not give " \
            "input_size=(b, c, x, y(, z)). Give input_size=(x, y(, z))!"
        return self.encoder.compute_conv_feature_map_size(input_size) + \
            self.decoder.compute_conv_feature_map_size(input_size)
