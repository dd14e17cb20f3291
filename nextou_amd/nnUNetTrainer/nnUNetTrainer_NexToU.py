"""``nnUNetTrainer_NexToU`` — the trainer plug-in nnU-Net v2 finds by class name.

Mirror of the reference's ``nnUNetTrainer/nnUNetTrainer_NexToU.py:17-91``: the static
``build_network_architecture`` derives ``conv_op`` from the plans, ``features_per_stage =
min(base * 2**i, max)`` (:78-79), BatchNorm(eps 1e-5, affine) + LeakyReLU(inplace) + conv bias
(:52-58), instantiates :class:`NexToU` and applies ``InitWeights_He(1e-2)`` (:88).

Inside an nnU-Net installation the base class is the real ``nnUNetTrainer`` (drop-in, see
INTEGRATION.md); without one it is the minimal stand-in of ``nextou_amd.harness``.
"""
from __future__ import annotations

from torch import nn

from ..network_architecture.NexToU import NexToU
from ..network_architecture.conv_blocks import InitWeights_He, convert_dim_to_conv_op, get_matching_batchnorm

try:
    from nnunetv2.training.nnUNetTrainer.nnUNetTrainer import nnUNetTrainer  # type: ignore
except ImportError:
    from ..harness import StandaloneTrainerBase as nnUNetTrainer


class nnUNetTrainer_NexToU(nnUNetTrainer):
    @staticmethod
    def build_network_architecture(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                   enable_deep_supervision: bool = True) -> nn.Module:
        num_stages = len(configuration_manager.conv_kernel_sizes)
        dim = len(configuration_manager.conv_kernel_sizes[0])
        conv_op = convert_dim_to_conv_op(dim)
        label_manager = plans_manager.get_label_manager(dataset_json)
        features = [min(configuration_manager.UNet_base_num_features * 2 ** i,
                        configuration_manager.unet_max_num_features) for i in range(num_stages)]
        model = NexToU(
            input_channels=num_input_channels,
            patch_size=configuration_manager.patch_size,
            n_stages=num_stages,
            features_per_stage=features,
            conv_op=conv_op,
            kernel_sizes=configuration_manager.conv_kernel_sizes,
            strides=configuration_manager.pool_op_kernel_sizes,
            n_conv_per_stage=configuration_manager.n_conv_per_stage_encoder,
            num_classes=label_manager.num_segmentation_heads,
            n_conv_per_stage_decoder=configuration_manager.n_conv_per_stage_decoder,
            conv_bias=True,
            norm_op=get_matching_batchnorm(conv_op),
            norm_op_kwargs={'eps': 1e-5, 'affine': True},
            dropout_op=None, dropout_op_kwargs=None,
            nonlin=nn.LeakyReLU, nonlin_kwargs={'inplace': True},
            deep_supervision=enable_deep_supervision,
        )
        model.apply(InitWeights_He(1e-2))
        return model
