"""Memory-layout policy of the conv stages: which resolution stages run channels-last (N,D,H,W,C).

MIOpen's CK convolutions are NDHWC kernels; fed NCDHW tensors, every convolution is wrapped in
``batched_transpose`` launches (35 ms of the cfg-2 step, profiles/r01_cfg2_step_kernel_trace_k6.md).  At the
full-resolution stages that wrapper costs as much as the convolution itself (tools/conv_probe.py: 33 -> 33 channels
at 64x224x192, forward 6.05 ms NCDHW vs 3.33 ms NDHWC), so the plain conv stages (no graph blocks) of 3-D models keep
their activations channels-last from the stage input to the stage output; the graph stages need (B, C, N) rows and
stay NCDHW.  What made channels-last unusable on stock PyTorch-ROCm — MIOpen's NHWC batch norm and ATen's bias-gradient
reduction (170 ms on the two stages, tools/layout_probe.py) — is K6's job here (csrc/norm_act.hip).

Round 2: the graph stages follow.  Their 3x3x3 convolutions are CK NDHWC kernels too (7.3 ms of ``batched_transpose``
per cfg-2 step around them), and what pinned them to NCDHW — the graph kernels' channel-major rows (B', C, N) — is handled
where the data has to move anyway: the window shift / partition / reverse and the query max-pool / unpool are HIP kernels
(csrc/layout_ops.hip) that read or write the channels-last volume directly, and K6 has column-blocked kernels for rows
wider than 256 channels.  So with ``auto`` every stage of a 3-D model is channels-last from the network input to the logits.

Round 5: the FILTERS of those convolutions are stored channels-last too (:func:`filters_to_channels_last`): the library's
NDHWC kernels take (K, Z, Y, X, C) filters, and with the parameters stored (K, C, Z, Y, X)-contiguous every step paid one
conversion of each filter in the forward, one in the backward and one more for the weight gradient on its way into ``.grad``
(AccumulateGrad copies a gradient whose strides differ from the parameter's) — ~100 small copy kernels, 0.3 ms of the cfg-2
step (profiles/r05_aten_glue.md).  Shapes, values and the state_dict are unchanged (``load_state_dict`` copies by value, ``.to()``
keeps the strides); ``NEXTOU_CHANNELS_LAST_FILTERS=0`` keeps them contiguous (A/B).

``NEXTOU_CHANNELS_LAST_STAGES``: ``auto`` (default: all stages), ``plain`` (round 1: only the stages without graph blocks),
``none``, or a comma list of stage indices.  Under reduced-precision autocast the policy applies only together with the
internal channel padding (:func:`layout_policy_applies`).
"""
from __future__ import annotations

import os
from typing import FrozenSet

import torch
from torch import nn


def channels_last_stages(conv_op, n_plain_conv_stages: int, n_stages: int = None) -> FrozenSet[int]:
    spec = os.environ.get("NEXTOU_CHANNELS_LAST_STAGES", "auto").strip().lower()
    if spec in ("none", "", "0x"):
        return frozenset()
    if spec == "plain" or (spec == "auto" and n_stages is None):
        return frozenset(range(n_plain_conv_stages)) if conv_op is nn.Conv3d else frozenset()
    if spec == "auto":
        return frozenset(range(n_stages)) if conv_op is nn.Conv3d else frozenset()
    return frozenset(int(t) for t in spec.split(",") if t.strip() != "")


def is_channels_last_volume(x: torch.Tensor) -> bool:
    """True for a floating-point device tensor (B,C,*spatial), C > 1, stored (B,*spatial,C)-contiguous: what the fused
    window / pool kernels of the graph blocks take (fp32 as it is; bf16 / fp16 autocast tensors are widened on the way in —
    the graph kernels always compute in fp32)."""
    if not x.is_cuda or not x.is_floating_point() or x.dtype == torch.float64 or x.dim() not in (4, 5) or x.shape[1] == 1:
        return False
    mf = torch.channels_last if x.dim() == 4 else torch.channels_last_3d
    return x.is_contiguous(memory_format=mf) and not x.is_contiguous()


def to_channels_last(x: torch.Tensor) -> torch.Tensor:
    """channels_last / channels_last_3d view or copy of (B,C,*spatial).  A single-channel tensor is the same bytes in
    both layouts; it is re-strided (stride 1 on the channel axis) so that the convolution that consumes it picks the
    channels-last kernels and produces a channels-last output."""
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}[x.dim()]
    if x.shape[1] == 1:
        x = x.contiguous()
        strides = list(x.stride())
        strides[1] = 1
        return x.as_strided(x.shape, strides)
    return x.contiguous(memory_format=mf)


def runs_in_fp32(x: torch.Tensor) -> bool:
    """True when the convolutions fed by ``x`` will execute in fp32: an fp32 tensor outside reduced-precision autocast."""
    if x.dtype != torch.float32:
        return False
    if x.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") != torch.float32:
        return False
    return True


def layout_policy_applies(x: torch.Tensor, reduced_precision_ok=None) -> bool:
    """Whether the channels-last stages and the internal channel padding apply to the convolutions fed by ``x``.
    ``reduced_precision_ok``: what the MODEL says about itself (``NexToU.__init__``: its plain stages really carry channel
    counts that are multiples of 8, padded or native) — ``None`` falls back to the environment-level switch.

    fp32: always.  Reduced precision (bf16 / fp16 autocast): only together with the channel padding.  Round 1 measured
    NDHWC under bf16 autocast as a 125 ms *loss* on cfg 2 (309 vs 185 ms, profiles/r01_bf16_regression_ab.md) and fenced the
    policy off; the round-2 traces say why — at 33 / 66 channels MIOpen's bf16 NDHWC solvers are slow, and in NCDHW it falls
    back to im2col + GEMM (32 ms of ``Col2Im3dU`` per step) — and that with 40 / 72 channels the same NDHWC path is the fast
    one: 189.1 ms (NCDHW) -> 111.1 ms per step (profiles/r02_bf16_ndhwc_trace.md).  ``NEXTOU_REDUCED_PRECISION_LAYOUT=ncdhw``
    restores the round-1 behaviour for A/B runs.
    """
    if runs_in_fp32(x):
        return True
    if not x.is_floating_point():
        return False
    if os.environ.get("NEXTOU_REDUCED_PRECISION_LAYOUT", "auto").strip().lower() == "ncdhw":
        return False
    from .channel_pad import pad_multiple
    if reduced_precision_ok is not None:
        # a model whose padding could not be applied (conv_bias=False, another norm class) runs 33 / 66 channels;
        # NDHWC under bf16 at those counts is the 309 vs 185 ms regression round 1 fenced off.  NEXTOU_PAD_CHANNELS=0 at call
        # time still switches padding + NDHWC off for a built model (A/B runs).
        return bool(reduced_precision_ok) and pad_multiple() > 0
    return pad_multiple() > 0


def set_stage_layout(x: torch.Tensor, channels_last: bool, reduced_precision_ok=None) -> torch.Tensor:
    """Layout conversion at a stage boundary; only device tensors are ever moved (the CPU checker path is NCDHW).
    Encoder and decoder call this with the same stage set in the same forward, so they agree on every skip."""
    if not x.is_cuda:
        return x
    return to_channels_last(x) if (channels_last and layout_policy_applies(x, reduced_precision_ok)) else x.contiguous()


def filters_to_channels_last(root: nn.Module) -> int:
    """Re-stride, in place, the weight of every 3-D (4-D: 2-D) convolution / transposed convolution under ``root`` whose kernel is
    larger than 1 to channels_last_3d (channels_last) memory; returns the number of parameters converted.  Kernel-1 filters are
    the same bytes in both layouts and stay as they are."""
    if os.environ.get("NEXTOU_CHANNELS_LAST_FILTERS", "1") == "0":
        return 0
    n = 0
    for m in root.modules():
        if not isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            continue
        w = m.weight
        if w is None or not w.is_floating_point() or all(int(k) == 1 for k in w.shape[2:]):
            continue
        mf = torch.channels_last if w.dim() == 4 else torch.channels_last_3d
        if not w.is_contiguous(memory_format=mf):
            w.data = w.data.contiguous(memory_format=mf)
            n += 1
    return n
