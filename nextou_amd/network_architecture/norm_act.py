"""Fused (norm -> LeakyReLU) modules of the NexToU path, backed by K6 of libnextou_hip.so.

Every conv block on the path ends in ``norm -> nonlin``: the reference's ``BasicConv``
(torch_nn.py:84-90), ``FFN`` (NexToU_Encoder_Decoder.py:384-390), the graphers' ``fc1`` / ``fc2``
(:710-720, :833-842) and the conv stages' ``ConvDropoutNormReLU`` (:125-136, :281-298).  PyTorch-ROCm
runs them as MIOpen batch norm + ``leaky_relu`` kernels — 44 ms of the 327 ms cfg-2 step.  The classes
here are the stock ``nn.BatchNormNd`` / ``nn.InstanceNormNd`` with one extra attribute,
``negative_slope`` (1.0 = no activation), and a ``forward`` that calls
:func:`nextou_amd.graph_ops.norm_act`.  Parameters, buffers and ``state_dict`` keys are untouched, so
checkpoints interchange with the reference.

:func:`fuse_norm_act` converts a built model in place (class swap — no parameter is copied): every
``BatchNorm`` / ``InstanceNorm`` that sits in an ``nn.Sequential`` becomes its fused class and a
``LeakyReLU`` that directly follows it is absorbed (replaced by ``nn.Identity``); a convolution directly in front of it
runs without its bias, which the norm folds in (:class:`_ConvBiasFolded`).  ``NexToU.__init__`` applies it unless
``NEXTOU_FUSE_NORM_ACT=0`` (kept for A/B measurements).
"""
from __future__ import annotations

import os
import threading

import torch
from torch import nn

from .. import graph_ops
from .channel_pad import norm_input_is_padded, pad_image_channels, padded_conv_params

__all__ = ["BatchNormAct1d", "BatchNormAct2d", "BatchNormAct3d", "InstanceNormAct1d", "InstanceNormAct2d",
           "InstanceNormAct3d", "ConvBiasFolded1d", "ConvBiasFolded2d", "ConvBiasFolded3d", "ConvOwnBias2d",
           "ConvOwnBias3d", "ConvTransposeOwnBias2d", "ConvTransposeOwnBias3d", "fuse_norm_act", "fusion_enabled", "up_conv_cat", "DeferredCounters", "StemSequential", "fuse_stem_block",
           "attach_deferred_counters"]


def _pre_bias(norm: nn.Module):
    """bias of the (bias-free running) convolution in front of ``norm`` — see :class:`_ConvBiasFolded`."""
    src = getattr(norm, "_pre_bias_src", None)
    return None if src is None else src[0].bias


class _ConvBiasFolded:
    """A convolution directly followed by a fused norm: its bias is not added to the output tensor.

    ``norm(conv(x) + b)`` does not depend on ``b`` under batch / instance statistics (the mean absorbs it), and with
    running statistics it is one more term of the shift.  The conv therefore runs bias-free, the fused norm reads
    ``conv.bias`` (K6's ``pre_bias``) and autograd never launches the bias-gradient reduction over the conv output —
    a full read of every activation gradient of the network (5.4 ms of the cfg-2 step).  ``bias`` stays a parameter
    of this module (same ``state_dict`` key); its gradient comes from the norm's backward (exactly 0 in training).
    """

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        weight, _ = padded_conv_params(self, x, with_bias=False)      # channel_pad.py: zero-padded 33 -> 40 / 66 -> 72
        x, weight = pad_image_channels(self, x, weight)
        if graph_ops.pointwise_eligible(self, x, weight):
            return graph_ops.pointwise_conv(x, weight, None, self.groups)      # K7: 1x1 convolutions as own f32-MFMA GEMMs
        if graph_ops.rows_gemm_eligible(self, x, weight):
            return graph_ops.rows_gemm(x, weight)                              # small volumes: the BLAS GEMM over the (points, channels) view
        if graph_ops.grouped_cm_gemm_eligible(self, x, weight):
            return graph_ops.grouped_cm_gemm(x, weight, self.groups)           # Pool MRConv's grouped 1x1 conv on channel-major rows
        if x.requires_grad and graph_ops.dgrad_as_forward_eligible(self, x):
            return graph_ops.conv_dgrad_as_forward(x, weight, self.padding)    # backward-data as a forward convolution
        if not isinstance(self.padding, str) and self.padding_mode == "zeros" and \
                graph_ops.flat_depth_eligible(x, weight, self.stride, self.padding, self.dilation):
            # kernel [1,k,k] on a channels-last volume: the 2-D convolution of the (B*D, C, H, W) view (grouped or not)
            y = torch.nn.functional.conv2d(graph_ops.flat_depth(x), weight.squeeze(2), None, self.stride[1:], self.padding[1:],
                                           self.dilation[1:], self.groups)
            return graph_ops.unflat_depth(y, x.shape[0], x.shape[2])
        if (x.requires_grad or weight.requires_grad) and graph_ops.depth_unrolled_grads_eligible(self, x, weight):
            return graph_ops.conv_depth_unrolled_grads(x, weight, self.stride, self.padding)   # strided [3,3,3]: 2-D gradients
        return self._conv_forward(x, weight, None)


class ConvBiasFolded1d(_ConvBiasFolded, nn.Conv1d):
    pass


class ConvBiasFolded2d(_ConvBiasFolded, nn.Conv2d):
    pass


class ConvBiasFolded3d(_ConvBiasFolded, nn.Conv3d):
    pass


_FOLDED = {nn.Conv1d: ConvBiasFolded1d, nn.Conv2d: ConvBiasFolded2d, nn.Conv3d: ConvBiasFolded3d}


class _ConvOwnBias:
    """A biased convolution that is NOT followed by a norm (segmentation heads, transposed convolutions): same
    forward, but the bias gradient is K6's per-channel sum (graph_ops.conv_own_bias_grad) — ATen's generic reduction
    needs 5.5 ms for the 727 MB channels-last gradient of cfg 2's full-resolution transposed convolution."""

    def _own(self, x):
        return x.is_cuda and self.bias is not None and not isinstance(self.padding, str) and \
            getattr(self, "padding_mode", "zeros") == "zeros"

    def forward(self, x: torch.Tensor, *args) -> torch.Tensor:
        if args or (not self._own(x) and getattr(self, "_pad_spec", None) is None):
            return super().forward(x, *args)
        n = len(self.stride)
        if self.transposed:
            out_pad = self._output_padding(x, None, self.stride, self.padding, self.kernel_size, n, self.dilation)
        else:
            out_pad = (0,) * n
        weight, bias = padded_conv_params(self, x, with_bias=True)
        if self._own(x) and graph_ops.head_rows_eligible(self, x, weight):
            return graph_ops.head_rows(x, weight, bias)            # K8: the segmentation heads on own kernels, forward and backward
        if self._own(x):
            return graph_ops.conv_own_bias_grad(x, weight, bias, self.stride, self.padding, self.dilation,
                                                self.transposed, out_pad, self.groups)
        # CPU checker path of a padded module: the same convolution through ATen's autograd
        return torch.convolution(x, weight, bias, self.stride, self.padding, self.dilation, self.transposed, out_pad,
                                 self.groups)


def up_conv_cat(up: nn.Module, x: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """The decoder's ``torch.cat((up(x), skip), 1)``.  For an own-bias (transposed) convolution on dense channels-last fp32 tensors the
    convolution runs WITHOUT its bias and one pass writes ``[conv(x) + bias, skip]`` (graph_ops.cat_bias): ATen's separate bias-add pass
    over the up-sampled tensor disappears.  Anything else: the plain concatenation."""
    if isinstance(up, _ConvOwnBias) and up._own(x) and not torch.is_autocast_enabled("cuda") and x.dtype == torch.float32:
        n = len(up.stride)
        out_pad = up._output_padding(x, None, up.stride, up.padding, up.kernel_size, n, up.dilation) if up.transposed else (0,) * n
        weight, bias = padded_conv_params(up, x, with_bias=True)
        if up.transposed and graph_ops.upconv_cat_eligible(x, weight, bias, skip, up.stride, up.padding, up.dilation, out_pad, up.groups,
                                                           up.kernel_size):
            # kernel == stride: the transposed convolution is a K7 GEMM + a pixel shuffle that rides on the concatenation pass (round 5)
            return graph_ops.upconv_cat(x, weight, bias, skip, up.stride)
        y = graph_ops.conv_own_bias_grad(x, weight, None, up.stride, up.padding, up.dilation, up.transposed, out_pad, up.groups)
        if graph_ops.cat_bias_eligible(y, bias, skip):
            return graph_ops.cat_bias(y, bias, skip)
        if bias is not None:
            y = y + bias.view(1, -1, *([1] * n))
        return torch.cat((y, skip), 1)
    return torch.cat((up(x), skip), 1)


class ConvOwnBias2d(_ConvOwnBias, nn.Conv2d):
    pass


class ConvOwnBias3d(_ConvOwnBias, nn.Conv3d):
    pass


class ConvTransposeOwnBias2d(_ConvOwnBias, nn.ConvTranspose2d):
    pass


class ConvTransposeOwnBias3d(_ConvOwnBias, nn.ConvTranspose3d):
    pass


_OWN_BIAS = {nn.Conv2d: ConvOwnBias2d, nn.Conv3d: ConvOwnBias3d, nn.ConvTranspose2d: ConvTransposeOwnBias2d,
             nn.ConvTranspose3d: ConvTransposeOwnBias3d}


def _verify_batch_size(x: torch.Tensor) -> None:
    """torch.nn.functional.batch_norm's guard for batch statistics (the reference's modules raise the same ValueError)."""
    if x.numel() // max(x.shape[1], 1) == 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (x.size(),))


def _verify_spatial_size(x: torch.Tensor) -> None:
    """torch.nn.functional.instance_norm's guard (statistics over the spatial elements of one sample)."""
    n = 1
    for v in x.shape[2:]:
        n *= int(v)
    if n == 1:
        raise ValueError("Expected more than 1 spatial element when training, got input size %s" % (x.size(),))


class DeferredCounters(threading.local):
    """(State per thread, like graph_ops.ZeroGradScope.)  ``num_batches_tracked += 1`` of every batch norm of a network as ONE multi-tensor launch per forward instead of one
    single-element kernel per norm (78 launches, 0.34 ms of the cfg-2 step — profiles/r05_aten_glue.md).  The network's forward
    runs inside ``with counters:``; norms whose momentum is a number (the counter is pure bookkeeping then, as in
    torch.nn.modules.batchnorm._BatchNorm.forward) hand their counter over instead of advancing it themselves, and leaving the
    block advances them all — also when the forward raises, so the counters never disagree with the norms that ran.  Norms called
    outside such a block, or with ``momentum=None`` (cumulative average: the factor needs the counter's value), keep the eager
    increment."""

    def __init__(self):
        self.active = False
        self._pending = []

    def __reduce__(self):            # deepcopy / pickle of a model: a fresh, idle instance (shared by the copy's norms through the memo)
        return (DeferredCounters, ())

    def defer(self, counter: torch.Tensor) -> None:
        self._pending.append(counter)

    def __enter__(self):
        self.active = True
        self._pending = []
        return self

    def __exit__(self, *exc):
        self.active = False
        pending, self._pending = self._pending, []
        times = {}
        for t in pending:            # a module called twice in one forward counts twice
            times[id(t)] = (t, times.get(id(t), (t, 0))[1] + 1)
        for n in sorted({v[1] for v in times.values()}):
            group = [t for t, k in times.values() if k == n]
            by_device = {}
            for t in group:
                by_device.setdefault(t.device, []).append(t)
            for tensors in by_device.values():
                torch._foreach_add_(tensors, n)
        return False


def attach_deferred_counters(root: nn.Module) -> "DeferredCounters":
    """Give every fused batch norm under ``root`` the same :class:`DeferredCounters` (held in a tuple: not a sub-module)."""
    counters = DeferredCounters()
    for m in root.modules():
        if isinstance(m, _BatchNormAct):
            m._counter_group = (counters,)
    return counters


class _BatchNormAct:
    negative_slope: float = 1.0

    def _step(self):
        """``(use_batch_stats, momentum factor, keep_running)`` of this call — the bookkeeping of
        torch.nn.modules.batchnorm._BatchNorm.forward (advances ``num_batches_tracked``); also called by the fused point-wise
        pipeline (graph_ops.pointwise_chain), which runs this norm inside its GEMMs."""
        factor = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            group = getattr(self, "_counter_group", None)
            if group is not None and group[0].active and self.momentum is not None:
                group[0].defer(self.num_batches_tracked)        # one multi-tensor add for the whole network (DeferredCounters)
            else:
                self.num_batches_tracked.add_(1)
                factor = 1.0 / float(self.num_batches_tracked) if self.momentum is None else self.momentum
        use_batch_stats = self.training or (self.running_mean is None and self.running_var is None)
        keep_running = (not self.training) or self.track_running_stats
        return use_batch_stats, factor, keep_running

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input_dim(x)
        if self.training:
            _verify_batch_size(x)
        use_batch_stats, factor, keep_running = self._step()
        return graph_ops.norm_act(x, self.weight, self.bias, self.running_mean if keep_running else None,
                                  self.running_var if keep_running else None, use_batch_stats, factor, self.eps,
                                  self.negative_slope, pre_bias=_pre_bias(self),
                                  pad_holder=self if norm_input_is_padded(self, x) else None)

    def extra_repr(self) -> str:
        return super().extra_repr() + ", negative_slope=%g" % self.negative_slope


class _InstanceNormAct:
    negative_slope: float = 1.0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input_dim(x)
        if self.track_running_stats:
            raise NotImplementedError("InstanceNormAct: track_running_stats=True is not on the NexToU path")
        if x.dim() == self._get_no_batch_dim():
            return self.forward(x.unsqueeze(0)).squeeze(0)
        _verify_spatial_size(x)
        return graph_ops.norm_act(x, self.weight, self.bias, None, None, True, 0.0, self.eps, self.negative_slope,
                                  instance=True, pre_bias=_pre_bias(self))

    def extra_repr(self) -> str:
        return super().extra_repr() + ", negative_slope=%g" % self.negative_slope


class BatchNormAct1d(_BatchNormAct, nn.BatchNorm1d):
    pass


class BatchNormAct2d(_BatchNormAct, nn.BatchNorm2d):
    pass


class BatchNormAct3d(_BatchNormAct, nn.BatchNorm3d):
    pass


class InstanceNormAct1d(_InstanceNormAct, nn.InstanceNorm1d):
    pass


class InstanceNormAct2d(_InstanceNormAct, nn.InstanceNorm2d):
    pass


class InstanceNormAct3d(_InstanceNormAct, nn.InstanceNorm3d):
    pass


_FUSED = {
    nn.BatchNorm1d: BatchNormAct1d, nn.BatchNorm2d: BatchNormAct2d, nn.BatchNorm3d: BatchNormAct3d,
    nn.InstanceNorm1d: InstanceNormAct1d, nn.InstanceNorm2d: InstanceNormAct2d, nn.InstanceNorm3d: InstanceNormAct3d,
}


class StemSequential(nn.Sequential):
    """``all_modules`` of the network's first ConvDropoutNormReLU (conv -> fused norm [-> Identity]): the same modules, the same
    state_dict keys; when the call qualifies (graph_ops.stem_block_eligible) the pair runs on K9 — csrc/stem_conv.hip, the
    convolution's output never stored — otherwise module by module as any nn.Sequential."""

    def forward(self, x):
        mods = list(self._modules.values())
        if len(mods) >= 2 and all(type(m) is nn.Identity for m in mods[2:]) and graph_ops.stem_block_eligible(mods[0], mods[1], x):
            y = graph_ops.stem_block(x, mods[0], mods[1])
            if y is not None:
                return y
        return super().forward(x)


def fuse_stem_block(model: nn.Module) -> bool:
    """Class swap of the first block's ``all_modules`` (3-D and 2-D alike; what qualifies is decided per call).  Returns whether the
    swap was made."""
    try:
        block = model.encoder.stages[0][0].convs[0]
    except (AttributeError, IndexError, TypeError, KeyError):
        return False
    seq = getattr(block, "all_modules", None)
    if type(seq) is not nn.Sequential or len(seq) < 2 or not isinstance(seq[0], (nn.Conv2d, nn.Conv3d)) or \
            not isinstance(seq[1], _BatchNormAct) or seq[0].in_channels != 1:
        return False
    seq.__class__ = StemSequential
    return True


def fusion_enabled() -> bool:
    return os.environ.get("NEXTOU_FUSE_NORM_ACT", "1") != "0"


def _convert(norm: nn.Module) -> bool:
    cls = _FUSED.get(type(norm))
    if cls is None:
        return False
    if isinstance(norm, nn.modules.instancenorm._InstanceNorm) and norm.track_running_stats:
        return False
    norm.__class__ = cls
    norm.negative_slope = 1.0
    return True


def norm_of(seq: nn.Sequential, name: str) -> nn.Module:
    return seq._modules[name]


def fuse_norm_act(root: nn.Module) -> int:
    """In-place conversion described in the module docstring; returns the number of norms converted."""
    converted = 0
    for m in list(root.modules()):
        if isinstance(m, nn.Sequential):
            names = list(m._modules)
            for i, name in enumerate(names):
                if not _convert(m._modules[name]):
                    continue
                converted += 1
                prev = m._modules[names[i - 1]] if i > 0 else None
                if type(prev) in _FOLDED and prev.bias is not None and getattr(norm_of(m, name), "_pre_bias_src", None) is None:
                    prev.__class__ = _FOLDED[type(prev)]
                    norm_of(m, name)._pre_bias_src = (prev,)       # a tuple: not registered as a sub-module
                if i + 1 < len(names) and type(m._modules[names[i + 1]]) is nn.LeakyReLU:
                    m._modules[name].negative_slope = float(m._modules[names[i + 1]].negative_slope)
                    m._modules[names[i + 1]] = nn.Identity()
    for m in list(root.modules()):  # after the Sequentials: blocks whose activation sits beside one (FFN)
        hook = getattr(m, "_absorb_activation", None)
        if callable(hook):
            hook()
    for m in list(root.modules()):  # what is left with a bias was not folded into a norm: own bias gradient
        if type(m) in _OWN_BIAS and m.bias is not None:
            m.__class__ = _OWN_BIAS[type(m)]
    return converted
