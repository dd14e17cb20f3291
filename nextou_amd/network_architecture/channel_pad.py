"""Internal channel padding of the plain conv stages: 33 -> 40 and 66 -> 72 channels inside the network, nowhere else.

MIOpen's CK implicit-GEMM kernels run the fp32 convolutions of cfg 2's full-resolution stages (33 and 66 channels) at
15-24 % of the fp32 MFMA peak while the matrix pipe is ~40 % busy: the kernels pad 33 / 66 to their own tile sizes
internally.  The same layers with the channel count zero-padded to a multiple of 8 *before* they reach MIOpen pick better
tiles and run 1.4-1.7x faster although they compute more (measured, profiles/r02_conv_evidence_padding_ab.md:
33 -> 33 at 64x224x192 fwd+dgrad+wgrad 12.3 ms -> 8.7 ms at 40 -> 40; 66 -> 66 at 64x112x96 28.2 ms -> 18.7 ms at 72 -> 72).
The convolutions themselves stay on PyTorch-ROCm, as BASELINE.json's north_star prescribes — this only changes the shapes
they are handed.

Mechanics.  Parameters and buffers keep the reference's shapes (``state_dict`` interchange, ``InitWeights_He``, DDP buckets
and optimizer state see 33 / 66).  On the forward pass a padded module builds its padded weight / bias / norm parameters on
the fly (``F.pad`` / ``cat`` of a few KB — autograd slices the gradients back; norm parameters are staged into a persistent
padded buffer by graph_ops.norm_act) and the activations between the padded
modules carry ``pad`` extra all-zero channels:

* an *entry* module (first convolution of stage 0; an up-convolution fed by a graph stage) decides per call: it pads its
  output iff its input is a device tensor the layout policy applies to (``layout.layout_policy_applies``; the CPU checker
  path stays un-padded), and records the decision in the model's shared :class:`PadRegime`;
* every other padded module follows the channel count it receives (real count = "not padded", padded count = "padded")
  when the two differ, and reads the shared regime when they do not: with feature counts like 12 / 24, where one plain
  stage needs padding and the next is a multiple already, the count alone cannot tell the regimes apart;
* *exit* modules (segmentation heads, the first convolution of the first graph stage) always emit the real channel count.

Zero channels stay exactly zero through conv (zero filters) -> batch norm (mean 0, variance 0, weight 1, bias 0 -> 0) ->
LeakyReLU, and receive exactly zero gradient (the next layer's columns for them are zero), so the function and its
gradients are those of the un-padded network up to the summation order inside the convolution kernels.
``NEXTOU_PAD_CHANNELS``: ``auto`` (default, multiple of 8), ``0`` / ``none`` (off), or the multiple to pad to.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .layout import layout_policy_applies, runs_in_fp32


_force_entry: Optional[bool] = None     # test hook: True / False overrides the entry modules' per-call decision


class force_padding:
    """``with force_padding(True):`` makes the entry modules pad whatever the device (CPU equivalence tests);
    ``force_padding(False)`` switches the padded regime off for a built model (A/B on the GPU)."""

    def __init__(self, flag: Optional[bool]):
        self.flag = flag

    def __enter__(self):
        global _force_entry
        self.prev, _force_entry = _force_entry, self.flag
        return self

    def __exit__(self, *exc):
        global _force_entry
        _force_entry = self.prev
        return False


def pad_multiple() -> int:
    spec = os.environ.get("NEXTOU_PAD_CHANNELS", "auto").strip().lower()
    if spec in ("0", "none", "off", ""):
        return 0
    return 8 if spec == "auto" else int(spec)


def padded(c: int, multiple: int) -> int:
    return -(-c // multiple) * multiple


@dataclass(frozen=True)
class ConvPad:
    in_blocks: Tuple[int, ...]      # real channel counts of the concatenated inputs (one block unless fed by a cat)
    multiple: int
    entry: bool = False             # decides whether the padded regime starts here
    exit: bool = False              # always emits the real output channel count
    pad_input: bool = True          # False: the input is never padded (network input, graph-stage features)
    image_channels_to: int = 0      # entry only: zero-pad the network input itself to this many channels when padding


class PadRegime:
    """Per-forward state shared by the padded modules of ONE model: whether the activations between them currently carry
    padding channels.  Written by the entry modules on every call, read by everything downstream."""

    __slots__ = ("on",)

    def __init__(self):
        self.on = False


def _pad_axis(t: torch.Tensor, axis: int, new: int, value: float = 0.0) -> torch.Tensor:
    n = new - t.shape[axis]
    if n == 0:
        return t
    return F.pad(t, [0, 0] * (t.dim() - 1 - axis) + [0, n], value=value)


def _pad_blocks(t: torch.Tensor, axis: int, blocks, multiple: int) -> torch.Tensor:
    if len(blocks) == 1:
        return _pad_axis(t, axis, padded(blocks[0], multiple))
    pieces = t.split(list(blocks), dim=axis)
    return torch.cat([_pad_axis(p, axis, padded(b, multiple)) for p, b in zip(pieces, blocks)], axis)


def conv_pad_plan(module: nn.Module, x: torch.Tensor):
    """-> (pad_in, pad_out) for this call of a convolution carrying a ``_pad_spec``."""
    spec: ConvPad = module._pad_spec
    real_in = sum(spec.in_blocks)
    padded_in = sum(padded(b, spec.multiple) for b in spec.in_blocks) if spec.pad_input else real_in
    cin = x.shape[1]
    if cin == real_in:
        pad_in = False
    elif cin == padded_in:
        pad_in = True
    else:
        raise RuntimeError("channel padding: %s received %d channels, expected %d (real) or %d (padded)"
                           % (type(module).__name__, cin, real_in, padded_in))
    regime: PadRegime = module._pad_regime
    if spec.entry:
        # (a module that carries a spec belongs to a model whose padding was applied: reduced precision is fine)
        on = regime.on = bool(x.is_cuda and layout_policy_applies(x, True)) if _force_entry is None else bool(_force_entry)
    elif padded_in != real_in:
        on = pad_in             # the input's channel count is unambiguous (also when a sub-module is run on its own)
    else:                       # real == padded count for this input (e.g. 24 channels): only the regime can tell
        on = regime.on if _force_entry is None else bool(_force_entry)
    pad_out = on and not spec.exit
    return pad_in, pad_out


def padded_conv_params(module: nn.Module, x: torch.Tensor, with_bias: bool):
    """(weight, bias) of a (transposed) convolution for this call: the module's own parameters, zero-padded along the
    input-channel axis when ``x`` arrives padded and along the output-channel axis when the output is to be padded."""
    spec: Optional[ConvPad] = getattr(module, "_pad_spec", None)
    bias = module.bias if with_bias else None
    if spec is None:
        return filter_layout_for(x, module.weight), bias
    pad_in, pad_out = conv_pad_plan(module, x)
    w = module.weight
    in_axis, out_axis = (0, 1) if module.transposed else (1, 0)
    if pad_in:
        w = _pad_blocks(w, in_axis, spec.in_blocks, spec.multiple)
    if pad_out:
        w = _pad_axis(w, out_axis, padded(w.shape[out_axis], spec.multiple))
        if bias is not None:
            bias = _pad_axis(bias, 0, w.shape[out_axis])
    return filter_layout_for(x, w), bias


def filter_layout_for(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """The filter in the layout this call's input asks for.  Filters with a kernel larger than 1 are STORED channels-last
    (layout.filters_to_channels_last), and the library takes the output's layout from the filter as well as from the input: a
    channels-last filter turns an NCDHW stage channels-last behind the layout policy's back.  Where the policy keeps a stage NCDHW —
    reduced precision without the channel padding, ``NEXTOU_REDUCED_PRECISION_LAYOUT=ncdhw`` (layout.layout_policy_applies: NDHWC at
    33 / 66 bf16 channels was round 1's 309-vs-185-ms regression) — the call gets the contiguous form: the per-call conversion that the
    stored layout saves on the default, all-channels-last path.  A channels-last input — dense NDHWC, or the single-channel image
    re-strided by layout.to_channels_last — has stride 1 on its channel axis.

    Reduced precision (bf16 / fp16 autocast) always gets the contiguous form: autocast's cast keeps a filter's strides, and the library's
    reduced-precision solvers for channels-last FILTERS were never part of a clean measured run — the one `bench.py --autocast-bf16` run of
    the tree that stored filters channels-last ended in a GPU memory fault (known issue), while contiguous filters are
    what every earlier bf16 / fp16 run and test used."""
    if w.dim() < 4 or not x.is_cuda or w.is_contiguous() or x.dim() != w.dim():
        return w
    if x.stride(1) == 1 and (runs_in_fp32(x) or os.environ.get("NEXTOU_REDUCED_PRECISION_FILTERS", "contiguous") == "stored"):
        # ("stored": the channels-last filter under autocast too — the configuration of round 5's faulting run, kept reachable for the
        # A/B that convicts or clears it, profiles/r06_bf16/)
        return w
    return w.contiguous()


def pad_image_channels(module: nn.Module, x: torch.Tensor, weight: torch.Tensor):
    """Entry convolution in the padded regime: the network input (1 channel at cfg 2) is itself zero-padded to 4 channels.
    MIOpen's weight-gradient kernel for a 1-channel input needs 7.8 ms at 64x224x192 (0.3 % of the MFMA peak); the same
    layer with 4 input channels 1.5 ms, and its forward 0.39 ms instead of 1.07 ms (profiles/r02_conv_evidence_padding_ab.md)."""
    spec: Optional[ConvPad] = getattr(module, "_pad_spec", None)
    if spec is None or not spec.entry or not spec.image_channels_to or weight.shape[0] == module.out_channels:
        return x, weight
    c, to = x.shape[1], spec.image_channels_to
    if c >= to:
        return x, weight
    mf = torch.channels_last_3d if x.dim() == 5 else torch.channels_last
    xp = x.new_zeros((x.shape[0], to) + tuple(x.shape[2:])).contiguous(memory_format=mf)
    xp[:, :c] = x
    return xp, _pad_axis(weight, 1, to)


def norm_input_is_padded(norm: nn.Module, x: torch.Tensor) -> bool:
    """True when ``x`` reaches a fused norm with its padding channels (then graph_ops.norm_act stages the module's real-length
    parameters and running statistics into the padded width, ``pad_holder=norm``)."""
    multiple = getattr(norm, "_pad_multiple", 0)
    c_real = norm.num_features
    if not multiple or x.shape[1] == c_real:
        return False
    c_pad = padded(c_real, multiple)
    if x.shape[1] != c_pad:
        raise RuntimeError("channel padding: norm over %d channels received %d (padded count is %d)" % (c_real, x.shape[1], c_pad))
    return True


def pad_plain_stage_channels(model: nn.Module, multiple: int) -> int:
    """Attach padding specs to the plain conv stages of a built (and norm-fused) NexToU; returns the number of modules
    marked.  Only 3-D models with at least one plain conv stage whose feature count is not a multiple already."""
    enc, dec = model.encoder, model.decoder
    n_conv = enc.n_conv_stages
    feats = list(enc.output_channels)
    if multiple <= 0 or enc.conv_op is not nn.Conv3d or n_conv == 0:
        return 0
    if all(f % multiple == 0 for f in feats[:n_conv]):
        return 0
    from . import norm_act as na
    plan = []           # (module, attribute, value): applied only if every module involved can carry it

    def mark_conv(conv, in_blocks, **kw):
        plan.append((conv, "_pad_spec", ConvPad(tuple(int(b) for b in in_blocks), multiple, **kw)))

    def mark_norm(block):
        plan.append((getattr(block, "norm", None), "_pad_multiple", multiple))

    # encoder: plain stages, then the first convolution of the first graph stage (exit)
    c_prev = None
    for s in range(n_conv):
        blocks = enc.stages[s][0].convs
        for i, blk in enumerate(blocks):
            if s == 0 and i == 0:
                mark_conv(blk.conv, [blk.conv.in_channels], entry=True, pad_input=False,
                          image_channels_to=4 if blk.conv.in_channels < 4 else 0)
            else:
                mark_conv(blk.conv, [c_prev])
            mark_norm(blk)
            c_prev = blk.conv.out_channels
    first_graph = enc.stages[n_conv][0][0].convs[0]
    mark_conv(first_graph.conv, [c_prev], exit=True)
    # decoder: stage j mirrors encoder level n_enc - (j + 2)
    n_enc = len(feats)
    for j in range(len(dec.stages)):
        level = n_enc - (j + 2)
        up = dec.transpconvs[j]
        if level < n_conv:                      # output feeds a plain stage: padded; input padded iff it comes from one
            from_plain = level + 1 < n_conv
            mark_conv(up, [up.in_channels], pad_input=from_plain, entry=not from_plain)
            blocks = dec.stages[j].convs
            mark_conv(blocks[0].conv, [feats[level], feats[level]])
            mark_norm(blocks[0])
            for blk in blocks[1:]:
                mark_conv(blk.conv, [feats[level]])
                mark_norm(blk)
            mark_conv(dec.seg_layers[j], [feats[level]], exit=True)
    capable = (na.ConvBiasFolded3d, na.ConvOwnBias3d, na.ConvTransposeOwnBias3d, na.BatchNormAct3d)
    if not all(isinstance(m, capable) for m, _, _ in plan):
        return 0        # e.g. conv_bias=False or another norm: those module classes do not know about padding
    regime = PadRegime()
    for m, attr, value in plan:
        setattr(m, attr, value)
        if attr == "_pad_spec":
            m._pad_regime = regime
    return len(plan)
