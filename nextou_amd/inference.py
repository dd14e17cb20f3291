"""Sliding-window inference for NexToU: the step after training in nnU-Net.

What the reference itself contributes to inference is small and is mirrored exactly: ``NexToU_Decoder.forward``
returns the single full-resolution tensor when ``decoder.deep_supervision`` is off
(NexToU_Encoder_Decoder.py:333-337), the graph blocks assert that the input equals ``patch_size``
(:772,777), and the ``*_NoMirroring`` trainers switch the mirroring test-time augmentation off
(nnUNetTrainer_NexToU_NoMirroring.py:5-10).  The tiling itself lives in nnU-Net v2.0
(``nnunetv2.inference.sliding_window_prediction`` / ``predict_from_raw_data``, not under /root/reference);
this module restates its published algorithm — tile grid from a relative step, Gaussian importance map
(sigma = patch/8), mirror TTA over all axis subsets, accumulate / normalise — so the own harness can run the path.
**Parity unpinned** for this file (no reference test or golden vector covers it); its tests check the
algorithm's invariants.

MI355X-first difference: tiles are *batched*.  A NexToU forward at a fixed patch size is per-sample independent in
eval mode (BatchNorm uses running statistics, windows and kNN graphs never cross samples), and 288 GB of HBM holds
many eval-mode patches, so the mirrored copies of a tile (8 in 3-D) and neighbouring tiles go through the network as one
batch instead of one launch train per copy.
"""
from __future__ import annotations

import itertools
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ["compute_gaussian", "compute_steps_for_sliding_window", "mirror_axis_subsets", "predict_sliding_window"]


def compute_steps_for_sliding_window(image_size: Sequence[int], tile_size: Sequence[int],
                                     tile_step_size: float) -> List[List[int]]:
    """Tile origins per axis: steps of at most ``tile_step_size * tile`` voxels, first tile at 0, last flush with
    the image end (nnU-Net v2.0 ``compute_steps_for_sliding_window``)."""
    assert all(i >= t for i, t in zip(image_size, tile_size)), "image must be at least as large as the patch"
    assert 0 < tile_step_size <= 1, "tile_step_size must be in (0, 1]"
    steps = []
    for size, tile in zip(image_size, tile_size):
        target = tile * tile_step_size
        num = int(np.ceil((size - tile) / target)) + 1
        last = size - tile
        actual = last / (num - 1) if num > 1 else 0.0
        steps.append([int(np.round(actual * i)) for i in range(num)])
    return steps


def compute_gaussian(tile_size: Sequence[int], sigma_scale: float = 1. / 8, dtype=torch.float32,
                     device=torch.device("cpu")) -> torch.Tensor:
    """Importance map of a tile: a unit impulse at the centre blurred with sigma = tile * sigma_scale (scipy's
    ``gaussian_filter``: truncated at 4 sigma, zero padding), scaled to max 1, zeros lifted to the smallest non-zero
    value.  The filter is separable, so the map is the outer product of one 1-D response per axis."""
    from scipy.ndimage import gaussian_filter1d
    axes = []
    for n in tile_size:
        e = np.zeros(n, dtype=np.float64)
        e[n // 2] = 1.0
        axes.append(gaussian_filter1d(e, n * sigma_scale, mode="constant", cval=0.0))
    g = axes[0]
    for a in axes[1:]:
        g = np.multiply.outer(g, a)
    g = g / g.max()
    g[g == 0] = g[g != 0].min()
    return torch.from_numpy(g).to(device=device, dtype=dtype)


def mirror_axis_subsets(mirror_axes: Optional[Iterable[int]]) -> List[Tuple[int, ...]]:
    """() plus every non-empty subset of the spatial axes to mirror (nnU-Net's ``_internal_maybe_mirror_and_predict``
    enumerates the same 2^n combinations); ``None`` / empty = no test-time augmentation."""
    axes = tuple(sorted(set(mirror_axes))) if mirror_axes else ()
    return [c for r in range(len(axes) + 1) for c in itertools.combinations(axes, r)]


@torch.no_grad()
def predict_sliding_window(network: torch.nn.Module, image: torch.Tensor, patch_size: Sequence[int],
                           tile_step_size: float = 0.5, use_gaussian: bool = True,
                           mirror_axes: Optional[Iterable[int]] = None, batch_size: int = 8,
                           autocast_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Logits ``(num_classes, *spatial)`` (float32) of one pre-processed case ``image (C_in, *spatial)``.

    ``mirror_axes`` are spatial axes (0 = first spatial axis) as in nnU-Net's
    ``inference_allowed_mirroring_axes``; ``batch_size`` = network forwards batched together (tiles x mirror copies).
    Images smaller than the patch are zero-padded symmetrically and the result is cropped back.  The network is
    put in eval mode with deep supervision off for the call and restored afterwards.
    """
    dim = len(patch_size)
    assert image.dim() == dim + 1, "image must be (C, *spatial) with %d spatial axes" % dim
    decoder = getattr(network, "decoder", None)
    was_training, had_ds = network.training, getattr(decoder, "deep_supervision", None)
    network.eval()
    if had_ds is not None:
        decoder.deep_supervision = False
    try:
        spatial = tuple(image.shape[1:])
        pad = [max(p - s, 0) for p, s in zip(patch_size, spatial)]
        lo = [p // 2 for p in pad]
        if any(pad):
            widths = []
            for l, p in zip(reversed(lo), reversed(pad)):
                widths += [l, p - l]
            image = torch.nn.functional.pad(image, widths)
        padded = tuple(image.shape[1:])
        steps = compute_steps_for_sliding_window(padded, patch_size, tile_step_size)
        origins = list(itertools.product(*steps))
        flips = mirror_axis_subsets(mirror_axes)
        device = image.device
        weight = compute_gaussian(patch_size, device=device) if use_gaussian and len(origins) > 1 else \
            torch.ones(tuple(patch_size), device=device)
        logits = count = None
        jobs = [(o, f) for o in origins for f in flips]
        for start in range(0, len(jobs), max(1, batch_size)):
            chunk = jobs[start:start + max(1, batch_size)]
            tiles = []
            for origin, flip in chunk:
                sl = (slice(None),) + tuple(slice(o, o + p) for o, p in zip(origin, patch_size))
                t = image[sl]
                tiles.append(torch.flip(t, [a + 1 for a in flip]) if flip else t)
            batch = torch.stack(tiles).contiguous()
            if autocast_dtype is not None:
                with torch.autocast(device_type=device.type, dtype=autocast_dtype):
                    out = network(batch)
            else:
                out = network(batch)
            out = out.float()
            if logits is None:
                logits = torch.zeros((out.shape[1],) + padded, dtype=torch.float32, device=device)
                count = torch.zeros(padded, dtype=torch.float32, device=device)
            for (origin, flip), o in zip(chunk, out):
                if flip:
                    o = torch.flip(o, [a + 1 for a in flip])
                sl = tuple(slice(a, a + p) for a, p in zip(origin, patch_size))
                logits[(slice(None),) + sl] += o * (weight / len(flips))
                if flip == flips[0]:
                    count[sl] += weight
        logits /= count
        if any(pad):
            logits = logits[(slice(None),) + tuple(slice(l, l + s) for l, s in zip(lo, spatial))]
        return logits
    finally:
        network.train(was_training)
        if had_ds is not None:
            decoder.deep_supervision = had_ds
