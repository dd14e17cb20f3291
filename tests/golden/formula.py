"""Deterministic, formula-generated tensors shared by the golden generator and the tests.

Weights and inputs of the fixtures are *recomputed* from (name, shape, seed) on both sides instead
of being stored: ``torch.manual_seed`` streams are not a contract, numpy's PCG64 bit stream is.
"""
from __future__ import annotations

import contextlib
import zlib

import numpy as np
import torch


def _rng(tag: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(tag.encode()), seed]))


def gaussian(tag: str, shape, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((_rng(tag, seed).standard_normal(tuple(shape)) * scale).astype(np.float32))


def fill_module_(module: torch.nn.Module, seed: int = 0) -> None:
    """He-style formula weights keyed by the ``state_dict`` name (identical key grammar on the
    reference and on nextou_amd).  ``relative_pos`` tables are left as built."""
    seen = {}
    with torch.no_grad():
        for name, p in list(module.named_parameters(remove_duplicate=False)) + \
                list(module.named_buffers(remove_duplicate=False)):
            if name.endswith("relative_pos") or name.endswith("num_batches_tracked"):
                continue
            if id(p) in seen:  # aliases (decoder.encoder.*, all_modules.*) share storage
                continue
            seen[id(p)] = name
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "weight" and p.dim() >= 3:          # conv / transposed conv
                fan_in = int(np.prod(p.shape[1:]))
                v = gaussian(name, p.shape, seed, scale=float(np.sqrt(2.0 / fan_in)))
            elif leaf == "weight":                          # norm scale
                v = 1.0 + gaussian(name, p.shape, seed, scale=0.1)
            elif leaf == "bias":
                v = gaussian(name, p.shape, seed, scale=0.05)
            elif leaf == "running_mean":
                v = gaussian(name, p.shape, seed, scale=0.1)
            elif leaf == "running_var":
                v = 1.0 + gaussian(name, p.shape, seed, scale=0.1).abs()
            else:
                raise KeyError("no formula for %s" % name)
            p.copy_(v)


def blob_labels(shape, n_classes: int, n_seeds: int = 40, seed: int = 4321) -> np.ndarray:
    """BTCV-style synthetic label volume: nearest-seed Voronoi cells with a background shell
    (SURVEY.md §8d cfg 4) — i.i.d. labels would make ~93 % of the voxels critical."""
    rng = _rng("blob_labels", seed)
    dim = len(shape)
    pts = rng.random((n_seeds, dim)) * np.asarray(shape)
    cls = rng.integers(1, n_classes, size=n_seeds)
    grids = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing="ij")
    coords = np.stack(grids, -1).reshape(-1, dim)
    best = np.full(coords.shape[0], np.inf, dtype=np.float32)
    lab = np.zeros(coords.shape[0], dtype=np.int64)
    for p, c in zip(pts, cls):
        d = ((coords - p.astype(np.float32)) ** 2).sum(1)
        take = d < best
        best[take] = d[take]
        lab[take] = c
    centre = (np.asarray(shape, dtype=np.float32) - 1) / 2
    r = np.sqrt((((coords - centre) / (np.asarray(shape, dtype=np.float32) / 2)) ** 2).sum(1))
    lab[r > 0.9] = 0
    return lab.reshape(shape)


def coherent_logits(tag: str, n_classes: int, shape, batch: int = 2):
    """(logits (B,L,*shape) float32, target (B,1,*shape) float32): gaussian logits with a per-class bias from blob
    labels, so that the arg-max label map is spatially coherent, and an independent blob label map as the target."""
    shape = tuple(shape)
    logits = gaussian(tag + ".logits", (batch, n_classes) + shape, scale=2.0)
    for b in range(batch):
        lab = torch.from_numpy(blob_labels(shape, n_classes, n_seeds=12, seed=100 + b))
        logits[b].scatter_add_(0, lab.unsqueeze(0), torch.full((1,) + shape, 3.0))
    target = torch.from_numpy(np.stack([blob_labels(shape, n_classes, n_seeds=12, seed=200 + b)
                                        for b in range(batch)])).unsqueeze(1).float()
    return logits, target


_CONV_FUNCTIONS = ("conv1d", "conv2d", "conv3d", "conv_transpose1d", "conv_transpose2d", "conv_transpose3d")


def near_tie_logits(tag: str, n_classes: int = 14, n_voxels: int = 60000):
    """Logits (1, L, V) whose two largest entries per voxel are planted 0 ... 2^-21 apart with the SMALLER one at the earlier class index:
    the band in which float32 softmax values coincide and ``argmax(softmax(x))`` (reference bti_loss.py:132-134) returns the first index
    instead of the largest logit.  Four magnitude regimes (logit scale 0.02 / 0.2 / 1 / 3), a quarter of the voxels each; gaps are
    uniform in [0, 2^-21] with one voxel in eight an exact tie or a gap of a few ulps.  Returns (logits, gap (V,) float32 = max - runner-up)."""
    rng = _rng(tag, 7)
    x = rng.standard_normal((n_voxels, n_classes))
    scale = np.repeat(np.array([0.02, 0.2, 1.0, 3.0]), -(-n_voxels // 4))[:n_voxels]
    x = (x * scale[:, None]).astype(np.float32)
    hi = x.argmax(1)
    hi = np.where(hi == 0, 1 + rng.integers(0, n_classes - 1, n_voxels), hi)      # the maximum must not sit at class 0
    lo = rng.integers(0, 1 << 30, n_voxels) % hi                                   # an earlier class
    m = np.abs(x).max(1).astype(np.float32) + np.float32(0.25) * scale.astype(np.float32)
    gap = (rng.random(n_voxels) * 2.0 ** -21).astype(np.float32)
    few = rng.integers(0, 8, n_voxels) == 0
    ulps = rng.integers(0, 4, n_voxels)
    v = (m - gap).astype(np.float32)
    w = m.copy()
    for k in range(1, 4):
        w = np.where(ulps >= k, np.nextafter(w, np.float32(-np.inf), dtype=np.float32), w)
    v = np.where(few, w, v).astype(np.float32)
    rows = np.arange(n_voxels)
    x[rows, hi] = m
    x[rows, lo] = v
    logits = torch.from_numpy(np.ascontiguousarray(x.T)).reshape(1, n_classes, n_voxels)
    return logits, torch.from_numpy((m - v).astype(np.float32))


@contextlib.contextmanager
def convs_in_float64():
    """While active every ``torch.nn.functional`` (transposed) convolution computes in float64 and rounds its result
    back to the input dtype: the convolution arithmetic becomes the same on every backend (oneDNN, MIOpen / CK,
    rocBLAS) up to one final rounding, so what is left of a CPU-vs-GPU difference is NOT the dense stages' library.
    Used by make_golden.py on the reference and by the equal-convolution parity tests on nextou_amd."""
    import torch.nn.functional as F

    def wrap(fn):
        def conv(input, weight, bias=None, *args, **kwargs):
            out = fn(input.double(), weight.double(), None if bias is None else bias.double(), *args, **kwargs)
            return out.to(input.dtype)
        return conv

    saved = {n: getattr(F, n) for n in _CONV_FUNCTIONS}
    for n, fn in saved.items():
        setattr(F, n, wrap(fn))
    try:
        yield
    finally:
        for n, fn in saved.items():
            setattr(F, n, fn)
