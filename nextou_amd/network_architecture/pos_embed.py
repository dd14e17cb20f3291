"""Sin-cos relative position bias tables (init-time only, float64 on the host).

Mirrors the *results* of the reference's ``pos_embed.py`` (reference
``network_architecture/pos_embed.py:22-40`` for the two public entry points,
``:48-123`` for the sin-cos tables) with one n-dimensional implementation
instead of separate 2-D / 3-D code paths.

Semantics that must be reproduced:

* the grid is a cube/square of side ``grid_size``; the flat point index is
  row-major ``p = (i * g + j) * g + k`` (3-D) / ``p = i * g + j`` (2-D);
* ``numpy.meshgrid`` default ``'xy'`` indexing swaps the first two axes, so
  the first channel block encodes ``j``, the second ``i`` and (3-D) the third
  ``k`` (reference ``:55-56, 72-75``);
* every axis gets ``embed_dim // dim`` channels, ``[sin | cos]`` halves with
  frequencies ``10000 ** (-d / (D/2))`` (reference ``:107-123``);
* ``relative_pos = 2 * P @ P.T / embed_dim`` in float64 (reference ``:29-30``).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "get_1d_sincos_pos_embed_from_grid",
    "get_2d_sincos_pos_embed",
    "get_3d_sincos_pos_embed",
    "get_2d_relative_pos_embed",
    "get_3d_relative_pos_embed",
    "get_nd_relative_pos_embed",
]


def get_1d_sincos_pos_embed_from_grid(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """(M,) positions -> (M, embed_dim) ``[sin | cos]`` table, float64."""
    if embed_dim % 2:
        raise AssertionError("embed_dim per axis must be even, got %d" % embed_dim)
    half = embed_dim // 2
    freq = np.arange(half, dtype=np.float64)
    freq /= embed_dim / 2.0
    freq = 1.0 / 10000 ** freq
    phase = np.einsum("m,d->md", np.asarray(pos).reshape(-1), freq)
    return np.concatenate([np.sin(phase), np.cos(phase)], axis=1)


def _axis_coordinates(dim: int, g: int):
    """Per-axis float32 coordinate of every flat grid point, in the channel-block order."""
    ar = np.arange(g, dtype=np.float32)
    if dim == 2:
        i, j = np.meshgrid(ar, ar, indexing="ij")
        return [j.reshape(-1), i.reshape(-1)]
    if dim == 3:
        i, j, k = np.meshgrid(ar, ar, ar, indexing="ij")
        return [j.reshape(-1), i.reshape(-1), k.reshape(-1)]
    raise ValueError("only 2-D and 3-D grids are supported, got dim=%d" % dim)


def _nd_sincos(embed_dim: int, grid_size: int, dim: int) -> np.ndarray:
    if embed_dim % dim:
        raise AssertionError("embed_dim %d not divisible by %d" % (embed_dim, dim))
    per_axis = embed_dim // dim
    blocks = [get_1d_sincos_pos_embed_from_grid(per_axis, c) for c in _axis_coordinates(dim, grid_size)]
    return np.concatenate(blocks, axis=1)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    emb = _nd_sincos(embed_dim, grid_size, 2)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def get_3d_sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    emb = _nd_sincos(embed_dim, grid_size, 3)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def get_nd_relative_pos_embed(embed_dim: int, grid_size: int, dim: int) -> np.ndarray:
    """(g**dim, g**dim) float64 matrix ``2 P P^T / embed_dim``."""
    table = _nd_sincos(embed_dim, grid_size, dim)
    return 2 * np.matmul(table, table.transpose()) / table.shape[1]


def get_2d_relative_pos_embed(embed_dim: int, grid_size: int) -> np.ndarray:
    return get_nd_relative_pos_embed(embed_dim, grid_size, 2)


def get_3d_relative_pos_embed(embed_dim: int, grid_size: int) -> np.ndarray:
    return get_nd_relative_pos_embed(embed_dim, grid_size, 3)
