"""CPU oracle of the NexToU graph hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; nothing under ``nextou_amd/`` does (tests/test_layout.py enforces it).

Two checkers, both speaking the backend protocol of ``nextou_amd.graph_ops``:

* :class:`CanonicalBackend` — ``liboracle.so`` (``nextou_oracle.c``): plain-C restatement with the
  canonical arithmetic (fma chains, (dist, index) tie order).  The HIP kernels must match it bit for
  bit on kNN indices and to fp32 rounding elsewhere.
* :class:`TorchRefBackend` (``ref_ops.py``) — the reference's own op sequence in PyTorch-CPU
  (normalize -> matmul -> topk -> gather -> sub -> max -> interleave); validated against golden
  vectors generated from the reference itself; used as the timed CPU baseline ("port").

Parity pin: ``tests/golden/*.npz`` were produced by importing ``/root/reference`` in the build
container (``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` holds both checkers to
them.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_int64, c_void_p

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            from . import build as _build
            _build.build(verbose=False)
        h = ctypes.CDLL(LIB_PATH)
        h.oracle_knn_graph.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6
        h.oracle_pairwise_distance.argtypes = [c_void_p, c_void_p, c_void_p] + [c_int] * 6
        h.oracle_mr_fwd.argtypes = [c_void_p] * 6 + [c_int] * 7
        h.oracle_mr_bwd_arg.argtypes = [c_void_p] * 4 + [c_int] * 4
        h.oracle_mr_bwd.argtypes = [c_void_p] * 7 + [c_int] * 7
        h.oracle_argmax_labels.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int64]
        h.oracle_bti_critical.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 6
        for f in ("oracle_knn_graph", "oracle_pairwise_distance", "oracle_mr_fwd", "oracle_mr_bwd", "oracle_mr_bwd_arg",
                  "oracle_argmax_labels", "oracle_bti_critical"):
            getattr(h, f).restype = c_int
        _lib = h
    return _lib


def _p(t):
    return None if t is None else t.data_ptr()


def _in(t, dtype=None):
    """Input normalisation at the C boundary: liboracle.so reads raw pointers and assumes dense row-major
    (B, C, ...) storage of a fixed dtype.  A channels-last or otherwise strided tensor (e.g. the logits of a model whose
    full-resolution stage runs NDHWC) would be read in the wrong order without this — silently, in the CHECKER."""
    if t is None:
        return None
    t = t.detach()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous(memory_format=torch.contiguous_format)


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("oracle.%s rejected its arguments (rc=%d)" % (what, rc))


class CanonicalBackend:
    """Backend protocol of nextou_amd.graph_ops on CPU tensors, via liboracle.so."""

    name = "oracle-canonical"

    @staticmethod
    def knn_graph(x, y, relpos, k_total, algo=0, normalize=True):
        x, y, relpos = _in(x, torch.float32), _in(y, torch.float32), _in(relpos, torch.float32)
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        out = torch.empty((B, N, k_total), dtype=torch.int32)
        _ok(lib().oracle_knn_graph(_p(x), _p(y), _p(relpos), _p(out), B, C, N, M, k_total, int(normalize)),
            "knn_graph")
        return out

    @staticmethod
    def pairwise_distance(x, y, row_start, row_end):
        x, y = _in(x, torch.float32), _in(y, torch.float32)
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        out = torch.empty((B, row_end - row_start, M), dtype=torch.float32)
        _ok(lib().oracle_pairwise_distance(_p(x), _p(y), _p(out), B, C, N, M, row_start, row_end),
            "pairwise_distance")
        return out

    @staticmethod
    def edge_index(nn_idx, dilation):
        B, N, K = nn_idx.shape
        nn = nn_idx[:, :, ::dilation].to(torch.int64)
        center = torch.arange(N, dtype=torch.int64).view(1, N, 1).expand_as(nn)
        return torch.stack((nn, center), dim=0).contiguous()

    @staticmethod
    def mr_has_arg(B, C, N, M, K):
        return M <= 65536

    @staticmethod
    def mr_fwd(x, y, nn_idx, center, K, idx_step, want_arg=False):
        x, y = _in(x, torch.float32), _in(y, torch.float32)
        nn_idx, center = _in(nn_idx, torch.int32), _in(center, torch.int32)
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        out = torch.empty((B, 2 * C, N), dtype=torch.float32)
        arg = torch.empty((B, C, N), dtype=torch.int16) if (want_arg and center is None and M <= 65536) else None
        _ok(lib().oracle_mr_fwd(_p(x), _p(y), _p(nn_idx), _p(center), _p(out), _p(arg), B, C, N, M, K,
                                nn_idx.shape[2], idx_step), "mr_fwd")
        return out, arg

    @staticmethod
    def mr_bwd_arg(gout, arg, M, has_y):
        gout, arg = _in(gout, torch.float32), _in(arg)
        B, C, N = arg.shape
        dx = torch.empty((B, C, N), dtype=torch.float32)
        dy = torch.empty((B, C, M), dtype=torch.float32) if has_y else None
        _ok(lib().oracle_mr_bwd_arg(_p(gout), _p(arg), _p(dx), _p(dy), B, C, N, M), "mr_bwd_arg")
        return dx, dy

    @staticmethod
    def mr_bwd(gout, x, y, nn_idx, center, K, idx_step):
        gout, x, y = _in(gout, torch.float32), _in(x, torch.float32), _in(y, torch.float32)
        nn_idx, center = _in(nn_idx, torch.int32), _in(center, torch.int32)
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        dx = torch.empty_like(x)            # row-major: x was normalised above
        dy = None if y is None else torch.empty_like(y)
        _ok(lib().oracle_mr_bwd(_p(gout), _p(x), _p(y), _p(nn_idx), _p(center), _p(dx), _p(dy), B, C, N, M,
                                K, nn_idx.shape[2], idx_step), "mr_bwd")
        return dx, dy

    @staticmethod
    def gather_fwd(src, idx):
        B, C, M = src.shape
        _, N, K = idx.shape
        flat = idx.to(torch.int64).reshape(B, 1, N * K).expand(B, C, N * K)
        return src.gather(2, flat).reshape(B, C, N, K)

    @staticmethod
    def gather_bwd(gout, idx, M):
        B, C, N, K = gout.shape
        flat = idx.to(torch.int64).reshape(B, 1, N * K).expand(B, C, N * K)
        return torch.zeros((B, C, M), dtype=gout.dtype).scatter_add_(2, flat, gout.reshape(B, C, N * K))

    @staticmethod
    def argmax_labels(logits):
        logits = _in(logits, torch.float32)     # channels-last logits (NDHWC stages) become (B, L, V) row-major here
        B, L = logits.shape[:2]
        V = logits[0, 0].numel()
        out = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.uint8)
        _ok(lib().oracle_argmax_labels(_p(logits), _p(out), B, L, V), "argmax_labels")
        return out

    @staticmethod
    def bti_ce_fwd(logits, target, critical):
        """reference bti_loss.py:141-143 op sequence (float64 CE 'none' * critical, summed per sample)."""
        ce = torch.nn.functional.cross_entropy(logits.double(), target.long(), reduction='none')
        return (ce * critical.double()).flatten(1).sum(1)

    @staticmethod
    def bti_ce_bwd(logits, target, critical, scale):
        with torch.enable_grad():
            x = logits.detach().requires_grad_(True)
            ce = torch.nn.functional.cross_entropy(x.double(), target.long(), reduction='none')
            per_sample = (ce * critical.double()).flatten(1).sum(1)
            (grad,) = torch.autograd.grad((per_sample * scale.reshape(-1)).sum(), x)
        return grad

    @staticmethod
    def bti_critical(labels, lut_a, lut_c, connectivity, min_thick):
        labels, lut_a, lut_c = _in(labels, torch.uint8), _in(lut_a), _in(lut_c)
        if labels.dim() == 3:
            (B, H, W), D = labels.shape, 1
        else:
            B, D, H, W = labels.shape
        out = torch.empty_like(labels)      # row-major: labels was normalised above
        _ok(lib().oracle_bti_critical(_p(labels), _p(lut_a), _p(lut_c), lut_a.numel(), _p(out), B, D, H, W,
                                      connectivity, min_thick), "bti_critical")
        return out

    @staticmethod
    def norm_act_fwd(x, weight, bias, running_mean, running_var, training, momentum, eps, slope, period,
                     pre_bias=None):
        from .ref_ops import norm_act_fwd_ref
        return norm_act_fwd_ref(x, weight, bias, running_mean, running_var, training, momentum, eps, slope, period,
                                pre_bias)

    @staticmethod
    def norm_act_bwd(x, gy, weight, bias, save_mean, save_invstd, training, slope, period, eps):
        from .ref_ops import norm_act_bwd_ref
        return norm_act_bwd_ref(x, gy, weight, bias, save_mean, save_invstd, training, slope, period, eps)

    @staticmethod
    def channel_sum(x, channels_last=False):
        return x.double().sum(dim=[0] + list(range(2, x.dim()))).float()

    # K3 / K4: the reference's roll / rearrange / MaxPool / MaxUnpool op sequences (ref_ops.py)
    @staticmethod
    def window_gather(x, window, shift):
        from .ref_ops import window_gather_ref
        return window_gather_ref(x, window, shift)

    @staticmethod
    def window_scatter(src, residual, spatial, window, shift):
        from .ref_ops import window_scatter_ref
        return window_scatter_ref(src, residual, spatial, window, shift)

    @staticmethod
    def pool_rows(x, pool):
        from .ref_ops import pool_rows_ref
        return pool_rows_ref(x, pool)

    @staticmethod
    def cell_gather(x, cell, pool):
        from .ref_ops import cell_gather_ref
        return cell_gather_ref(x, cell, pool)

    @staticmethod
    def cell_scatter(src, cell, spatial, pool):
        from .ref_ops import cell_scatter_ref
        return cell_scatter_ref(src, cell, spatial, pool)
