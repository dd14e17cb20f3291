"""Host-side operators of the NexToU graph hot path, backed by libnextou_hip.so.

Each operator is the fused replacement of a run of ATen ops in the reference:

=====================  =====================================================================
``knn_graph``          F.normalize -> bmm -> add -> add -> (+relative_pos) -> neg -> topk
                       (reference network_architecture/torch_edge.py:151-163, 58-110, 12-55)
``mr_aggregate``       2x batched_index_select -> sub -> max -> cat/reshape interleave
                       (reference NexToU_Encoder_Decoder.py:401-409, torch_nn.py:94-115)
``gather_neighbors``   batched_index_select (reference torch_nn.py:94-115)
``bti_critical_map``   softmax/argmax + per-interaction isin/conv3d/where loop
                       (reference loss/bti_loss.py:132-134, 76-117)
``norm_act``           batch_norm | instance_norm -> leaky_relu (reference torch_nn.py:84-90,
                       NexToU_Encoder_Decoder.py:384-390 and the conv stages' norm -> nonlin)
=====================  =====================================================================

Device tensors go to the HIP kernels through the C-ABI (plain pointers + the current HIP stream).
There is no CPU implementation in the product: a CPU tensor raises ``RuntimeError`` unless test
infrastructure has installed a checker with :func:`install_cpu_checker` (tests/ and bench.py's
``cpu_baseline`` leg install the oracle there; nothing in this package imports ``oracle``).
"""
from __future__ import annotations

import contextlib
import math
import threading
from typing import Callable, List, Optional

import torch

from . import _lib

__all__ = [
    "knn_graph", "pairwise_sq_distance", "edge_index_from_nn_idx", "mr_aggregate", "gather_neighbors",
    "argmax_labels", "bti_critical_map", "critical_cross_entropy", "norm_act", "install_cpu_checker", "IndexTape",
    "index_tape", "window_gather", "window_scatter", "pool_rows", "cell_scatter",
]


# ----------------------------------------------------------------------------------------------
# backend selection
# ----------------------------------------------------------------------------------------------
_cpu_checker = None


def install_cpu_checker(backend) -> None:
    """TEST / BASELINE INFRASTRUCTURE ONLY: route CPU tensors to ``backend`` (the oracle).

    ``backend`` must offer the same methods as :class:`_HipBackend`.  Pass ``None`` to remove.
    """
    global _cpu_checker
    _cpu_checker = backend


def _backend_for(t: torch.Tensor):
    if t.device.type == "cuda":
        return _HIP
    if _cpu_checker is not None:
        return _cpu_checker
    raise RuntimeError(
        "nextou_amd: the graph hot path only runs on an MI355X through libnextou_hip.so; got a %s "
        "tensor and there is no CPU fallback." % t.device.type)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """contiguous float32 view/copy (autocast-safe: K1/K2 always run in fp32)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class _HipBackend:
    """Thin marshalling layer: shapes -> ints, tensors -> device pointers."""

    name = "hip"

    @staticmethod
    def knn_graph(x, y, relpos, k_total, algo=_lib.KNN_AUTO, normalize=True):
        L = _lib.lib()
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        out = torch.empty((B, N, k_total), dtype=torch.int32, device=x.device)
        nbytes = L.nextou_knn_workspace_bytes(B, C, N, M, k_total, int(y is not None), algo)
        ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = L.nextou_knn_graph(x.data_ptr(), _ptr(y), _ptr(relpos), out.data_ptr(), ws.data_ptr(),
                                    ws.numel(), B, C, N, M, k_total, algo, int(normalize),
                                    _stream_ptr(x.device))
        _lib.check(rc, "knn_graph")
        return out

    @staticmethod
    def pairwise_distance(x, y, row_start, row_end):
        L = _lib.lib()
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        out = torch.empty((B, row_end - row_start, M), dtype=torch.float32, device=x.device)
        nbytes = L.nextou_pairwise_workspace_bytes(B, N, M, int(y is not None))
        ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = L.nextou_pairwise_distance(x.data_ptr(), _ptr(y), out.data_ptr(), ws.data_ptr(),
                                            ws.numel(), B, C, N, M, row_start, row_end,
                                            _stream_ptr(x.device))
        _lib.check(rc, "pairwise_distance")
        return out

    @staticmethod
    def edge_index(nn_idx, dilation):
        L = _lib.lib()
        B, N, K = nn_idx.shape
        k_out = len(range(0, K, dilation))
        out = torch.empty((2, B, N, k_out), dtype=torch.int64, device=nn_idx.device)
        with torch.cuda.device(nn_idx.device):
            rc = L.nextou_edge_index_i64(nn_idx.data_ptr(), out.data_ptr(), B, N, K, dilation,
                                         _stream_ptr(nn_idx.device))
        _lib.check(rc, "edge_index_i64")
        return out

    @staticmethod
    def mr_has_arg(B, C, N, M, K):
        return bool(_lib.lib().nextou_mr_aggregate_has_arg(B, C, N, M, K))

    @staticmethod
    def mr_fwd(x, y, nn_idx, center, K, idx_step, want_arg=False):
        L = _lib.lib()
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        out = torch.empty((B, 2 * C, N), dtype=torch.float32, device=x.device)
        arg = None
        if want_arg and center is None and L.nextou_mr_aggregate_has_arg(B, C, N, M, K):
            arg = torch.empty((B, C, N), dtype=torch.int16, device=x.device)   # uint16 payload
        with torch.cuda.device(x.device):
            rc = L.nextou_mr_aggregate_fwd(x.data_ptr(), _ptr(y), nn_idx.data_ptr(), _ptr(center),
                                           out.data_ptr(), _ptr(arg), B, C, N, M, K, nn_idx.shape[2], idx_step,
                                           _stream_ptr(x.device))
        _lib.check(rc, "mr_aggregate_fwd")
        return out, arg

    @staticmethod
    def mr_bwd_arg(gout, arg, M, has_y):
        L = _lib.lib()
        B, C, N = arg.shape
        dx = torch.empty((B, C, N), dtype=torch.float32, device=gout.device)
        dy = torch.empty((B, C, M), dtype=torch.float32, device=gout.device) if has_y else None
        with torch.cuda.device(gout.device):
            rc = L.nextou_mr_aggregate_bwd_arg(gout.data_ptr(), arg.data_ptr(), dx.data_ptr(), _ptr(dy), B, C, N, M,
                                               _stream_ptr(gout.device))
        _lib.check(rc, "mr_aggregate_bwd_arg")
        return dx, dy

    @staticmethod
    def mr_grouped_rows_supported(n_windows, C, groups, Nw, K):
        return bool(_lib.lib().nextou_mr_grouped_rows_supported(n_windows, C, groups, Nw, K))

    @staticmethod
    def mr_grouped_rows(windows, nn_idx, K, idx_step, w2, groups, batch, spatial, window, shift, want_a, want_arg, want_stats):
        """K2 + K7 in one launch (csrc/mr_aggregate.hip mr_grp_rows_kernel).  windows (B * nWin, C, Nw), w2 (2C, 2C / groups) ->
        (a, arg, h, partial): the aggregate as a channels-last (B, 2C, *spatial) volume (None unless ``want_a``), the uint16 arg-max
        tape (None unless ``want_arg``), the grouped convolution of the aggregate (channels-last volume) and its (2C, B * nWin, 2)
        float64 statistics partials (None unless ``want_stats``)."""
        L_ = _lib.lib()
        n_windows, C, Nw = windows.shape
        D, H, W = _dhw(spatial)
        wd, wh, ww = _dhw(window)
        sd, sh, sw = _dhw(shift, fill=0)
        shape = (batch, 2 * C) + tuple(spatial)
        h = _empty_channels_last(shape, windows.device)
        a = _empty_channels_last(shape, windows.device) if want_a else None
        arg = torch.empty((n_windows, C, Nw), dtype=torch.int16, device=windows.device) if want_arg else None
        partial = torch.empty((2 * C, n_windows, 2), dtype=torch.float64, device=windows.device) if want_stats else None
        with torch.cuda.device(windows.device):
            rc = L_.nextou_mr_grouped_rows(windows.data_ptr(), nn_idx.data_ptr(), nn_idx.shape[2], idx_step, K, w2.data_ptr(), _ptr(a),
                                           _ptr(arg), h.data_ptr(), _ptr(partial), n_windows if want_stats else 0, batch, C, D, H, W,
                                           wd, wh, ww, sd, sh, sw, groups, _stream_ptr(windows.device))
        _lib.check(rc, "mr_grouped_rows")
        return a, arg, h, partial

    @staticmethod
    def mr_grouped_rows_bwd(dh, w2, arg, groups, spatial, window, shift):
        """dh channels-last (B, 2C, *spatial), arg (B * nWin, C, Nw) -> dx (B * nWin, C, Nw): csrc/mr_aggregate.hip mr_grp_rows_bwd_kernel."""
        L_ = _lib.lib()
        D, H, W = _dhw(spatial)
        wd, wh, ww = _dhw(window)
        sd, sh, sw = _dhw(shift, fill=0)
        dx = torch.empty(arg.shape, dtype=torch.float32, device=dh.device)
        with torch.cuda.device(dh.device):
            rc = L_.nextou_mr_grouped_rows_bwd(dh.data_ptr(), w2.data_ptr(), arg.data_ptr(), dx.data_ptr(), dh.shape[0], arg.shape[1],
                                               D, H, W, wd, wh, ww, sd, sh, sw, groups, _stream_ptr(dh.device))
        _lib.check(rc, "mr_grouped_rows_bwd")
        return dx

    @staticmethod
    def mr_grouped_cm_tiles(B, C, groups, Ng, N, M, K):
        """statistics partials per (sample, channel) the channel-major K2 + K7 launch writes for this shape; 0 = shape not taken"""
        return int(_lib.lib().nextou_mr_grouped_cm_tiles(B, C, groups, Ng, N, M, K))

    @staticmethod
    def mr_grouped_cm(x, y, nn_idx, K, idx_step, w2, groups, want_a, want_arg, want_stats):
        """K2 + K7 for a pooled / self graph in one launch (csrc/mr_aggregate.hip mr_grp_cm_kernel).  x (B, C, N), y (B, C, M) | None,
        w2 (groups * Ng, 2C / groups) -> (a (B, 2C, N) | None, arg (B, C, N) int16 | None, h (B, groups * Ng, N),
        partial (B * groups * Ng, tiles, 2) float64 | None)."""
        L_ = _lib.lib()
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        Ng = w2.shape[0] // groups
        tiles = int(L_.nextou_mr_grouped_cm_tiles(B, C, groups, Ng, N, M, K))
        if tiles <= 0:
            raise RuntimeError("mr_grouped_cm: shape not supported (check mr_grouped_cm_tiles first)")
        h = torch.empty((B, groups * Ng, N), dtype=torch.float32, device=x.device)
        a = torch.empty((B, 2 * C, N), dtype=torch.float32, device=x.device) if want_a else None
        arg = torch.empty((B, C, N), dtype=torch.int16, device=x.device) if want_arg else None
        partial = torch.empty((B * groups * Ng, tiles, 2), dtype=torch.float64, device=x.device) if want_stats else None
        with torch.cuda.device(x.device):
            rc = L_.nextou_mr_grouped_cm(x.data_ptr(), _ptr(y), nn_idx.data_ptr(), nn_idx.shape[2], idx_step, K, w2.data_ptr(), _ptr(a),
                                         _ptr(arg), h.data_ptr(), _ptr(partial), tiles if want_stats else 0, B, C, N, M, groups, Ng,
                                         _stream_ptr(x.device))
        _lib.check(rc, "mr_grouped_cm")
        return a, arg, h, partial

    @staticmethod
    def norm_act_fwd_partials(x, weight, bias, partial, period, eps, slope):
        """K6's normalise + activate for a channel-major fp32 (B, C, S) tensor from ready-made statistics partials (C, n, 2) float64
        -> (y, save_mean, save_invstd)."""
        L_ = _lib.lib()
        B, C = x.shape[:2]
        S = x.numel() // (B * C)
        y = torch.empty_like(x)
        save_mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_norm_act_fwd_partials(x.data_ptr(), _ptr(weight), _ptr(bias), None, None, y.data_ptr(), save_mean.data_ptr(),
                                                 save_invstd.data_ptr(), partial.data_ptr(), partial.shape[1], B, C, S, period, 0.0,
                                                 float(eps), float(slope), _stream_ptr(x.device))
        _lib.check(rc, "norm_act_fwd_partials")
        return y, save_mean, save_invstd

    @staticmethod
    def mr_bwd_wants_idx(B, C, N, K):
        return bool(_lib.lib().nextou_mr_aggregate_bwd_wants_idx(B, C, N, K))

    @staticmethod
    def mr_bwd_arg_idx(gout, arg, nn_idx, K, idx_step):
        """self graph of N <= 512 points: reverse-list gather, no float atomics (bit-reproducible)."""
        L = _lib.lib()
        B, C, N = arg.shape
        dx = torch.empty((B, C, N), dtype=torch.float32, device=gout.device)
        with torch.cuda.device(gout.device):
            rc = L.nextou_mr_aggregate_bwd_arg_idx(gout.data_ptr(), arg.data_ptr(), nn_idx.data_ptr(), dx.data_ptr(), B, C, N, K,
                                                   nn_idx.shape[2], idx_step, _stream_ptr(gout.device))
        _lib.check(rc, "mr_aggregate_bwd_arg_idx")
        return dx

    @staticmethod
    def mr_bwd(gout, x, y, nn_idx, center, K, idx_step):
        L = _lib.lib()
        B, C, N = x.shape
        M = N if y is None else y.shape[2]
        dx = torch.empty_like(x)
        dy = None if y is None else torch.empty_like(y)
        with torch.cuda.device(x.device):
            rc = L.nextou_mr_aggregate_bwd(gout.data_ptr(), x.data_ptr(), _ptr(y), nn_idx.data_ptr(),
                                           _ptr(center), dx.data_ptr(), _ptr(dy), B, C, N, M, K,
                                           nn_idx.shape[2], idx_step, _stream_ptr(x.device))
        _lib.check(rc, "mr_aggregate_bwd")
        return dx, dy

    @staticmethod
    def gather_fwd(src, idx):
        L = _lib.lib()
        B, C, M = src.shape
        _, N, K = idx.shape
        out = torch.empty((B, C, N, K), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            rc = L.nextou_gather_fwd(src.data_ptr(), idx.data_ptr(), out.data_ptr(), B, C, M, N, K,
                                     _stream_ptr(src.device))
        _lib.check(rc, "gather_fwd")
        return out

    @staticmethod
    def gather_bwd(gout, idx, M):
        L = _lib.lib()
        B, C, N, K = gout.shape
        dsrc = torch.empty((B, C, M), dtype=torch.float32, device=gout.device)
        with torch.cuda.device(gout.device):
            rc = L.nextou_gather_bwd(gout.data_ptr(), idx.data_ptr(), dsrc.data_ptr(), B, C, M, N, K,
                                     _stream_ptr(gout.device))
        _lib.check(rc, "gather_bwd")
        return dsrc

    @staticmethod
    def argmax_labels(logits):
        """logits (B, L, *sp) fp32, dense NCDHW or dense channels-last (read where they lie)."""
        L = _lib.lib()
        B, nl = logits.shape[:2]
        V = logits.numel() // (B * nl)
        sl, sv = _logit_strides(logits, nl, V)
        out = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.uint8, device=logits.device)
        with torch.cuda.device(logits.device):
            rc = L.nextou_argmax_labels(logits.data_ptr(), out.data_ptr(), B, nl, V, sl, sv, _stream_ptr(logits.device))
        _lib.check(rc, "argmax_labels")
        return out

    @staticmethod
    def labels_u8(target, n_classes, flag):
        """target float32 / int64 / uint8 (any shape, contiguous) -> uint8 of the same shape; flag (1,) int32 |= 1 if out of range."""
        L = _lib.lib()
        code = {torch.float32: 0, torch.int64: 1, torch.uint8: 2}[target.dtype]
        out = torch.empty(target.shape, dtype=torch.uint8, device=target.device)
        with torch.cuda.device(target.device):
            rc = L.nextou_labels_u8(target.data_ptr(), code, out.data_ptr(), target.numel(), int(n_classes), flag.data_ptr(),
                                    _stream_ptr(target.device))
        _lib.check(rc, "labels_u8")
        return out

    @staticmethod
    def bti_critical(labels, lut_a, lut_c, connectivity, min_thick):
        L = _lib.lib()
        if labels.dim() == 3:
            B, H, W = labels.shape
            D = 1
        else:
            B, D, H, W = labels.shape
        out = torch.empty_like(labels)
        with torch.cuda.device(labels.device):
            rc = L.nextou_bti_critical_map(labels.data_ptr(), lut_a.data_ptr(), lut_c.data_ptr(),
                                           lut_a.numel(), out.data_ptr(), B, D, H, W, connectivity,
                                           min_thick, _stream_ptr(labels.device))
        _lib.check(rc, "bti_critical_map")
        return out


    @staticmethod
    def bti_ce_fwd(logits, target, critical):
        L_ = _lib.lib()
        B, nl = logits.shape[:2]
        V = logits.numel() // (B * nl)
        sl, sv = _logit_strides(logits, nl, V)
        partial = torch.empty((B, L_.nextou_bti_ce_partials()), dtype=torch.float64, device=logits.device)
        with torch.cuda.device(logits.device):
            rc = L_.nextou_bti_ce_fwd(logits.data_ptr(), target.data_ptr(), critical.data_ptr(), partial.data_ptr(),
                                      B, nl, V, sl, sv, _stream_ptr(logits.device))
        _lib.check(rc, "bti_ce_fwd")
        return partial.sum(1)

    @staticmethod
    def bti_ce_bwd(logits, target, critical, scale):
        L_ = _lib.lib()
        B, nl = logits.shape[:2]
        V = logits.numel() // (B * nl)
        sl, sv = _logit_strides(logits, nl, V)
        grad = torch.empty_like(logits)             # preserves the (dense) layout
        with torch.cuda.device(logits.device):
            rc = L_.nextou_bti_ce_bwd(logits.data_ptr(), target.data_ptr(), critical.data_ptr(), scale.data_ptr(),
                                      grad.data_ptr(), B, nl, V, sl, sv, _stream_ptr(logits.device))
        _lib.check(rc, "bti_ce_bwd")
        return grad

    @staticmethod
    def dice_stats_fwd(logits, target, mask):
        """-> (B, L, 3) float64: (intersect, sum_pred, sum_gt) per sample and class (softmax over L inside the kernel)."""
        L_ = _lib.lib()
        B, nl = logits.shape[:2]
        V = logits.numel() // (B * nl)
        sl, sv = _logit_strides(logits, nl, V)
        T = int(L_.nextou_dice_stats_partials())
        partial = torch.empty((B, T, nl, 3), dtype=torch.float64, device=logits.device)
        with torch.cuda.device(logits.device):
            rc = L_.nextou_dice_stats_fwd(logits.data_ptr(), target.data_ptr(), _ptr(mask), partial.data_ptr(), B, nl, V, sl, sv,
                                          _stream_ptr(logits.device))
        _lib.check(rc, "dice_stats_fwd")
        return partial.sum(1)

    @staticmethod
    def dice_stats_bwd(logits, target, mask, g_inter, g_pred):
        L_ = _lib.lib()
        B, nl = logits.shape[:2]
        V = logits.numel() // (B * nl)
        sl, sv = _logit_strides(logits, nl, V)
        grad = torch.empty_like(logits)             # preserves the (dense) layout
        with torch.cuda.device(logits.device):
            rc = L_.nextou_dice_stats_bwd(logits.data_ptr(), target.data_ptr(), _ptr(mask), g_inter.data_ptr(), g_pred.data_ptr(),
                                          grad.data_ptr(), B, nl, V, sl, sv, _stream_ptr(logits.device))
        _lib.check(rc, "dice_stats_bwd")
        return grad

    @staticmethod
    def ce_mean_fwd(logits, target, ignore_index):
        """logits (B, L, *sp) fp32, dense NCDHW or dense channels-last; target int64 (B, *sp) -> (loss sum, counted voxels) float64 scalars."""
        L_ = _lib.lib()
        B, nl = logits.shape[:2]
        V = logits.numel() // (B * nl)
        sl, sv = (1, nl) if _dense_channels_last(logits) is not None else (V, 1)
        partial = torch.empty((L_.nextou_ce_mean_partials(), 2), dtype=torch.float64, device=logits.device)
        with torch.cuda.device(logits.device):
            rc = L_.nextou_ce_mean_fwd(logits.data_ptr(), target.data_ptr(), partial.data_ptr(), B, nl, V, sl, sv, int(ignore_index),
                                       _stream_ptr(logits.device))
        _lib.check(rc, "ce_mean_fwd")
        tot = partial.sum(0)
        return tot[0], tot[1]

    @staticmethod
    def ce_mean_bwd(logits, target, scale, ignore_index):
        """scale: 0-dim fp32 device tensor (upstream gradient / count) -> gradient in the layout of ``logits``."""
        L_ = _lib.lib()
        B, nl = logits.shape[:2]
        V = logits.numel() // (B * nl)
        sl, sv = (1, nl) if _dense_channels_last(logits) is not None else (V, 1)
        grad = torch.empty_like(logits)             # preserves the (dense) layout
        with torch.cuda.device(logits.device):
            rc = L_.nextou_ce_mean_bwd(logits.data_ptr(), target.data_ptr(), scale.data_ptr(), grad.data_ptr(), B, nl, V, sl, sv,
                                       int(ignore_index), _stream_ptr(logits.device))
        _lib.check(rc, "ce_mean_bwd")
        return grad

    @staticmethod
    def norm_act_fwd(x, weight, bias, running_mean, running_var, training, momentum, eps, slope, period,
                     pre_bias=None, channels_last=False):
        """x (B,C,S) f32/bf16 — or, with ``channels_last``, a dense channels_last(_3d) tensor (B,C,*sp) —
        -> (y, save_mean, save_invstd); running statistics updated in place."""
        L_ = _lib.lib()
        B, C = x.shape[:2]
        S = x.numel() // (B * C)
        dt = _NORM_DTYPES[x.dtype]
        y = torch.empty_like(x)
        save_mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = torch.empty((int(L_.nextou_norm_act_workspace_bytes(B, C, S, dt)),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_norm_act_fwd(x.data_ptr(), _ptr(weight), _ptr(bias), _ptr(pre_bias), _ptr(running_mean),
                                        _ptr(running_var),
                                        y.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), ws.data_ptr(),
                                        ws.numel(), B, C, S, period, dt, int(channels_last), int(training),
                                        float(momentum), float(eps), float(slope), _stream_ptr(x.device))
        _lib.check(rc, "norm_act_fwd")
        return y, save_mean, save_invstd

    @staticmethod
    def norm_act_bwd(x, gy, weight, bias, save_mean, save_invstd, training, slope, period, eps=0.0,
                     channels_last=False):
        """(``eps`` is only read by the CPU checker.)  -> (gx, gweight (C,), gbias (C,)) — per normalised channel;
        the caller folds instance-norm rows.  ``gy`` must have the memory layout of ``x``."""
        L_ = _lib.lib()
        B, C = x.shape[:2]
        S = x.numel() // (B * C)
        dt = _NORM_DTYPES[x.dtype]
        gx = torch.empty_like(x)
        gw = torch.empty((C,), dtype=torch.float32, device=x.device)
        gb = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = torch.empty((int(L_.nextou_norm_act_workspace_bytes(B, C, S, dt)),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_norm_act_bwd(x.data_ptr(), gy.data_ptr(), _ptr(weight), _ptr(bias), save_mean.data_ptr(),
                                        save_invstd.data_ptr(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                        ws.data_ptr(), ws.numel(), B, C, S, period, dt, int(channels_last),
                                        int(training), float(slope), _stream_ptr(x.device))
        _lib.check(rc, "norm_act_bwd")
        return gx, gw, gb

    @staticmethod
    def norm_act_bwd_two(x, gy, gy2, weight, bias, save_mean, save_invstd, training, slope):
        """K6's channels-last backward with the gradient arriving as two tensors (summed on load): ``gy`` dense like ``x``, ``gy2`` a
        channel range of wider channels-last rows (``two_gradients_eligible``).  -> (gx, gweight, gbias)"""
        L_ = _lib.lib()
        B, C = x.shape[:2]
        S = x.numel() // (B * C)
        gx = torch.empty_like(x)
        gw = torch.empty((C,), dtype=torch.float32, device=x.device)
        gb = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = torch.empty((int(L_.nextou_norm_act_workspace_bytes(B, C, S, _lib.DTYPE_F32)),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_norm_act_bwd_two(x.data_ptr(), gy.data_ptr(), gy2.data_ptr(), int(gy2.stride(-1)), _ptr(weight), _ptr(bias),
                                            save_mean.data_ptr(), save_invstd.data_ptr(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                            ws.data_ptr(), ws.numel(), B, C, S, int(training), float(slope), _stream_ptr(x.device))
        _lib.check(rc, "norm_act_bwd_two")
        return gx, gw, gb

    @staticmethod
    def channel_sum(x, channels_last=False):
        """(C,) float32 sums over batch and space of x (B,C,*sp) in either memory layout."""
        L_ = _lib.lib()
        B, C = x.shape[:2]
        S = x.numel() // (B * C)
        dt = _NORM_DTYPES[x.dtype]
        out = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = torch.empty((int(L_.nextou_norm_act_workspace_bytes(B, C, S, dt)),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_channel_sum(x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, C, S, dt,
                                       int(channels_last), _stream_ptr(x.device))
        _lib.check(rc, "channel_sum")
        return out

    @staticmethod
    def narrow_copy_sum(g, c_off, C):
        """``(g.narrow(1, c_off, C).contiguous(channels_last), its per-channel sums)`` of a dense channels-last float32 ``g`` in ONE
        pass over the channel range (nextou_narrow_copy_sum), or None for a shape the kernel does not take."""
        import os
        if os.environ.get("NEXTOU_NARROW_COPY_SUM", "1") == "0" or g.dtype != torch.float32 or _dense_channels_last(g) is None:
            return None
        ld = g.shape[1]
        if C % 4 or ld % 4 or c_off % 4 or C > 128 or g.data_ptr() % 16:
            return None
        L_ = _lib.lib()
        P = g.numel() // ld
        dst = _empty_channels_last((g.shape[0], C) + tuple(g.shape[2:]), g.device)
        out = torch.empty((C,), dtype=torch.float32, device=g.device)
        ws = torch.empty((int(L_.nextou_norm_act_workspace_bytes(1, C, P, _lib.DTYPE_F32)),), dtype=torch.uint8, device=g.device)
        with torch.cuda.device(g.device):
            rc = L_.nextou_narrow_copy_sum(g.data_ptr(), dst.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), P, C, ld, c_off,
                                           _stream_ptr(g.device))
        if rc == _lib.ENOTSUP:
            return None
        _lib.check(rc, "narrow_copy_sum")
        return dst, out

    @staticmethod
    def upconv_cat_rows(y2, bias, skip, stride):
        """y2: channels-last (B, T*C1, *sp_in) = the K7 product of the up-convolution's input with its filter; skip: dense channels-last
        (B, C2, *sp_out) -> channels-last (B, C1 + C2, *sp_out) = [pixel-shuffled y2 + bias, skip] (nextou_upconv_cat_rows)."""
        L_ = _lib.lib()
        T = 1
        for s_ in stride:
            T *= int(s_)
        B, c1, c2 = y2.shape[0], y2.shape[1] // T, skip.shape[1]
        sp = tuple(y2.shape[2:])
        d, h, w = ((1,) + sp)[-3:]
        sd, sh, sw = ((1,) + tuple(int(v) for v in stride))[-3:]
        out = _empty_channels_last((B, c1 + c2) + tuple(skip.shape[2:]), y2.device)
        with torch.cuda.device(y2.device):
            rc = L_.nextou_upconv_cat_rows(y2.data_ptr(), _ptr(bias), skip.data_ptr(), out.data_ptr(), B, d, h, w, sd, sh, sw, c1, c2,
                                           _stream_ptr(y2.device))
        _lib.check(rc, "upconv_cat_rows")
        return out

    @staticmethod
    def upconv_cat_direct(x, w2, bias, skip, stride, cout):
        """The same concatenation without the (P_in, T * C1) intermediate (round 6): K7 stores the up-convolution's product rows where the
        pixel shuffle puts them, inside the concatenation buffer (nextou_pw_rows_up); one pass copies the skip half beside them
        (nextou_upconv_cat_rows, y2 = NULL).  x: dense channels-last (B, Cin, *sp_in); w2 (T * cout, Cin); strides 1 / 2 / 4."""
        L_ = _lib.lib()
        B, cin, c2 = x.shape[0], x.shape[1], skip.shape[1]
        sp = tuple(x.shape[2:])
        d, h, w = ((1,) + sp)[-3:]
        sd, sh, sw = ((1,) + tuple(int(v) for v in stride))[-3:]
        out = _empty_channels_last((B, cout + c2) + tuple(skip.shape[2:]), x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_pw_rows_up(x.data_ptr(), w2.data_ptr(), _ptr(bias), out.data_ptr(), w2.shape[0], cin, cin, cout + c2, B, d, h, w,
                                      sd, sh, sw, cout, _stream_ptr(x.device))
            _lib.check(rc, "pw_rows_up")
            rc = L_.nextou_upconv_cat_rows(None, None, skip.data_ptr(), out.data_ptr(), B, d, h, w, sd, sh, sw, cout, c2,
                                           _stream_ptr(x.device))
        _lib.check(rc, "upconv_cat_rows(skip half)")
        return out

    @staticmethod
    def upconv_cat_rows_bwd(g, c1, sp_in, stride):
        """g: dense channels-last (B, C1 + C2, *sp_out) -> (gy2 channels-last (B, T*C1, *sp_in), per-channel sums of g[:, :C1]) in one pass,
        or None for a channel count the kernel does not take."""
        L_ = _lib.lib()
        T = 1
        for s_ in stride:
            T *= int(s_)
        B, ctot = g.shape[0], g.shape[1]
        d, h, w = ((1,) + tuple(sp_in))[-3:]
        sd, sh, sw = ((1,) + tuple(int(v) for v in stride))[-3:]
        P = g.numel() // ctot
        gy2 = _empty_channels_last((B, T * c1) + tuple(sp_in), g.device)
        gb = torch.empty((c1,), dtype=torch.float32, device=g.device)
        ws = torch.empty((int(L_.nextou_norm_act_workspace_bytes(1, c1, P, _lib.DTYPE_F32)),), dtype=torch.uint8, device=g.device)
        with torch.cuda.device(g.device):
            rc = L_.nextou_upconv_cat_rows_bwd(g.data_ptr(), gy2.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws.numel(), B, d, h, w, sd, sh, sw,
                                               c1, ctot - c1, _stream_ptr(g.device))
        if rc == _lib.ENOTSUP:
            return None
        _lib.check(rc, "upconv_cat_rows_bwd")
        return gy2, gb


    # ---- K3 / K4: window / pool data movement between channels-last volumes and channel-major rows ----
    @staticmethod
    def window_gather(x_cl, window, shift):
        """x_cl: dense channels_last(_3d) (B,C,*sp) -> (B * n_windows, C, Nw) contiguous."""
        L_ = _lib.lib()
        B, C = x_cl.shape[:2]
        D, H, W = _dhw(x_cl.shape[2:])
        wd, wh, ww = _dhw(window)
        sd, sh, sw = _dhw(shift, fill=0)
        n_win = (D // wd) * (H // wh) * (W // ww)
        out = torch.empty((B * n_win, C, wd * wh * ww), dtype=torch.float32, device=x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_window_gather(x_cl.data_ptr(), out.data_ptr(), B, C, D, H, W, wd, wh, ww, sd, sh, sw,
                                         _stream_ptr(x_cl.device))
        _lib.check(rc, "window_gather")
        return out

    @staticmethod
    def window_scatter(src_cm, residual_cl, spatial, window, shift):
        """(B * n_windows, C, Nw) [+ residual_cl] -> channels_last(_3d) tensor (B,C,*spatial)."""
        L_ = _lib.lib()
        D, H, W = _dhw(spatial)
        wd, wh, ww = _dhw(window)
        sd, sh, sw = _dhw(shift, fill=0)
        n_win = (D // wd) * (H // wh) * (W // ww)
        B, C = src_cm.shape[0] // n_win, src_cm.shape[1]
        out = _empty_channels_last((B, C) + tuple(spatial), src_cm.device)
        with torch.cuda.device(src_cm.device):
            rc = L_.nextou_window_scatter(src_cm.data_ptr(), _ptr(residual_cl), out.data_ptr(), B, C, D, H, W, wd, wh, ww,
                                          sd, sh, sw, _stream_ptr(src_cm.device))
        _lib.check(rc, "window_scatter")
        return out

    @staticmethod
    def pool_rows(x_cl, pool):
        """channels-last (B,C,*sp) -> (values (B,C,N) contiguous, cell (B,N,C) uint8)."""
        L_ = _lib.lib()
        B, C = x_cl.shape[:2]
        D, H, W = _dhw(x_cl.shape[2:])
        pd, ph, pw = _dhw(pool)
        N = (D // pd) * (H // ph) * (W // pw)
        values = torch.empty((B, C, N), dtype=torch.float32, device=x_cl.device)
        cell = torch.empty((B, N, C), dtype=torch.uint8, device=x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_pool_rows(x_cl.data_ptr(), values.data_ptr(), cell.data_ptr(), B, C, D, H, W, pd, ph, pw,
                                     _stream_ptr(x_cl.device))
        _lib.check(rc, "pool_rows")
        return values, cell

    @staticmethod
    def cell_gather(x_cl, cell, pool):
        L_ = _lib.lib()
        B, C2 = x_cl.shape[:2]
        D, H, W = _dhw(x_cl.shape[2:])
        pd, ph, pw = _dhw(pool)
        N, C = cell.shape[1], cell.shape[2]
        out = torch.empty((B, C2, N), dtype=torch.float32, device=x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_cell_gather(x_cl.data_ptr(), cell.data_ptr(), out.data_ptr(), B, C2, C, D, H, W, pd, ph, pw,
                                       _stream_ptr(x_cl.device))
        _lib.check(rc, "cell_gather")
        return out

    @staticmethod
    def cell_scatter(src_cm, cell, spatial, pool):
        L_ = _lib.lib()
        B, C2 = src_cm.shape[:2]
        D, H, W = _dhw(spatial)
        pd, ph, pw = _dhw(pool)
        C = cell.shape[2]
        out = _empty_channels_last((B, C2) + tuple(spatial), src_cm.device)
        with torch.cuda.device(src_cm.device):
            rc = L_.nextou_cell_scatter(src_cm.data_ptr(), cell.data_ptr(), out.data_ptr(), B, C2, C, D, H, W, pd, ph, pw,
                                        _stream_ptr(src_cm.device))
        _lib.check(rc, "cell_scatter")
        return out


    @staticmethod
    def filter_flip_t(weight):
        """(Co, Ci, *k) float32 filter, any dense layout -> (Ci, Co, *k) channels-last: ``weight.transpose(0, 1).flip(spatial axes)``."""
        L_ = _lib.lib()
        co, ci = weight.shape[:2]
        k = [int(v) for v in weight.shape[2:]]
        st = [int(v) for v in weight.stride()]
        kd, kh, kw = ([1] * (3 - len(k))) + k
        s_sp = ([0] * (3 - len(k))) + st[2:]
        out = _empty_channels_last((ci, co) + tuple(k), weight.device)
        with torch.cuda.device(weight.device):
            rc = L_.nextou_filter_flip_t(weight.data_ptr(), out.data_ptr(), co, ci, kd, kh, kw, st[0], st[1], s_sp[0], s_sp[1], s_sp[2],
                                         _stream_ptr(weight.device))
        _lib.check(rc, "filter_flip_t")
        return out

    @staticmethod
    def depth_unroll(x_cl):
        """dense channels_last_3d (B,C,D,H,W) -> (B*D, 3C, H, W) channels_last: depth taps -1, 0, +1 stacked over channels."""
        L_ = _lib.lib()
        B, C, D, H, W = x_cl.shape
        out = torch.empty((B * D, H, W, 3 * C), dtype=torch.float32, device=x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_depth_unroll(x_cl.data_ptr(), out.data_ptr(), B, C, D, H, W, _stream_ptr(x_cl.device))
        _lib.check(rc, "depth_unroll")
        return out.permute(0, 3, 1, 2)

    # ---- K7: point-wise convolutions on channels-last rows ----
    @staticmethod
    def pw_rows(x_cl, w2, bias, groups):
        """x_cl: dense channels-last (B, groups*K, *sp) float32; w2: (groups*N, K) contiguous -> channels-last (B, groups*N, *sp)."""
        L_ = _lib.lib()
        cin = x_cl.shape[1]
        P = x_cl.numel() // cin
        N, K = w2.shape[0] // groups, w2.shape[1]
        y = _empty_channels_last((x_cl.shape[0], w2.shape[0]) + tuple(x_cl.shape[2:]), x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_pw_rows(x_cl.data_ptr(), w2.data_ptr(), _ptr(bias), y.data_ptr(), P, N, K, groups, cin, w2.shape[0],
                                   _stream_ptr(x_cl.device))
        _lib.check(rc, "pw_rows")
        return y

    @staticmethod
    def pw_wgrad(gy_cl, x_cl, groups):
        """(groups*N, K) float32 = sum over points of gy (B, groups*N, *sp) x (B, groups*K, *sp), both dense channels-last."""
        import ctypes
        L_ = _lib.lib()
        cout, cin = gy_cl.shape[1], x_cl.shape[1]
        P = x_cl.numel() // cin
        N, K = cout // groups, cin // groups
        need = ctypes.c_size_t(0)
        _lib.check(L_.nextou_pw_wgrad_workspace(P, N, K, groups, ctypes.byref(need)), "pw_wgrad_workspace")
        ws = torch.empty((max(int(need.value), 4) // 4,), dtype=torch.float32, device=x_cl.device)
        dw = torch.empty((cout, K), dtype=torch.float32, device=x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_pw_wgrad(gy_cl.data_ptr(), x_cl.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, P, N, K, groups,
                                    cout, cin, 0, _stream_ptr(x_cl.device))
        _lib.check(rc, "pw_wgrad")
        return dw


    # ---- K8: segmentation heads (biased 1x1 convolutions to the class logits) on channels-last rows ----
    @staticmethod
    def head_rows_fwd(x_cl, w2, bias):
        """x_cl: dense channels-last (B, C, *sp) float32; w2: (L, C) contiguous; bias (L,) or None -> channels-last (B, L, *sp)."""
        L_ = _lib.lib()
        C = x_cl.shape[1]
        P = x_cl.numel() // C
        L = w2.shape[0]
        y = _empty_channels_last((x_cl.shape[0], L) + tuple(x_cl.shape[2:]), x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_head_rows_fwd(x_cl.data_ptr(), w2.data_ptr(), _ptr(bias), y.data_ptr(), P, L, C, C, L, _stream_ptr(x_cl.device))
        _lib.check(rc, "head_rows_fwd")
        return y

    @staticmethod
    def head_rows_bwd(gy_cl, x_cl, w2, want_gx, want_gw):
        """(gx channels-last like x_cl | None, gw (L, C) | None, gb (L,) | None) of the head; gy_cl, x_cl dense channels-last float32."""
        import ctypes
        L_ = _lib.lib()
        L, C = w2.shape
        P = x_cl.numel() // C
        gx = _empty_channels_last(tuple(x_cl.shape), x_cl.device) if want_gx else None
        gw = gb = ws = None
        if want_gw:
            need = ctypes.c_size_t(0)
            _lib.check(L_.nextou_head_rows_bwd_workspace(P, L, C, ctypes.byref(need)), "head_rows_bwd_workspace")
            ws = torch.empty((max(int(need.value), 4) // 4,), dtype=torch.float32, device=x_cl.device)
            gw = torch.empty((L, C), dtype=torch.float32, device=x_cl.device)
            gb = torch.empty((L,), dtype=torch.float32, device=x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_head_rows_bwd(gy_cl.data_ptr(), x_cl.data_ptr(), w2.data_ptr(), _ptr(gx), _ptr(gw), _ptr(gb), _ptr(ws),
                                         0 if ws is None else ws.numel() * 4, P, L, C, L, C, _stream_ptr(x_cl.device))
        _lib.check(rc, "head_rows_bwd")
        return gx, gw, gb


    # ---- K9: the stem block (one-channel image -> conv [1,]3x3 -> batch norm -> LeakyReLU) without the convolution's output in memory ----
    @staticmethod
    def stem_fwd(x, w2, pre_bias, gamma, beta, running_mean, running_var, training, momentum, eps, slope, c_pad, want_mask=True):
        """x: dense (B, 1, *sp) float32 image; w2 (C, 9) contiguous -> (y channels-last (B, c_pad, *sp), mean (C,), invstd (C,), moments (54,) f64 | None,
        act (ceil(B D H / 8), W, c_pad / 4) int32 | None: the LeakyReLU mask bits the backward reads)"""
        L_ = _lib.lib()
        B, sp = x.shape[0], tuple(x.shape[2:])
        D, H, W = (1,) * (3 - len(sp)) + sp
        C = w2.shape[0]
        y = _empty_channels_last((B, c_pad) + sp, x.device)
        mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        moments = torch.empty((54,), dtype=torch.float64, device=x.device) if training else None
        act = torch.empty(((B * D * H + 7) // 8, W, c_pad // 4), dtype=torch.int32, device=x.device) if (training and want_mask) else None
        need = int(L_.nextou_stem_workspace_bytes(B, D, H, W, c_pad))
        ws = torch.empty((max(need, 8) // 8,), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_stem_fwd(x.data_ptr(), w2.data_ptr(), _ptr(pre_bias), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                    y.data_ptr(), _ptr(act), mean.data_ptr(), invstd.data_ptr(), _ptr(moments), ws.data_ptr(), ws.numel() * 8,
                                    B, D, H, W, C, c_pad, 1 if training else 0, float(momentum), float(eps), float(slope), _stream_ptr(x.device))
        _lib.check(rc, "stem_fwd")
        return y, mean, invstd, moments, act

    @staticmethod
    def stem_bwd(x, gy_cl, act, w2, gamma, mean, invstd, moments, slope, want_gw, want_gg, want_gb):
        """(gw (C, 9) | None, ggamma (C,) | None, gbeta (C,) | None) of the stem block from gy_cl, dense channels-last (B, c_pad, *sp) float32."""
        L_ = _lib.lib()
        B, sp = x.shape[0], tuple(x.shape[2:])
        D, H, W = (1,) * (3 - len(sp)) + sp
        C, c_pad = w2.shape[0], gy_cl.shape[1]
        gw = torch.empty((C, 9), dtype=torch.float32, device=x.device) if want_gw else None
        gg = torch.empty((C,), dtype=torch.float32, device=x.device) if want_gg else None
        gb = torch.empty((C,), dtype=torch.float32, device=x.device) if want_gb else None
        need = int(L_.nextou_stem_workspace_bytes(B, D, H, W, c_pad))
        ws = torch.empty((max(need, 8) // 8,), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = L_.nextou_stem_bwd(x.data_ptr(), gy_cl.data_ptr(), act.data_ptr(), w2.data_ptr(), _ptr(gamma), mean.data_ptr(), invstd.data_ptr(),
                                    moments.data_ptr(), _ptr(gw), _ptr(gg), _ptr(gb), ws.data_ptr(), ws.numel() * 8, B, D, H, W, C, c_pad,
                                    float(slope), _stream_ptr(x.device))
        _lib.check(rc, "stem_bwd")
        return gw, gg, gb

    # ---- K7 + K6 fused: statistics epilogue / operand prologue GEMMs and K6 in pieces ----
    @staticmethod
    def pw_rows_fused(x_cl, w2, groups, pro=None, want_stats=False, bwd=None):
        """x_cl dense channels-last (B, groups*K, *sp); w2 (groups*N, K).  ``pro`` = (scale, shift, slope): normalise + activate the
        operand on load.  ``want_stats``: also return the (C, T, 2) float64 (sum, sum of squares) partials of the output.
        ``bwd`` = (h_cl, weight, bias, mean, invstd, slope): gradient-statistics epilogue, returns the (sum dz, sum dz*xhat) partials."""
        L_ = _lib.lib()
        cin = x_cl.shape[1]
        P = x_cl.numel() // cin
        N, K = w2.shape[0] // groups, w2.shape[1]
        y = _empty_channels_last((x_cl.shape[0], w2.shape[0]) + tuple(x_cl.shape[2:]), x_cl.device)
        partial, tiles = None, 0
        if want_stats or bwd is not None:
            tiles = int(L_.nextou_pw_rows_tiles(P, N, K, groups, cin, w2.shape[0], int(pro is not None), int(bwd is not None)))
            partial = torch.empty((w2.shape[0], tiles, 2), dtype=torch.float64, device=x_cl.device)
        ps, psh, pslope = (pro[0], pro[1], float(pro[2])) if pro is not None else (None, None, 1.0)
        bh, bw, bb, bm, bi, bslope = bwd if bwd is not None else (None, None, None, None, None, 1.0)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_pw_rows_fused(x_cl.data_ptr(), w2.data_ptr(), y.data_ptr(), P, N, K, groups, cin, w2.shape[0],
                                         _ptr(ps), _ptr(psh), pslope, _ptr(partial), tiles, _ptr(bh), 0 if bh is None else bh.shape[1],
                                         _ptr(bw), _ptr(bb), _ptr(bm), _ptr(bi), float(bslope), _stream_ptr(x_cl.device))
        _lib.check(rc, "pw_rows_fused")
        return y, partial

    @staticmethod
    def pw_wgrad_fused(gy_cl, x_cl, groups, pro):
        import ctypes
        L_ = _lib.lib()
        cout, cin = gy_cl.shape[1], x_cl.shape[1]
        P = x_cl.numel() // cin
        N, K = cout // groups, cin // groups
        need = ctypes.c_size_t(0)
        _lib.check(L_.nextou_pw_wgrad_workspace(P, N, K, groups, ctypes.byref(need)), "pw_wgrad_workspace")
        ws = torch.empty((max(int(need.value), 4) // 4,), dtype=torch.float32, device=x_cl.device)
        dw = torch.empty((cout, K), dtype=torch.float32, device=x_cl.device)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_pw_wgrad_fused(gy_cl.data_ptr(), x_cl.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, P, N, K, groups,
                                          cout, cin, 0, pro[0].data_ptr(), pro[1].data_ptr(), float(pro[2]), _stream_ptr(x_cl.device))
        _lib.check(rc, "pw_wgrad_fused")
        return dw

    @staticmethod
    def norm_finalize(partial, count, C, device, weight, bias, pre_bias, running_mean, running_var, training, momentum, eps):
        """-> (save_mean, save_invstd, scale, shift), each (C,) float32; running statistics updated in place when training."""
        L_ = _lib.lib()
        out = torch.empty((4, C), dtype=torch.float32, device=device)
        tiles = 0 if partial is None else partial.shape[1]
        with torch.cuda.device(device):
            rc = L_.nextou_norm_finalize(_ptr(partial), tiles, float(count), _ptr(pre_bias), _ptr(running_mean), _ptr(running_var),
                                         out[0].data_ptr(), out[1].data_ptr(), _ptr(weight), _ptr(bias), out[2].data_ptr(),
                                         out[3].data_ptr(), C, int(training), float(momentum), float(eps), _stream_ptr(device))
        _lib.check(rc, "norm_finalize")
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def norm_apply_rows(x_cl, residual_cl, weight, bias, mean, invstd, slope):
        L_ = _lib.lib()
        C = x_cl.shape[1]
        y = torch.empty_like(x_cl)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_norm_apply_rows(x_cl.data_ptr(), _ptr(residual_cl), y.data_ptr(), _ptr(weight), _ptr(bias), mean.data_ptr(),
                                           invstd.data_ptr(), x_cl.numel() // C, C, float(slope), _stream_ptr(x_cl.device))
        _lib.check(rc, "norm_apply_rows")
        return y

    @staticmethod
    def norm_bwd_finalize(partial, count, C, device, training):
        """-> (coeff (C, 2), gweight (C,), gbias (C,))"""
        L_ = _lib.lib()
        coeff = torch.empty((C, 2), dtype=torch.float32, device=device)
        gwb = torch.empty((2, C), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            rc = L_.nextou_norm_bwd_finalize(partial.data_ptr(), partial.shape[1], float(count), coeff.data_ptr(), gwb[0].data_ptr(),
                                             gwb[1].data_ptr(), C, int(training), _stream_ptr(device))
        _lib.check(rc, "norm_bwd_finalize")
        return coeff, gwb[0], gwb[1]

    @staticmethod
    def norm_bwd_apply_rows(x_cl, gy_cl, coeff, weight, bias, mean, invstd, slope):
        L_ = _lib.lib()
        C = x_cl.shape[1]
        gx = torch.empty_like(x_cl)
        with torch.cuda.device(x_cl.device):
            rc = L_.nextou_norm_bwd_apply_rows(x_cl.data_ptr(), gy_cl.data_ptr(), gx.data_ptr(), coeff.data_ptr(), _ptr(weight), _ptr(bias),
                                               mean.data_ptr(), invstd.data_ptr(), x_cl.numel() // C, C, float(slope),
                                               _stream_ptr(x_cl.device))
        _lib.check(rc, "norm_bwd_apply_rows")
        return gx


def _logit_strides(logits, nl, V):
    """(stride_l, stride_v) of dense logits (B, L, *sp): (1, L) when stored channels-last (even class count: the rows kernels read
    8-byte pairs), else (V, 1) — the caller made NCDHW-dense tensors of everything else (:func:`_logits_in_place`)."""
    return (1, nl) if (_dense_channels_last(logits) is not None and nl % 2 == 0 and nl <= 32) else (V, 1)


def _logits_in_place(t: torch.Tensor) -> torch.Tensor:
    """fp32 logits the K5 kernels can read where they lie: dense channels-last with an even class count <= 32 stays as it is,
    anything else becomes an NCDHW-contiguous fp32 tensor."""
    if t.dtype != torch.float32:
        t = t.float()
    if t.is_cuda and _dense_channels_last(t) is not None and t.shape[1] % 2 == 0 and t.shape[1] <= 32:
        return t
    return t.contiguous()


def _dhw(sizes, fill=1):
    """(H,W) or (D,H,W) -> (D,H,W): 2-D volumes are one slice thick."""
    sizes = [int(v) for v in sizes]
    return tuple([fill] * (3 - len(sizes)) + sizes)


def _empty_channels_last(shape, device):
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}[len(shape)]
    return torch.empty(shape, dtype=torch.float32, device=device, memory_format=mf)


_NORM_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}
_HIP = _HipBackend()


# ----------------------------------------------------------------------------------------------
# index tape: record / replay of the discrete decisions (kNN ids, max-pool arg-max locations).
# The model is discontinuous in those (SURVEY.md §7 hard part 0), so whole-model parity is
# measured teacher-forced (protocol P-B): record on one implementation, replay on the other.
# ----------------------------------------------------------------------------------------------
class IndexTape:
    def __init__(self, entries: Optional[List[torch.Tensor]] = None):
        self.entries: List[torch.Tensor] = list(entries or [])
        self.cursor = 0
        self.replay = entries is not None

    def take(self, compute: Callable[[], torch.Tensor], device) -> torch.Tensor:
        if self.replay:
            if self.cursor >= len(self.entries):
                raise RuntimeError("IndexTape exhausted after %d entries" % self.cursor)
            t = self.entries[self.cursor].to(device)
            self.cursor += 1
            return t
        t = compute()
        self.entries.append(t.detach().cpu())
        return t


_tape: Optional[IndexTape] = None


@contextlib.contextmanager
def index_tape(tape: IndexTape):
    """Test hook: while active, ``knn_graph`` and the pooled stage's arg-max go through ``tape``."""
    global _tape
    prev, _tape = _tape, tape
    try:
        yield tape
    finally:
        _tape = prev


def taped(compute: Callable[[], torch.Tensor], device) -> torch.Tensor:
    return compute() if _tape is None else _tape.take(compute, device)


def tape_active() -> bool:
    return _tape is not None


def tape_replaying() -> bool:
    return _tape is not None and _tape.replay


# ----------------------------------------------------------------------------------------------
# public operators
# ----------------------------------------------------------------------------------------------
_ALGOS = {"auto": _lib.KNN_AUTO, "fused": _lib.KNN_FUSED, "naive": _lib.KNN_NAIVE}


@torch.no_grad()
def knn_graph(x: torch.Tensor, y: Optional[torch.Tensor] = None,
              relative_pos: Optional[torch.Tensor] = None, k: int = 9, algo: str = "auto",
              normalize: bool = True) -> torch.Tensor:
    """Indices of the ``k`` nearest candidates of every point, ascending by (distance, index).

    x: (B,C,N[,1]) queries; y: (B,C,M[,1]) candidates or None (self graph); relative_pos:
    (1,N,M) or (N,M) additive bias.  Returns int32 (B,N,k).  With ``normalize`` (default) the
    L2 normalisation over channels is part of the op, as in the reference's
    DenseDilatedKnnGraph.forward; ``normalize=False`` is ``dense_knn_matrix`` called directly.
    """
    x3 = _f32c(x.detach().reshape(x.shape[0], x.shape[1], -1))
    y3 = None if y is None else _f32c(y.detach().reshape(y.shape[0], y.shape[1], -1))
    B, C, N = x3.shape
    M = N if y3 is None else y3.shape[2]
    if y3 is not None and (y3.shape[0] != B or y3.shape[1] != C):
        raise ValueError("knn_graph: x %s and y %s disagree in batch/channels" % (tuple(x3.shape), tuple(y3.shape)))
    rp = None
    if relative_pos is not None:
        rp = _f32c(relative_pos.detach()).reshape(-1, relative_pos.shape[-1])
        if tuple(rp.shape) != (N, M):
            raise ValueError("knn_graph: relative_pos %s does not match (N=%d, M=%d)" % (tuple(relative_pos.shape), N, M))
    if k > M:
        raise RuntimeError("knn_graph: k=%d out of range for %d candidates" % (k, M))
    be = _backend_for(x3)
    return taped(lambda: be.knn_graph(x3, y3, rp, int(k), _ALGOS[algo], bool(normalize)), x3.device)


@torch.no_grad()
def pairwise_sq_distance(x: torch.Tensor, y: Optional[torch.Tensor] = None, row_start: int = 0,
                         row_end: Optional[int] = None) -> torch.Tensor:
    """``(|x|^2 + (-2 x.y^T)) + |y|^2^T`` for x (B,C,N), y (B,C,M) or None -> (B,rows,M) float32."""
    x3 = _f32c(x.detach())
    y3 = None if y is None else _f32c(y.detach())
    row_end = x3.shape[2] if row_end is None else row_end
    return _backend_for(x3).pairwise_distance(x3, y3, int(row_start), int(row_end))


def edge_index_from_nn_idx(nn_idx: torch.Tensor, dilation: int = 1) -> torch.Tensor:
    """(B,N,K) int32 -> the reference's (2,B,N,K/d) int64 ``edge_index`` (torch_edge.py:89-90,126-136)."""
    return _backend_for(nn_idx).edge_index(nn_idx.contiguous(), int(dilation))


class _MRAggregate(torch.autograd.Function):
    """Forward records which source id won every max (uint16) whenever a gradient will be needed and
    the kernel supports it; backward is then a pure scatter-add and x / y / ids are not kept alive."""

    @staticmethod
    def forward(ctx, x, y, nn_idx, center, K, idx_step):
        be = _backend_for(x)
        needs_grad = x.requires_grad or (y is not None and y.requires_grad)
        out, arg = be.mr_fwd(x, y, nn_idx, center, K, idx_step, want_arg=needs_grad)
        ctx.has_y, ctx.has_center = y is not None, center is not None
        ctx.K, ctx.idx_step = K, idx_step
        ctx.M = x.shape[2] if y is None else y.shape[2]
        ctx.use_arg = arg is not None
        ctx.use_idx = bool(ctx.use_arg and y is None and getattr(be, "mr_bwd_wants_idx", lambda *a: False)(
            x.shape[0], x.shape[1], x.shape[2], K))
        if ctx.use_idx:         # window graphs: the backward gathers over reverse neighbour lists built from nn_idx
            ctx.save_for_backward(arg, nn_idx)
        elif ctx.use_arg:
            ctx.save_for_backward(arg)
        else:
            ctx.save_for_backward(x, y if y is not None else x.new_empty(0), nn_idx,
                                  center if center is not None else nn_idx.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, gout):
        gout = _f32c(gout)
        if ctx.use_idx:
            arg, nn_idx = ctx.saved_tensors
            return _backend_for(gout).mr_bwd_arg_idx(gout, arg, nn_idx, ctx.K, ctx.idx_step), None, None, None, None, None
        if ctx.use_arg:
            (arg,) = ctx.saved_tensors
            dx, dy = _backend_for(gout).mr_bwd_arg(gout, arg, ctx.M, ctx.has_y)
            return dx, dy, None, None, None, None
        x, y, nn_idx, center = ctx.saved_tensors
        y = y if ctx.has_y else None
        center = center if ctx.has_center else None
        dx, dy = _backend_for(x).mr_bwd(gout, x, y, nn_idx, center, ctx.K, ctx.idx_step)
        return dx, dy, None, None, None, None


def mr_aggregate(x: torch.Tensor, nn_idx: torch.Tensor, y: Optional[torch.Tensor] = None,
                 center_idx: Optional[torch.Tensor] = None, k: Optional[int] = None,
                 idx_step: int = 1) -> torch.Tensor:
    """Max-relative aggregation with interleaved output channels.

    x: (B,C,N) float; y: (B,C,M) or None; nn_idx: (B,N,K_total) int32 ids into y (or x);
    uses neighbours ``nn_idx[..., ::idx_step][..., :k]``.  Returns (B,2C,N):
    ``out[:,2c] = x[:,c]``, ``out[:,2c+1] = max_j(src[:,c,nn_idx_j] - x[:,c,centre_j])``.
    """
    if k is None:
        k = len(range(0, nn_idx.shape[2], idx_step))
    x = _f32c(x)
    y = None if y is None else _f32c(y)
    nn_idx = nn_idx.contiguous()
    if nn_idx.dtype != torch.int32:
        nn_idx = nn_idx.to(torch.int32)
    if center_idx is not None:
        center_idx = center_idx.contiguous().to(torch.int32)
    return _MRAggregate.apply(x, y, nn_idx, center_idx, int(k), int(idx_step))


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx):
        ctx.save_for_backward(idx)
        ctx.M = src.shape[2]
        return _backend_for(src).gather_fwd(src, idx)

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        return _backend_for(gout).gather_bwd(_f32c(gout), idx, ctx.M), None


def gather_neighbors(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``out[b,c,n,j] = x[b,c,idx[b,n,j]]`` — x (B,C,M), idx (B,N,K) -> (B,C,N,K)."""
    return _Gather.apply(_f32c(x), idx.contiguous().to(torch.int32))


class _CriticalCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, critical):
        ctx.save_for_backward(logits, target, critical)
        return _backend_for(logits).bti_ce_fwd(logits, target, critical)          # (B,) float64

    @staticmethod
    def backward(ctx, g):
        logits, target, critical = ctx.saved_tensors
        scale = g.to(torch.float64).reshape(-1).contiguous()      # (B,) upstream gradient per sample
        return _backend_for(logits).bti_ce_bwd(logits, target, critical, scale), None, None


def critical_cross_entropy(logits: torch.Tensor, target: torch.Tensor, critical: torch.Tensor) -> torch.Tensor:
    """``sum_v critical * CE_float64(logits, target)`` per batch element -> (B,) float64.

    logits (B,L,*sp) float32, target / critical (B,*sp) uint8.  The fused replacement of
    ``CrossEntropyLoss(reduction='none')(x.double(), y) * critical`` + sum (reference bti_loss.py:141-143).
    """
    return _CriticalCE.apply(_logits_in_place(logits), target.contiguous(), critical.contiguous())


class _MeanCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        total, count = _HIP.ce_mean_fwd(logits, target, ignore_index)
        ctx.save_for_backward(logits, target, count)
        ctx.ignore_index = ignore_index
        return (total / count).to(torch.float32)          # (0 / 0 = NaN when nothing counts, as torch.nn.CrossEntropyLoss)

    @staticmethod
    def backward(ctx, g):
        logits, target, count = ctx.saved_tensors
        scale = (g.to(torch.float64) / count).to(torch.float32).reshape(1)
        return _HIP.ce_mean_bwd(logits, target, scale, ctx.ignore_index), None, None


def cross_entropy_mean_eligible(logits: torch.Tensor, target: torch.Tensor) -> bool:
    """fp32 device logits (B, L <= 32, *spatial), dense in NCDHW or channels-last memory, int64 class-index target (B, *spatial)."""
    import os
    if os.environ.get("NEXTOU_FUSED_CE", "1") == "0":
        return False
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() < 3 or logits.shape[1] > 32 or logits.shape[1] < 2:
        return False
    if target.dtype != torch.int64 or target.device != logits.device or tuple(target.shape) != (logits.shape[0],) + tuple(logits.shape[2:]):
        return False
    return logits.is_contiguous() or _dense_channels_last(logits) is not None


def cross_entropy_mean(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """``torch.nn.functional.cross_entropy(logits, target, reduction='mean', ignore_index=...)`` as one kernel each way over the logits
    where they lie (K5c, csrc/bti_critical.hip): the deep-supervision CE of the NexToU trainers' losses (reference
    nnUNetTrainer_NexToU*.py via nnU-Net's RobustCrossEntropyLoss).  fp32 arithmetic as ATen's; float64 partial sums in a fixed order;
    the gradient comes back in the logits' own memory layout (channels-last stays channels-last)."""
    return _MeanCE.apply(logits, target.contiguous(), int(ignore_index))


class _DiceStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, mask):
        stats = _HIP.dice_stats_fwd(logits, target, mask)                  # (B, L, 3) float64
        ctx.save_for_backward(logits, target, mask if mask is not None else target.new_empty(0))
        ctx.has_mask = mask is not None
        out = stats.to(torch.float32)
        return out[..., 0].contiguous(), out[..., 1].contiguous(), out[..., 2].contiguous()

    @staticmethod
    def backward(ctx, g_inter, g_pred, g_gt):
        logits, target, mask = ctx.saved_tensors
        gi = torch.zeros(logits.shape[:2], dtype=torch.float64, device=logits.device) if g_inter is None else g_inter.to(torch.float64).contiguous()
        gp = torch.zeros(logits.shape[:2], dtype=torch.float64, device=logits.device) if g_pred is None else g_pred.to(torch.float64).contiguous()
        return _HIP.dice_stats_bwd(logits, target, mask if ctx.has_mask else None, gi, gp), None, None


def dice_stats_eligible(logits: torch.Tensor, target: torch.Tensor) -> bool:
    """fp32 device logits (B, 2 <= L <= 32, *spatial), dense in NCDHW or channels-last memory, outside autocast; a label-map target
    (B, 1, *spatial) or (B, *spatial) of any real dtype (not one-hot).  ``NEXTOU_FUSED_DICE=0`` keeps nnU-Net's op sequence (A/B)."""
    import os
    if os.environ.get("NEXTOU_FUSED_DICE", "1") == "0":
        return False
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() < 3 or not 2 <= logits.shape[1] <= 32:
        return False
    if torch.is_autocast_enabled("cuda") or target.device != logits.device or target.is_complex():
        return False
    if tuple(target.shape) not in ((logits.shape[0], 1) + tuple(logits.shape[2:]), (logits.shape[0],) + tuple(logits.shape[2:])):
        return False
    return logits.is_contiguous() or _dense_channels_last(logits) is not None


def dice_stats(logits: torch.Tensor, target: torch.Tensor, loss_mask: Optional[torch.Tensor] = None):
    """``(intersect, sum_pred, sum_gt)``, each (B, L) float32: the volume sums of nnU-Net's soft Dice with ``softmax(x, 1)`` as its
    non-linearity — ``sum_v w p [y = l]``, ``sum_v w p``, ``sum_v w [y = l]`` (w = ``loss_mask`` or 1) — from ONE pass over the logits
    where they lie (K5d, csrc/bti_critical.hip); the gradient w.r.t. the logits is one more pass.  Reference call sites:
    loss/compound_bti_loss.py:29-30, :53-55 (nnunetv2 SoftDiceLoss / MemoryEfficientSoftDiceLoss with softmax_helper_dim1)."""
    y = target[:, 0] if target.dim() == logits.dim() else target
    y = y.to(torch.uint8).contiguous()
    m = None
    if loss_mask is not None:
        m = (loss_mask[:, 0] if loss_mask.dim() == logits.dim() else loss_mask).to(torch.uint8).contiguous()
    return _DiceStats.apply(_logits_in_place(logits), y, m)


class _NormAct(torch.autograd.Function):
    """(batch | instance) norm -> LeakyReLU as one op; saves only ``x`` and 2C floats for backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, slope, instance,
                pre_bias, pad_holder=None, stats_partial=None):
        shape = x.shape
        B, C = shape[0], shape[1]
        period = C if instance else 0
        be = _backend_for(x)
        # zero-padded input channels (channel_pad.py): the parameters keep their real length; they are staged into a persistent
        # (5, C) buffer of the norm module — rows weight / bias / pre_bias / running mean / running var, padding lanes 1 / 0 / 0 /
        # 0 / 1 — with ONE multi-tensor copy, and the running statistics come back with one more (F.pad per vector cost ten
        # launches per call and two for the write-back: ~170 of the step's ~1 700 launches)
        c_real = C
        stage = None
        if pad_holder is not None:
            c_real = pad_holder.num_features
            stage = _pad_stage(pad_holder, C, c_real, x.device)
            dst, src = [], []
            for row, t in enumerate((weight, bias, pre_bias, running_mean, running_var)):
                if t is not None:
                    dst.append(stage[row, :c_real])
                    src.append(t.detach())
            if dst:
                torch._foreach_copy_(dst, src)
            k_weight = stage[0] if weight is not None else None
            k_bias = stage[1] if bias is not None else None
            k_pre = stage[2] if pre_bias is not None else None
            k_rm = stage[3] if running_mean is not None else None
            k_rv = stage[4] if running_var is not None else None
        else:
            k_weight, k_bias, k_pre, k_rm, k_rv = weight, bias, pre_bias, running_mean, running_var
        cl = _dense_channels_last(x) if (x.is_cuda and not instance) else None
        if cl is not None:      # NDHWC / NHWC memory goes to the channels-last kernels as it is
            y, mean, invstd = be.norm_act_fwd(x, k_weight, k_bias, k_rm, k_rv, training, momentum, eps,
                                              slope, 0, k_pre, channels_last=True)
            x3 = x
        else:
            x3 = x.contiguous()
            x3 = x3.view(1, B * C, -1) if instance else x3.view(B, C, -1)
            if stats_partial is not None:       # instance statistics already summed by the producer (mr_grouped_cm): no statistics pass
                y, mean, invstd = be.norm_act_fwd_partials(x3, k_weight, k_bias, stats_partial, period, eps, slope)
            else:
                y, mean, invstd = be.norm_act_fwd(x3, k_weight, k_bias, k_rm, k_rv, training, momentum, eps,
                                                  slope, period, k_pre)
            y = y.view(shape)
        if stage is not None:
            if training and running_mean is not None and running_var is not None:
                torch._foreach_copy_([running_mean, running_var], [stage[3, :c_real], stage[4, :c_real]])
            wb = stage[0:2].clone()         # the stage is rewritten by the next call; backward reads its own copy
            k_weight = wb[0] if weight is not None else None
            k_bias = wb[1] if bias is not None else None
        ctx.save_for_backward(x3, k_weight, k_bias, mean, invstd)
        ctx.pre_bias_like = pre_bias
        ctx.cfg = (bool(training), float(slope), period, shape, B, C, float(eps), cl, c_real)
        return y

    @staticmethod
    def backward(ctx, gy):
        x3, weight, bias, mean, invstd = ctx.saved_tensors
        training, slope, period, shape, B, C, eps, cl, c_real = ctx.cfg
        box = getattr(ctx, "fork_box", None)            # skip_fork(): the skip connection's gradient, parked by _SkipFork.backward
        g2 = box.pop() if box else None
        if gy.dtype != x3.dtype:
            gy = gy.to(x3.dtype)
        if g2 is not None and not (cl is not None and two_gradients_eligible(x3, gy, g2)):
            gy = gy + g2                                 # what autograd itself would have done
            g2 = None
        if g2 is not None:
            gx, gw, gb = _HIP.norm_act_bwd_two(x3, gy.contiguous(memory_format=cl), g2, weight, bias, mean, invstd, training, slope)
        elif cl is not None:
            gx, gw, gb = _backend_for(x3).norm_act_bwd(x3, gy.contiguous(memory_format=cl), weight, bias, mean, invstd,
                                                       training, slope, 0, eps, channels_last=True)
        else:
            gx, gw, gb = _backend_for(x3).norm_act_bwd(x3, gy.contiguous().view(x3.shape), weight, bias, mean, invstd,
                                                       training, slope, period, eps)
            gx = gx.view(shape)
        if period:
            gw, gb = gw.view(B, C).sum(0), gb.view(B, C).sum(0)
        gpre = None
        if ctx.pre_bias_like is not None and ctx.needs_input_grad[10]:
            # d/d(pre_bias): under batch statistics the output does not depend on it at all; with running
            # statistics z = (x + b - rm) * invstd * w + beta, so the gradient is w * invstd * sum(dz)
            if training:
                gpre = torch.zeros_like(ctx.pre_bias_like)
            else:
                gpre = (gb * invstd * (weight if weight is not None else 1.0))[:c_real].to(ctx.pre_bias_like.dtype)
        gw = gw[:c_real].to(weight.dtype) if weight is not None and ctx.needs_input_grad[1] else None
        gb = gb[:c_real].to(bias.dtype) if bias is not None and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None, None, None, None, None, gpre, None, None


def two_gradients_eligible(x: torch.Tensor, gy: torch.Tensor, g2: torch.Tensor) -> bool:
    """K6's two-gradient backward (nextou_norm_act_bwd_two): dense channels-last fp32 ``x`` with C <= 128 a multiple of 4, and ``g2`` the
    same logical shape as a channel range of wider channels-last rows (stride 1 on the channel axis, row stride a multiple of 4, dense
    rows in batch / space order, 16-byte aligned) — the ``g.narrow(1, c, C)`` the concatenation's backward hands the skip connection."""
    if not (x.is_cuda and x.dtype == torch.float32 and g2.dtype == torch.float32 and gy.dtype == torch.float32) or x.dim() not in (4, 5):
        return False
    C = x.shape[1]
    if tuple(g2.shape) != tuple(x.shape) or C % 4 or C > 128 or _dense_channels_last(x) is None:
        return False
    ld = g2.stride(-1)
    if g2.stride(1) != 1 or ld < C or ld % 4 or g2.data_ptr() % 16 or x.data_ptr() % 16:
        return False
    expect = ld
    for d in range(x.dim() - 1, 1, -1):                 # spatial axes innermost first, then the batch: rows in (B, *spatial) order
        if g2.stride(d) != expect:
            return False
        expect *= x.shape[d]
    return g2.stride(0) == expect


SKIP_FORK_DEFAULT = "0"


class _SkipFork(torch.autograd.Function):
    """Identity with two outputs for a tensor that has two consumers — an encoder stage's output going to the next stage and, as the skip
    connection, to the decoder (reference NexToU_Encoder_Decoder.py:143-150).  Its backward does NOT add the two incoming gradients: it
    parks the second in the box shared with the producing _NormAct node, whose backward kernels read both (nextou_norm_act_bwd_two) or,
    where they cannot, add them there.  Removes autograd's own aten::add pass over the stage-0 / stage-1 tensors (0.98 ms of the cfg-2
    step) — at the price of reading the strided skip gradient twice; see skip_fork for the measured outcome."""

    @staticmethod
    def forward(ctx, y, box):
        ctx.box = box
        ctx.set_materialize_grads(False)
        return y.view_as(y), y.view_as(y)

    @staticmethod
    def backward(ctx, g_main, g_skip):
        if g_skip is None or g_main is None:
            return (g_main if g_skip is None else g_skip), None
        ctx.box.append(g_skip)
        return g_main, None


def skip_fork(y: torch.Tensor):
    """``(y_next, y_skip)``: the same values twice; when ``y`` comes straight out of a fused norm (_NormAct) and gradients are being
    recorded, the two are tied to that node so that its backward takes their gradients unsummed (see _SkipFork).  Anything else:
    ``(y, y)``.

    OFF by default (``NEXTOU_SKIP_FORK=1`` switches it on): measured on one MI355X box, cfg 2, alternating runs
    (profiles/r06_step_ab.md) the step takes 166.16 / 166.25 ms with it and 166.19 ms without — the two-gradient kernels read the skip
    gradient as 160-byte halves of 320-byte rows (half-used cache lines, twice), which costs what autograd's add pass saved."""
    import os
    node = y.grad_fn
    if node is None or not torch.is_grad_enabled() or type(node).__name__ != "_NormActBackward" or \
            os.environ.get("NEXTOU_SKIP_FORK", SKIP_FORK_DEFAULT) != "1" or getattr(node, "fork_box", None) is not None:
        return y, y
    box = []
    node.fork_box = box
    return _SkipFork.apply(y, box)


def _pad_stage(holder, C: int, c_real: int, device) -> torch.Tensor:
    """The persistent (5, C) parameter stage of a norm module whose input carries ``C - c_real`` zero channels."""
    stage = getattr(holder, "_pad_stage", None)
    if stage is None or stage.shape[1] != C or stage.device != device:
        stage = torch.zeros((5, C), dtype=torch.float32, device=device)
        stage[0].fill_(1.0)       # weight 1, bias 0 -> a zero channel stays exactly zero (mean 0, variance 0)
        stage[4].fill_(1.0)       # running variance of a padding lane (never read back)
        holder._pad_stage = stage
    return stage


class _ZeroGradRider(torch.autograd.Function):
    """Identity on ``out``; its backward hands every ``param`` a slice of ONE zero-filled buffer as gradient (see ZeroGradScope).
    Two consequences of the form (both stated rather than worked around): the output is a view created inside a custom
    Function, so an IN-PLACE op on the network's first gradient-carrying output raises autograd's usual error for such views (nnU-Net's
    losses do not modify logits in place); and the zero gradients reach the folded conv biases only when the loss depends on that output —
    a loss built from the other heads alone leaves them ``grad = None`` (weight decay / momentum then skip them for that step, where the
    reference would apply them to a round-off-sized gradient)."""

    @staticmethod
    def forward(ctx, out, *params):
        ctx.meta = [(tuple(p.shape), p.dtype, p.device) for p in params]
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        pools, grads = {}, []
        for shape, dtype, device in ctx.meta:
            key = (dtype, device)
            pools[key] = pools.get(key, 0) + int(math.prod(shape))
        bufs = {key: torch.zeros((n,), dtype=key[0], device=key[1]) for key, n in pools.items()}
        offs = {key: 0 for key in pools}
        for shape, dtype, device in ctx.meta:
            key, n = (dtype, device), int(math.prod(shape))
            grads.append(bufs[key].narrow(0, offs[key], n).view(shape))
            offs[key] += n
        return (g,) + tuple(grads)


class ZeroGradScope(threading.local):
    """(State per thread: replicas driven by threads — nn.DataParallel — each see their own scope.)
    The bias of a convolution folded into a batch- / instance-statistics norm has gradient exactly zero (the statistics absorb a
    per-channel constant); the optimizer must still SEE a zero gradient — weight decay and momentum act on the parameter as they do
    in the reference, where the gradient is round-off around zero.  Producing that zero per norm is one fill kernel each: 85 launches,
    0.38 ms of the cfg-2 step (profiles/r05_aten_glue.md).  Inside ``with scope:`` (the network's training forward) the norms hand
    the bias to their kernels detached and register the parameter here; :meth:`attach` then ties all registered parameters to the
    network's output through one identity node whose backward zero-fills ONE buffer and returns its slices.  Outside a scope the
    norms keep the per-norm ``zeros_like``."""

    def __init__(self):
        self.active = False
        self._params = {}

    def __enter__(self):
        self.active = True
        self._params = {}
        return self

    def __exit__(self, *exc):
        self.active = False
        self._params = {}
        return False

    def take(self, param, statistics_from_input: bool):
        """what the norm passes to its autograd function as ``pre_bias``"""
        if (self.active and param is not None and statistics_from_input and torch.is_grad_enabled() and param.requires_grad
                and param.is_leaf):
            self._params[id(param)] = param
            return param.detach()
        return param

    def attach(self, outputs):
        """``outputs`` (tensor, or list / tuple of tensors) with the first gradient-carrying one routed through the rider node"""
        params = list(self._params.values())
        self._params = {}
        if not params or not torch.is_grad_enabled():
            return outputs
        seq = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        for i, t in enumerate(seq):
            if isinstance(t, torch.Tensor) and t.requires_grad:
                seq[i] = _ZeroGradRider.apply(t, *params)
                break
        else:
            return outputs
        if isinstance(outputs, (list, tuple)):
            return type(outputs)(seq)
        return seq[0]


ZERO_GRADS = ZeroGradScope()


def norm_act(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
             running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], training: bool,
             momentum: float, eps: float, negative_slope: float = 1.0, instance: bool = False,
             pre_bias: Optional[torch.Tensor] = None, pad_holder=None, stats_partial: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``leaky_relu(batch_norm(x [+ pre_bias]) | instance_norm(x), negative_slope)`` on (B,C,*spatial) fp32 / bf16 / fp16
    (statistics and the normalisation itself are always computed in fp32 / fp64; only loads and stores are narrow).

    ``pre_bias`` (C,) is the bias of the convolution that produced ``x`` when that convolution was run without it:
    a per-channel constant does not survive batch statistics, so it only has to enter the running mean (training) or
    the shift (inference) — the conv's bias-gradient reduction over the whole tensor disappears with it.

    The fused replacement of ``norm -> act`` in the reference's BasicConv (torch_nn.py:84-90), FFN / fc1 / fc2
    (NexToU_Encoder_Decoder.py:384-390, 710-720, 833-842) and the conv stages' ConvDropoutNormReLU.
    ``negative_slope=1`` is the bare normalisation.  Running statistics are updated in place as
    ``F.batch_norm`` does (unbiased variance, ``momentum``); ``training=False`` uses them.

    ``pad_holder``: the norm module, when ``x`` carries more channels than its ``num_features`` — the extra ones being the
    all-zero padding channels of network_architecture/channel_pad.py; parameters and statistics keep their real length.
    """
    if x.dtype not in _NORM_DTYPES:
        raise TypeError("norm_act: dtype %s not in (float32, bfloat16, float16)" % x.dtype)
    if instance and not training:
        raise ValueError("norm_act: instance norm always uses the statistics of its input")
    if weight is not None and weight.dtype != torch.float32:
        weight = weight.float()
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    if pre_bias is not None and pre_bias.dtype != torch.float32:
        pre_bias = pre_bias.float()
    if pad_holder is not None and instance:
        raise ValueError("norm_act: channel padding is only defined for batch statistics")
    if stats_partial is not None and not (instance and x.is_cuda and x.dtype == torch.float32 and pad_holder is None):
        raise ValueError("norm_act: ready-made statistics are for fp32 instance norm on the device")
    pre_bias = ZERO_GRADS.take(pre_bias, bool(training))       # (instance norm: training is always True)
    return _NormAct.apply(x, weight, bias, running_mean, running_var, bool(training), float(momentum),
                          float(eps), float(negative_slope), bool(instance), pre_bias, pad_holder, stats_partial)


def _dense_channels_last(x: torch.Tensor):
    """torch.channels_last(_3d) when ``x`` is stored (B,*spatial,C)-contiguous with C > 1 and is NOT also
    (B,C,*spatial)-contiguous; ``None`` otherwise (C = 1 and 1x1.. tensors are the same bytes in both layouts)."""
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}.get(x.dim())
    if mf is None or x.shape[1] == 1 or x.is_contiguous() or not x.is_contiguous(memory_format=mf):
        return None
    return mf


class _ConvOwnBiasGrad(torch.autograd.Function):
    """``aten.convolution`` whose bias gradient is K6's per-channel sum instead of ATen's generic reduction (which
    collapses on channels-last tensors: 5.5 ms for the 727 MB gradient of cfg 2's full-resolution stage)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, transposed, output_padding, groups):
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x, weight = x.to(dt), weight.to(dt)
            bias_c = None if bias is None else bias.to(dt)
        else:
            bias_c = bias
        with torch.autocast("cuda", enabled=False):
            y = torch.ops.aten.convolution(x, weight, bias_c, stride, padding, dilation, transposed, output_padding, groups)
        ctx.save_for_backward(x, weight)
        ctx.conf = (stride, padding, dilation, transposed, output_padding, groups)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, dilation, transposed, output_padding, groups = ctx.conf
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        # the gradient takes the layout of the convolution's INPUT: what arrives may be a channel slice of a
        # concatenation's gradient (dense in neither layout), and converting that to NCDHW under a channels-last
        # convolution costs two passes over the tensor instead of one
        cl = _dense_channels_last(x)
        gy = gy.contiguous(memory_format=cl) if cl is not None else gy.contiguous()
        gx, gw, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, stride, padding, dilation, transposed,
                                                        output_padding, groups,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        gb = None
        if ctx.needs_input_grad[2]:
            gb = _backend_for(gy).channel_sum(gy, channels_last=cl is not None)
        return gx, gw, gb, None, None, None, None, None, None


def _filter_flip_own() -> bool:
    """NEXTOU_FILTER_FLIP=0: ATen's transpose -> flip -> contiguous instead of nextou_filter_flip_t (A/B)."""
    import os
    return os.environ.get("NEXTOU_FILTER_FLIP", "1") != "0"


class _ConvDgradAsForward(torch.autograd.Function):
    """Stride-1 'same' convolution whose data gradient is computed as a FORWARD convolution of the output gradient with
    the flipped, transposed filter — the same numbers in another summation order.  MIOpen's CK forward kernels are
    markedly faster than its backward-data kernels on these shapes (MI355X, NDHWC fp32, tools/conv_probe.py /
    profiles/r02_conv_evidence_padding_ab.md: 40 -> 40 at 64x224x192 dgrad 3.24 ms vs forward 2.32 ms; 72 -> 72 at
    64x112x96 7.28 vs 5.18 ms), and the library stays MIOpen either way.  The weight gradient is the library's own."""

    @staticmethod
    def forward(ctx, x, weight, padding):
        n = weight.dim() - 2
        ones, zeros = (1,) * n, (0,) * n
        y = torch.ops.aten.convolution(x, weight, None, ones, padding, ones, False, zeros, 1)
        ctx.save_for_backward(x, weight)
        ctx.padding = padding
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        n = weight.dim() - 2
        ones, zeros = (1,) * n, (0,) * n
        cl = _dense_channels_last(x)
        gy = gy.contiguous(memory_format=cl) if cl is not None else gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if cl is not None and weight.is_cuda and weight.dtype == torch.float32 and gy.dtype == torch.float32 and n in (2, 3) and \
                    _filter_flip_own():
                wt = _HIP.filter_flip_t(weight.detach())                        # one launch: transpose + flip + channels-last
            else:
                taps = [d for d in range(2, 2 + n) if weight.shape[d] > 1]      # (a flip over size-1 axes is only a copy)
                wt = weight.transpose(0, 1)
                if taps:
                    wt = wt.flip(*taps)
                wt = wt.contiguous(memory_format=cl) if cl is not None else wt.contiguous()
            gx = torch.ops.aten.convolution(gy, wt, None, ones, ctx.padding, ones, False, zeros, 1)
        if ctx.needs_input_grad[1] and wgrad_depth_unroll_eligible(x, weight, ctx.padding):
            # [3,3,3] kernel: the depth taps become input channels and the weight gradient a 2-D problem (MIOpen's 2-D kernels:
            # 72 -> 72 at 64x112x96 6.25 -> 5.09 ms, 144 -> 72 11.9 -> 10.5 ms, profiles/r02_conv_depth_unroll_probe.md)
            co, ci = weight.shape[:2]
            x3 = _HIP.depth_unroll(x)
            w2 = weight.permute(0, 2, 1, 3, 4).reshape(co, 3 * ci, 3, 3)
            _, gw2, _ = torch.ops.aten.convolution_backward(flat_depth(gy), x3, w2, None, (1, 1), tuple(ctx.padding[1:]), (1, 1), False,
                                                            (0, 0), 1, [False, True, False])
            gw = gw2.reshape(co, 3, ci, 3, 3).permute(0, 2, 1, 3, 4)
        elif ctx.needs_input_grad[1]:
            _, gw, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, ones, ctx.padding, ones, False, zeros, 1,
                                                           [False, True, False])
        return gx, gw, None


HEAD_ROWS_DEFAULT = "1"


class _HeadConv(torch.autograd.Function):
    """A segmentation head — biased 1x1 convolution from the stage's features to the class logits (reference
    NexToU_Encoder_Decoder.py:253-258, :311-337) — on K8 (csrc/head_rows.hip): forward, data gradient and weight + bias gradient
    over the channels-last rows, no library convolution involved.  (MIOpen's backward of exactly this convolution is where the
    averaged N > 1 eager step took its GPU memory fault, profiles/r05_n_gt_1.md.)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        w2 = weight.reshape(weight.shape[0], weight.shape[1]).contiguous()
        y = _HIP.head_rows_fwd(x, w2, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous(memory_format={4: torch.channels_last, 5: torch.channels_last_3d}[gy.dim()])
        want_gw = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        gx, gw, gb = _HIP.head_rows_bwd(gy, x, weight.reshape(weight.shape[0], weight.shape[1]).contiguous(), ctx.needs_input_grad[0],
                                        want_gw)
        if gw is not None:
            gw = gw.reshape(weight.shape)
        return gx, (gw if ctx.needs_input_grad[1] else None), (gb if ctx.has_bias and ctx.needs_input_grad[2] else None)


def head_rows_eligible(conv: torch.nn.Module, x: torch.Tensor, weight: torch.Tensor) -> bool:
    """An un-grouped kernel-1 / stride-1 / unpadded convolution of a dense channels-last fp32 device volume outside autocast to at most
    64 output channels — the segmentation heads.  ``NEXTOU_HEAD_ROWS=0`` hands them back to the library convolution (A/B)."""
    import os
    if os.environ.get("NEXTOU_HEAD_ROWS", HEAD_ROWS_DEFAULT) == "0":
        return False
    if not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
        return False
    if x.dim() not in (4, 5) or conv.transposed or conv.groups != 1 or isinstance(conv.padding, str):
        return False
    if any(k != 1 for k in weight.shape[2:]) or any(v != 1 for v in conv.stride) or any(v != 0 for v in conv.padding) or \
            any(v != 1 for v in conv.dilation):
        return False
    if weight.shape[0] > 64 or x.shape[1] != weight.shape[1] or weight.shape[1] > 1000:
        return False
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}[x.dim()]
    return x.is_contiguous(memory_format=mf) and x.shape[1] > 1


def head_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    return _HeadConv.apply(x, weight, bias)


class _StemBlock(torch.autograd.Function):
    """The network's first ConvDropoutNormReLU — conv(1 -> C, [1,]3x3) -> BatchNorm -> LeakyReLU (reference
    NexToU_Encoder_Decoder.py:125-141, encoder.stages[0]) — on K9 (csrc/stem_conv.hip): the batch statistics come from the nine-tap
    moments of the image, the output rows are written once, and the backward is one pass over the incoming gradient; the convolution's
    881-MB output (cfg 2) is never stored.  Saves the image, 2C + 54 numbers and half a mask byte per voxel and channel quad (28 MB at cfg 2, dwords of 8 rows x 4 channels)."""

    @staticmethod
    def forward(ctx, x, weight, conv_bias, gamma, beta, running_mean, running_var, training, momentum, eps, slope, c_pad):
        w2 = weight.detach().reshape(weight.shape[0], 9).contiguous()
        need_bwd = training and any(t is not None and t.requires_grad for t in (weight, conv_bias, gamma, beta))
        y, mean, invstd, moments, act = _HIP.stem_fwd(x, w2, conv_bias, gamma, beta, running_mean, running_var, training, momentum, eps, slope, c_pad,
                                                      want_mask=need_bwd)
        if need_bwd:
            ctx.save_for_backward(x, w2, gamma, beta, mean, invstd, moments, act)
        ctx.cfg = (bool(training), float(slope), tuple(weight.shape), conv_bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        training, slope, wshape, conv_bias = ctx.cfg
        if not training:
            raise RuntimeError("stem block (K9): the backward exists for batch statistics only; graph_ops.stem_block_eligible routes an "
                               "eval-mode block whose parameters need gradients through the op-by-op modules")
        x, w2, gamma, beta, mean, invstd, moments, act = ctx.saved_tensors
        gy = gy.contiguous(memory_format={4: torch.channels_last, 5: torch.channels_last_3d}[gy.dim()])
        if gy.dtype != torch.float32:
            gy = gy.float()
        need = ctx.needs_input_grad
        gw, gg, gb = _HIP.stem_bwd(x, gy, act, w2, gamma, mean, invstd, moments, slope, need[1], gamma is not None and need[3],
                                   beta is not None and need[4])
        gbias = torch.zeros_like(conv_bias) if (conv_bias is not None and need[2]) else None      # batch statistics absorb a per-channel constant
        return (None, None if gw is None else gw.reshape(wshape), gbias, gg, gb, None, None, None, None, None, None, None)


STEM_BLOCK_DEFAULT = "1"


def stem_block_eligible(conv: torch.nn.Module, norm: torch.nn.Module, x: torch.Tensor) -> bool:
    """The first block of the network on K9: a one-channel fp32 device image in the channels-last stage layout (stride 1 on its channel
    axis, layout.to_channels_last), outside autocast, that needs no gradient itself; a 1 -> C convolution with kernel [1,]3x3, stride 1,
    zero padding [0,]1,1 whose bias is folded into a fused BATCH norm directly behind it; C (after the internal channel padding) a
    multiple of 4 up to 48.  Batch statistics (training), or running statistics when nothing needs a gradient.
    ``NEXTOU_STEM_BLOCK=0`` hands the block back to the library convolution + K6 (A/B)."""
    import os
    from .network_architecture import norm_act as na
    if os.environ.get("NEXTOU_STEM_BLOCK", STEM_BLOCK_DEFAULT) == "0":
        return False
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
        return False
    if x.dim() not in (4, 5) or x.shape[1] != 1 or x.stride(1) != 1 or (x.requires_grad and torch.is_grad_enabled()):
        return False
    if not isinstance(conv, (torch.nn.Conv2d, torch.nn.Conv3d)) or not isinstance(norm, na._BatchNormAct) or conv.transposed:
        return False
    nd = x.dim() - 2
    want_k = (3, 3) if nd == 2 else (1, 3, 3)
    want_p = (1, 1) if nd == 2 else (0, 1, 1)
    if conv.in_channels != 1 or conv.groups != 1 or tuple(conv.kernel_size) != want_k or isinstance(conv.padding, str) or \
            tuple(conv.padding) != want_p or any(v != 1 for v in conv.stride) or any(v != 1 for v in conv.dilation) or \
            conv.padding_mode != "zeros" or conv.weight.dtype != torch.float32:
        return False
    src = getattr(norm, "_pre_bias_src", None)
    if conv.bias is not None and (src is None or src[0] is not conv):       # a bias that is NOT folded into the norm: not this block
        return False
    if norm.num_features != conv.out_channels or x.numel() == 0 or not x[:, 0].is_contiguous():     # (batch and spatial axes dense, row-major)
        return False
    use_batch_stats = norm.training or (norm.running_mean is None and norm.running_var is None)
    if not use_batch_stats and torch.is_grad_enabled() and any(p.requires_grad for p in list(conv.parameters()) + list(norm.parameters())):
        return False
    rows = x.numel() // max(int(x.shape[-1]), 1)
    return 0 < rows < 2 ** 31


def stem_block(x: torch.Tensor, conv: torch.nn.Module, norm: torch.nn.Module) -> torch.Tensor:
    """``norm(conv(x))`` of the stem block through K9 (see :func:`stem_block_eligible`); ``None`` when the channel padding decided for
    this call gives a channel count the kernels do not take (the caller then runs the modules)."""
    from .network_architecture import channel_pad as cp
    from .network_architecture import norm_act as na
    C = conv.out_channels
    c_pad = C
    if getattr(conv, "_pad_spec", None) is not None:
        _, pad_out = cp.conv_pad_plan(conv, x)          # the entry module's decision (and the shared regime) exactly as the op-by-op path makes it
        if pad_out:
            c_pad = cp.padded(C, conv._pad_spec.multiple)
    if c_pad % 4 or c_pad > 48:
        return None
    if norm.training:
        na._verify_batch_size(x)            # (one input channel: the same count of values per channel as the convolution's output)
    use_batch_stats, factor, keep_running = norm._step()
    pre_bias = ZERO_GRADS.take(conv.bias, bool(use_batch_stats))
    return _StemBlock.apply(x, conv.weight, pre_bias, norm.weight, norm.bias, norm.running_mean if keep_running else None,
                            norm.running_var if keep_running else None, bool(use_batch_stats), float(factor), float(norm.eps),
                            float(norm.negative_slope), int(c_pad))


PW_GEMM_DEFAULT = "0"


def _pw_mode() -> str:
    """NEXTOU_PW_GEMM: "0" MIOpen, "1" K7 for forward / data gradient / weight gradient, "wgrad" K7 for the weight gradient only."""
    import os
    return os.environ.get("NEXTOU_PW_GEMM", PW_GEMM_DEFAULT)


class _PointwiseConv(torch.autograd.Function):
    """Kernel-1 convolution of a dense channels-last fp32 volume on K7 (csrc/pw_gemm.hip): the forward and the data
    gradient are the same GEMM kernel over the (points, channels) rows (the latter with the per-group transposed weight),
    the weight gradient the split-over-points kernel; the bias gradient (rare: these convolutions usually run bias-free
    in front of a fused norm) is K6's channel sum."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups):
        ctx.own_rows = _pw_mode() != "wgrad"
        if ctx.own_rows:
            w2 = weight.reshape(weight.shape[0], weight.shape[1])
            y = _HIP.pw_rows(x, w2.contiguous(), bias, groups)
        else:        # A/B mode: MIOpen forward / data gradient (depth-flat 2-D view), own weight gradient
            y = unflat_depth(torch.nn.functional.conv2d(flat_depth(x), weight.squeeze(2), bias, groups=groups), x.shape[0], x.shape[2]) \
                if x.dim() == 5 else torch.nn.functional.conv2d(x, weight, bias, groups=groups)
        ctx.save_for_backward(x, weight)
        ctx.groups, ctx.has_bias = groups, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        g = ctx.groups
        n, k = weight.shape[0] // g, weight.shape[1]
        gy = gy.contiguous(memory_format={4: torch.channels_last, 5: torch.channels_last_3d}[gy.dim()])
        gx = gw = gb = None
        if ctx.needs_input_grad[0] and ctx.own_rows:
            wt = weight.reshape(g, n, k).transpose(1, 2).reshape(g * k, n).contiguous()
            gx = _HIP.pw_rows(gy, wt, None, g)
        elif ctx.needs_input_grad[0]:
            nd = weight.dim() - 2
            gx, _, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, (1,) * nd, (0,) * nd, (1,) * nd, False, (0,) * nd, g,
                                                           [True, False, False])
        if ctx.needs_input_grad[1]:
            gw = _HIP.pw_wgrad(gy, x, g).reshape(weight.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _HIP.channel_sum(gy, channels_last=True)
        return gx, gw, gb, None


def pointwise_eligible(conv: torch.nn.Module, x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Kernel-1 / stride-1 / unpadded convolution of a dense channels-last fp32 device volume outside autocast whose
    per-group channel counts are multiples of 4 (every 1x1 convolution of the Grapher / FFN blocks; not the 14-class
    heads).  Opt-in with ``NEXTOU_PW_GEMM=1`` while MIOpen's 2-D kernels are as fast on these shapes
    (profiles/r02_pw_gemm.md)."""
    if _pw_mode() == "0":
        return False
    if not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
        return False
    if conv.transposed or isinstance(conv.padding, str) or _dense_channels_last(x) is None:
        return False
    if any(k != 1 for k in weight.shape[2:]) or any(s != 1 for s in conv.stride) or any(p != 0 for p in conv.padding) or \
            any(d != 1 for d in conv.dilation):
        return False
    g = conv.groups
    return weight.shape[0] % g == 0 and (weight.shape[0] // g) % 4 == 0 and weight.shape[1] % 4 == 0 and \
        x.shape[1] == g * weight.shape[1]


def pointwise_conv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], groups: int) -> torch.Tensor:
    return _PointwiseConv.apply(x, weight, bias, int(groups))


PW_FUSE_DEFAULT = "1"


def _pw_fuse_mode() -> str:
    """NEXTOU_PW_FUSE: "0" op-by-op blocks (MIOpen / K7 convolutions + K6 passes), "1" (default) the fused point-wise pipeline with the
    gradient statistics in the data-gradient GEMM's epilogue, "fwd" the fused forward with K6's own backward reduce (A/B)."""
    import os
    return os.environ.get("NEXTOU_PW_FUSE", PW_FUSE_DEFAULT)


def _pw_fuse_min_points() -> int:
    """Point count from which the fused pipeline replaces the op-by-op block.  Measured on MI355X, cfg 2 (profiles/r03_pw_fused.md):
    at stage 2 (172 032 points) the fused FFN is 2.44 ms forward + backward against 2.75 ms op by op; at stages 3-5 (21 504 ... 336
    points) K7's GEMMs run far from full (a 128-point workgroup tile per CU or less) and MIOpen's kernels + K6 win.
    ``NEXTOU_PW_FUSE_MIN_POINTS`` overrides the threshold (tests set 0)."""
    import os
    return int(os.environ.get("NEXTOU_PW_FUSE_MIN_POINTS", "65536"))


class _NormState:
    """What one fused norm of a point-wise chain needs besides its parameters (plain Python, not a tensor argument)."""
    __slots__ = ("batch_stats", "momentum", "eps", "slope", "running_mean", "running_var")

    def __init__(self, batch_stats, momentum, eps, slope, running_mean, running_var):
        self.batch_stats, self.momentum, self.eps, self.slope = bool(batch_stats), float(momentum), float(eps), float(slope)
        self.running_mean, self.running_var = running_mean, running_var


class _PointwiseChain(torch.autograd.Function):
    """``x -> conv1x1 (groups) -> norm -> LeakyReLU [-> conv1x1 -> norm -> LeakyReLU] [+ residual]`` on a dense channels-last fp32
    volume as ONE pipeline of K7 GEMMs with K6 folded into them (csrc/pw_gemm.hip, include/nextou_hip.h "K7 + K6 fused"):

    forward   GEMM1 + statistics epilogue -> finalize -> [GEMM2 with norm + activation in the operand load + statistics epilogue ->
              finalize ->] apply (+ residual).  The activated hidden tensor is never written; K6's statistics passes never run.
    backward  K6 backward of the last norm -> [data-gradient GEMM with the hidden norm's gradient statistics in its epilogue ->
              finalize -> apply; weight-gradient GEMM re-creating the activated operand on load ->] data / weight gradient of conv1.

    The reference runs this as conv -> batch_norm -> leaky_relu -> conv -> batch_norm -> add (NexToU_Encoder_Decoder.py:368-390 FFN,
    :710-720 / :833-842 fc1 / fc2, torch_nn.py:66-92 BasicConv); same arithmetic per element as the op-by-op K7 / K6 path (the
    normalisation is K6's fmaf, the statistics are float64 sums of the same values in another order)."""

    @staticmethod
    def forward(ctx, x, residual, w1, g1, b1, cb1, w2, g2, b2, cb2, groups1, n1, n2, fuse_bwd, pre_h=None, pre_part=None):
        dev = x.device
        P = x.numel() // x.shape[1]
        c1 = w1.shape[0]
        ctx.pre = pre_h is not None
        if ctx.pre:     # GEMM1's output and statistics partials come from the kernel that produced x (mr_grouped_chain): that kernel
            h, part1 = pre_h, pre_part      # owns GEMM1's data gradient too — the gradient goes to pre_h (dh), none to x
        else:
            w1m = w1.reshape(c1, w1.shape[1]).contiguous()
            h, part1 = _HIP.pw_rows_fused(x, w1m, groups1, want_stats=n1.batch_stats)
        m1, i1, sc1, sh1 = _HIP.norm_finalize(part1, P, c1, dev, g1, b1, cb1, n1.running_mean, n1.running_var, n1.batch_stats,
                                              n1.momentum, n1.eps)
        if w2 is not None:
            c2 = w2.shape[0]
            w2m = w2.reshape(c2, w2.shape[1]).contiguous()
            y, part2 = _HIP.pw_rows_fused(h, w2m, 1, pro=(sc1, sh1, n1.slope), want_stats=n2.batch_stats)
            m2, i2, _, _ = _HIP.norm_finalize(part2, P, c2, dev, g2, b2, cb2, n2.running_mean, n2.running_var, n2.batch_stats,
                                              n2.momentum, n2.eps)
            out = _HIP.norm_apply_rows(y, residual, g2, b2, m2, i2, n2.slope)
            ctx.save_for_backward(x, h, y, w1, w2, g1, b1, g2, b2, m1, i1, sc1, sh1, m2, i2)
        else:
            out = _HIP.norm_apply_rows(h, residual, g1, b1, m1, i1, n1.slope)
            ctx.save_for_backward(x, h, w1, g1, b1, m1, i1)
        ctx.conf = (groups1, n1, n2, residual is not None, fuse_bwd, w2 is not None, cb1, cb2)
        return out

    @staticmethod
    def backward(ctx, g):
        groups1, n1, n2, has_res, fuse_bwd, two, cb1, cb2 = ctx.conf
        mf = {4: torch.channels_last, 5: torch.channels_last_3d}[g.dim()]
        g = g.contiguous(memory_format=mf)
        gw2 = gg2 = gb2 = gcb2 = None
        if two:
            x, h, y, w1, w2, g1, b1, g2, b2, m1, i1, sc1, sh1, m2, i2 = ctx.saved_tensors
            P = x.numel() // x.shape[1]
            c1, c2 = w1.shape[0], w2.shape[0]
            dy, gg2, gb2 = _HIP.norm_act_bwd(y, g, g2, b2, m2, i2, n2.batch_stats, n2.slope, 0, channels_last=True)
            w2t = w2.reshape(c2, c1).t().contiguous()
            if fuse_bwd and n1.batch_stats:
                da, part = _HIP.pw_rows_fused(dy, w2t, 1, bwd=(h, g1, b1, m1, i1, n1.slope))
                coeff, gg1, gb1 = _HIP.norm_bwd_finalize(part, P, c1, g.device, True)
                dh = _HIP.norm_bwd_apply_rows(h, da, coeff, g1, b1, m1, i1, n1.slope)
            else:
                da = _HIP.pw_rows(dy, w2t, None, 1)
                dh, gg1, gb1 = _HIP.norm_act_bwd(h, da, g1, b1, m1, i1, n1.batch_stats, n1.slope, 0, channels_last=True)
            del da
            if ctx.needs_input_grad[6]:
                gw2 = _HIP.pw_wgrad_fused(dy, h, 1, (sc1, sh1, n1.slope)).reshape(w2.shape)
            if cb2 is not None and ctx.needs_input_grad[9]:
                gcb2 = torch.zeros_like(cb2) if n2.batch_stats else (gb2 * i2 * (g2 if g2 is not None else 1.0))
        else:
            x, h, w1, g1, b1, m1, i1 = ctx.saved_tensors
            c1 = w1.shape[0]
            dh, gg1, gb1 = _HIP.norm_act_bwd(h, g, g1, b1, m1, i1, n1.batch_stats, n1.slope, 0, channels_last=True)
        gx = gw1 = gcb1 = None
        n, k = c1 // groups1, w1.shape[1]
        if ctx.needs_input_grad[0] and not ctx.pre:
            w1t = w1.reshape(groups1, n, k).transpose(1, 2).reshape(groups1 * k, n).contiguous()
            gx = _HIP.pw_rows(dh, w1t, None, groups1)
        if ctx.needs_input_grad[2]:
            gw1 = _HIP.pw_wgrad(dh, x, groups1).reshape(w1.shape)
        if cb1 is not None and ctx.needs_input_grad[5]:
            gcb1 = torch.zeros_like(cb1) if n1.batch_stats else (gb1 * i1 * (g1 if g1 is not None else 1.0))
        return (gx, g if has_res else None, gw1, gg1 if ctx.needs_input_grad[3] else None, gb1 if ctx.needs_input_grad[4] else None, gcb1,
                gw2, gg2 if ctx.needs_input_grad[7] else None, gb2 if ctx.needs_input_grad[8] else None, gcb2, None, None, None, None,
                dh if ctx.pre else None, None)


def pointwise_chain(x, residual, conv1, norm1, conv2=None, norm2=None):
    """Fused ``norm2(conv2(norm1(conv1(x)))) [+ residual]`` (each norm with its absorbed LeakyReLU) when every piece qualifies — see
    :func:`pointwise_chain_eligible` — else ``None``.  ``conv*``: the 1x1 convolution modules (bias folded into the norm or absent),
    ``norm*``: the fused ``BatchNormAct`` modules behind them."""
    mode = _pw_fuse_mode()
    if mode == "0" or not pointwise_chain_eligible(x, residual, conv1, norm1, conv2, norm2):
        return None
    n1 = _norm_state(norm1)
    n2 = _norm_state(norm2) if norm2 is not None else None
    cb1 = ZERO_GRADS.take(conv1.bias, n1.batch_stats)
    cb2 = ZERO_GRADS.take(conv2.bias, n2.batch_stats) if conv2 is not None else None
    res = None if residual is None else as_channels_last_rows(residual)
    return _PointwiseChain.apply(x, res, conv1.weight, norm1.weight, norm1.bias, cb1,
                                 None if conv2 is None else conv2.weight, None if norm2 is None else norm2.weight,
                                 None if norm2 is None else norm2.bias, cb2, int(conv1.groups), n1, n2, mode != "fwd")


class _MRAggregateRows(torch.autograd.Function):
    """K2 + K7 (csrc/mr_aggregate.hip): max-relative aggregation of Swin windows, window reverse and MRConv's grouped 1x1 convolution in
    one launch.  Returns ``(a, h)``: the aggregate as a channels-last volume (``window_scatter(mr_aggregate(windows))``; data for the
    weight gradient, not differentiable here) and the convolution's output; the statistics partials of ``h`` go to ``box``.
    Backward (gradient of ``h`` -> gradient of ``windows``): one launch as well; NEXTOU_MR_GROUPED_BWD=0 or an unsupported shape runs
    the three ops' own backwards (grouped data-gradient GEMM, window gather, arg-tape scatter)."""

    @staticmethod
    def forward(ctx, windows, nn_idx, K, idx_step, w1m, groups, batch, spatial, window, shift, want_stats, box):
        need_x = windows.requires_grad
        a, arg, h, part = _HIP.mr_grouped_rows(windows, nn_idx, K, idx_step, w1m, groups, batch, spatial, window, shift,
                                               want_a=True, want_arg=need_x, want_stats=want_stats)
        box["part"] = part
        ctx.conf = (window, shift, tuple(spatial), windows.shape[2], groups)
        if need_x:
            ctx.save_for_backward(arg, w1m)
        ctx.mark_non_differentiable(a)
        return a, h

    @staticmethod
    def backward(ctx, _ga, dh):
        import os
        window, shift, spatial, Nw, groups = ctx.conf
        arg, w1m = ctx.saved_tensors
        mf = {4: torch.channels_last, 5: torch.channels_last_3d}[dh.dim()]
        dh = dh.contiguous(memory_format=mf)
        if os.environ.get("NEXTOU_MR_GROUPED_BWD", "1") != "0":
            dx = _HIP.mr_grouped_rows_bwd(dh, w1m, arg, groups, spatial, window, shift)
        else:
            n, k = w1m.shape[0] // groups, w1m.shape[1]
            w1t = w1m.reshape(groups, n, k).transpose(1, 2).reshape(groups * k, n).contiguous()
            ga = _HIP.pw_rows(dh, w1t, None, groups)
            dx, _ = _HIP.mr_bwd_arg(_HIP.window_gather(ga, window, shift), arg, Nw, False)
        return (dx,) + (None,) * 11


class _MRGroupedConv(torch.autograd.Function):
    """``conv1x1(window_scatter(mr_aggregate(windows)), weight, groups)`` as the K2 + K7 launch, for the stages whose blocks do not take
    the fused point-wise chain (below NEXTOU_PW_FUSE_MIN_POINTS): returns the convolution's output as a channels-last volume; backward =
    the fused backward launch for the window tensor + K7's weight-gradient GEMM on the saved aggregate rows."""

    @staticmethod
    def forward(ctx, windows, weight, nn_idx, K, groups, batch, spatial, window, shift):
        w2 = weight.reshape(weight.shape[0], weight.shape[1]).contiguous()
        need_x, need_w = windows.requires_grad, weight.requires_grad
        a, arg, h, _ = _HIP.mr_grouped_rows(windows, nn_idx, K, 1, w2, groups, batch, spatial, window, shift,
                                            want_a=need_w, want_arg=need_x, want_stats=False)
        ctx.conf = (window, shift, tuple(spatial), groups, tuple(weight.shape))
        ctx.save_for_backward(arg if need_x else None, a if need_w else None, w2 if need_x else None)
        return h

    @staticmethod
    def backward(ctx, dh):
        window, shift, spatial, groups, wshape = ctx.conf
        arg, a, w2 = ctx.saved_tensors
        mf = {4: torch.channels_last, 5: torch.channels_last_3d}[dh.dim()]
        dh = dh.contiguous(memory_format=mf)
        dx = _HIP.mr_grouped_rows_bwd(dh, w2, arg, groups, spatial, window, shift) if ctx.needs_input_grad[0] else None
        gw = _HIP.pw_wgrad(dh, a, groups).reshape(wshape) if ctx.needs_input_grad[1] else None
        return (dx, gw) + (None,) * 7


class _MRGroupedCM(torch.autograd.Function):
    """MRConv of a pooled / self graph up to its grouped 1x1 convolution in ONE launch (K2 + K7, channel-major half: csrc/mr_aggregate.hip
    mr_grp_cm_kernel): ``conv1x1(mr_aggregate(x, nn_idx, y), weight, groups)`` with the InstanceNorm (sum, sum of squares) partials of
    the result.  Backward = the three ops' own backwards: the two strided-batched GEMMs of the grouped convolution (data gradient,
    weight gradient on the saved aggregate) and the fixed-point arg-tape scatter (mr_bwd_fix_kernel)."""

    @staticmethod
    def forward(ctx, x, y, nn_idx, weight, K, idx_step, groups, want_stats):
        w2 = weight.reshape(weight.shape[0], weight.shape[1]).contiguous()
        need_x = x.requires_grad or (y is not None and y.requires_grad)
        need_w = weight.requires_grad
        a, arg, h, partial = _HIP.mr_grouped_cm(x, y, nn_idx, K, idx_step, w2, groups, want_a=need_w, want_arg=need_x, want_stats=want_stats)
        ctx.conf = (groups, tuple(weight.shape), x.shape[2] if y is None else y.shape[2], y is not None)
        ctx.save_for_backward(arg if need_x else None, a if need_w else None, w2 if need_x else None)
        if partial is None:
            partial = h.new_empty(0)
        ctx.mark_non_differentiable(partial)
        return h, partial

    @staticmethod
    def backward(ctx, dh, _):
        groups, wshape, M, has_y = ctx.conf
        arg, a, w2 = ctx.saved_tensors
        B, _, N = dh.shape
        dh4 = dh.contiguous().view(B, groups, -1, N)
        dx = dy = gw = None
        if arg is not None:
            ga = torch.matmul(w2.view(groups, dh4.shape[2], -1).transpose(1, 2), dh4)          # (B, groups, 2C / groups, N)
            dx, dy = _HIP.mr_bwd_arg(ga.view(B, -1, N), arg, M, has_y)
        if a is not None:
            gw = torch.matmul(dh4, a.view(B, groups, -1, N).transpose(2, 3)).sum(0).reshape(wshape)
        return dx, dy, None, gw, None, None, None, None


def mr_grouped_cm_block(x, nn_idx, y, conv, norm):
    """MRConv of a Pool-GNN block — ``norm(conv(mr_aggregate(x, nn_idx, y)))`` with ``norm`` = InstanceNorm + the absorbed LeakyReLU —
    as the K2 + K7 launch followed by K6's apply on the launch's statistics partials (reference
    NexToU_Encoder_Decoder.py:401-418 inside :516-551, torch_nn.py:66-92), or ``None`` when a piece does not qualify (the caller runs
    mr_aggregate -> conv -> norm).  x (B, C, N), y (B, C, M) | None, nn_idx (B, N, K) int32 -> (B, 2C', N)."""
    import os
    if os.environ.get("NEXTOU_MR_GROUPED_CM", "1") == "0" or not x.is_cuda or x.dtype != torch.float32 or \
            torch.is_autocast_enabled("cuda") or nn_idx.dtype != torch.int32 or x.dim() != 3:
        return None
    from .network_architecture.norm_act import _InstanceNormAct
    if not isinstance(norm, _InstanceNormAct) or norm.track_running_stats:
        return None
    B, C, N = x.shape
    M = N if y is None else y.shape[2]
    w = conv.weight
    groups = int(conv.groups)
    if conv.transposed or isinstance(conv.padding, str) or w.dtype != torch.float32 or any(k != 1 for k in w.shape[2:]) or \
            any(v != 1 for v in conv.stride) or any(v != 0 for v in conv.padding) or any(v != 1 for v in conv.dilation) or \
            getattr(conv, "_pad_spec", None) is not None or w.shape[1] * groups != 2 * C or w.shape[0] % groups:
        return None
    if conv.bias is not None and getattr(norm, "_pre_bias_src", (None,))[0] is not conv:
        return None               # a bias the norm does not absorb would have to be added to h
    K = nn_idx.shape[2]
    if _HIP.mr_grouped_cm_tiles(B, C, groups, w.shape[0] // groups, N, M, K) <= 0:
        return None
    h, partial = _MRGroupedCM.apply(_f32c(x), None if y is None else _f32c(y), nn_idx.contiguous(), w, K, 1, groups, True)
    # the convolution's folded bias rides along as the norm's `pre_bias` exactly as in the op-by-op block: instance statistics absorb it
    # (no effect on the values), and it keeps its — exactly zero — gradient instead of `None`
    return norm_act(h, norm.weight, norm.bias, None, None, True, 0.0, norm.eps, norm.negative_slope, instance=True,
                    pre_bias=conv.bias, stats_partial=partial)


def mr_grouped_conv(windows, nn_idx, conv, norm, batch, spatial, window, shift):
    """MRConv inside Swin windows up to its grouped 1x1 convolution in ONE launch (K2 + K7), for blocks outside the fused point-wise
    chain: ``conv(window_scatter(mr_aggregate(windows, nn_idx)))`` as a channels-last volume, or ``None`` when the shape / module is not
    taken (the caller runs the three ops).  ``conv``'s bias must be absent or folded into ``norm`` (norm_act._ConvBiasFolded)."""
    import os
    if os.environ.get("NEXTOU_MR_GROUPED", "1") == "0" or not windows.is_cuda or windows.dtype != torch.float32 or \
            torch.is_autocast_enabled("cuda") or nn_idx.dtype != torch.int32:
        return None
    n_windows, C, Nw = windows.shape
    w = conv.weight
    if not _plain_1x1(conv, 2 * C) or w.shape[0] != 2 * C:
        return None
    if conv.bias is not None and getattr(norm, "_pre_bias_src", (None,))[0] is not conv:
        return None
    K, groups = nn_idx.shape[2], int(conv.groups)
    # one workgroup per (window, group): with fewer than a workgroup per CU the three small launches win as replayed hipGraphs (cfg 2:
    # stage 3, 768 workgroups, 967 vs 1 005 us per block forward + backward; stage 4, 96 workgroups, 406 vs 381; stage 5, 12 workgroups,
    # 349 vs 296 — profiles/r04_k2_k7_fused.md)
    if n_windows * groups < int(os.environ.get("NEXTOU_MR_GROUPED_MIN_WORKGROUPS", "256")):
        return None
    if not _HIP.mr_grouped_rows_supported(n_windows, C, groups, Nw, K):
        return None
    return _MRGroupedConv.apply(_f32c(windows), w, nn_idx.contiguous(), K, groups, int(batch), tuple(int(v) for v in spatial),
                                tuple(int(v) for v in window), tuple(int(v) for v in shift))


def mr_grouped_chain(windows, nn_idx, residual, conv1, norm1, conv2, norm2, spatial, window, shift):
    """SwinGrapher's ``fc2(BasicConv(mr_aggregate(windows)))  + residual`` with the aggregation, the window reverse and the grouped
    1x1 convolution in ONE kernel (reference NexToU_Encoder_Decoder.py:401-418, torch_nn.py:66-92), followed by the
    rest of the fused point-wise chain (:class:`_PointwiseChain`) — or ``None`` when a piece does not qualify (the caller then runs
    mr_aggregate -> window_scatter -> pointwise_chain).  ``windows``: (B * nWin, C, Nw) float32 from :func:`window_gather`;
    ``nn_idx``: (B * nWin, Nw, K) int32; ``residual``: the block's channels-last input."""
    mode = _pw_fuse_mode()
    if mode == "0" or not windows.is_cuda or windows.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
        return None
    n_windows, C, Nw = windows.shape
    K = nn_idx.shape[2]
    groups = int(conv1.groups)
    batch = residual.shape[0]
    shape = (batch, 2 * C) + tuple(int(v) for v in spatial)
    if nn_idx.dtype != torch.int32 or conv1.weight.shape[0] != 2 * C or not _chain_modules_eligible(shape, residual, conv1, norm1, conv2, norm2):
        return None
    if not _HIP.mr_grouped_rows_supported(n_windows, C, groups, Nw, K):
        return None
    n1, n2 = _norm_state(norm1), _norm_state(norm2)
    w1 = conv1.weight
    w1m = w1.detach().reshape(w1.shape[0], w1.shape[1]).contiguous()
    window, shift = tuple(int(v) for v in window), tuple(int(v) for v in shift)
    needs_grad = torch.is_grad_enabled() and (windows.requires_grad or w1.requires_grad)
    res = as_channels_last_rows(residual)
    if needs_grad:
        box = {}
        a, h = _MRAggregateRows.apply(_f32c(windows), nn_idx.contiguous(), K, 1, w1m, groups, batch, shape[2:], window, shift,
                                      n1.batch_stats, box)
        part = box["part"]
    else:               # eval / no_grad: the aggregate itself is never written
        _, _, h, part = _HIP.mr_grouped_rows(_f32c(windows), nn_idx.contiguous(), K, 1, w1m, groups, batch, shape[2:], window, shift,
                                             want_a=False, want_arg=False, want_stats=n1.batch_stats)
        a = h           # (x is only the chain's shape carrier here)
    return _PointwiseChain.apply(a, res, w1, norm1.weight, norm1.bias, ZERO_GRADS.take(conv1.bias, n1.batch_stats), conv2.weight,
                                 norm2.weight, norm2.bias, ZERO_GRADS.take(conv2.bias, n2.batch_stats), groups, n1, n2, mode != "fwd", h, part)


def _norm_state(norm) -> _NormState:
    """torch.nn.modules.batchnorm._BatchNorm.forward's bookkeeping for one fused norm module (see norm_act._BatchNormAct)."""
    batch_stats, factor, keep_running = norm._step()
    return _NormState(batch_stats, factor, norm.eps, norm.negative_slope, norm.running_mean if keep_running else None,
                      norm.running_var if keep_running else None)


def _plain_1x1(conv, cin) -> bool:
    w = conv.weight
    if conv.transposed or isinstance(conv.padding, str) or w.dtype != torch.float32:
        return False
    if any(k != 1 for k in w.shape[2:]) or any(v != 1 for v in conv.stride) or any(v != 0 for v in conv.padding) or \
            any(v != 1 for v in conv.dilation):
        return False
    g = conv.groups
    return w.shape[0] % g == 0 and (w.shape[0] // g) % 4 == 0 and w.shape[1] % 4 == 0 and cin == g * w.shape[1] and \
        getattr(conv, "_pad_spec", None) is None


def pointwise_chain_eligible(x, residual, conv1, norm1, conv2=None, norm2=None) -> bool:
    """Dense channels-last fp32 device volume outside autocast; 1x1 / stride-1 / unpadded convolutions with per-group channel counts
    that are multiples of 4, the second one un-grouped, each with its bias folded into the norm behind it (norm_act._ConvBiasFolded)
    or none; fused BatchNorm modules (batch or running statistics) without internal channel padding."""
    if not x.is_cuda or x.dtype != torch.float32 or torch.is_autocast_enabled("cuda") or _dense_channels_last(x) is None:
        return False
    return _chain_modules_eligible(tuple(x.shape), residual, conv1, norm1, conv2, norm2)


def _chain_modules_eligible(shape, residual, conv1, norm1, conv2=None, norm2=None) -> bool:
    """:func:`pointwise_chain_eligible` without the input tensor: ``shape`` = (B, C_in, *spatial) of the channels-last input."""
    points = 1
    for v in shape[2:]:
        points *= int(v)
    if shape[0] * points < max(_pw_fuse_min_points(), 2):       # (one value per channel: the norm modules raise torch's ValueError)
        return False
    if residual is not None and (residual.dtype != torch.float32 or residual.shape[0] != shape[0] or tuple(residual.shape[2:]) != tuple(shape[2:])):
        return False
    cin = shape[1]
    for conv, norm in ((conv1, norm1), (conv2, norm2)):
        if conv is None:
            continue
        if not _plain_1x1(conv, cin) or not hasattr(norm, "_step") or getattr(norm, "_pad_multiple", 0):
            return False
        if conv.bias is not None and getattr(norm, "_pre_bias_src", (None,))[0] is not conv:
            return False            # a bias that is really added to the conv output: not this pipeline
        if norm.num_features != conv.weight.shape[0] or norm.weight is None or norm.bias is None:
            return False            # affine norms only: the fused backward's gradient-statistics epilogue reads weight and bias
        batch_stats = norm.training or (norm.running_mean is None and norm.running_var is None)
        if not batch_stats and norm.running_mean is None:
            return False
        cin = conv.weight.shape[0]
    if conv2 is not None and conv2.groups != 1:
        return False
    out_c = (conv2 if conv2 is not None else conv1).weight.shape[0]
    return residual is None or residual.shape[1] == out_c


class _ConvDepthUnrolledGrads(torch.autograd.Function):
    """[3,3,3] convolution with depth stride 1 and in-plane stride > 1 (the stage-0 -> stage-1 down-sampling convolution of the
    3-D plans): MIOpen's forward as it is, both gradients as 2-D problems over depth-unrolled tensors (``nextou_depth_unroll``).
    Weight gradient: taps of x as input channels (see :class:`_ConvDgradAsForward`).  Data gradient: y[d] reads x[d + kd - 1],
    so gx[d'] = sum_kd convT2d(gy[d' - kd + 1], w[:, :, kd]) = the 2-D backward-data of the taps of gy (as 3*Cout output
    channels) through the depth-flipped filter — MIOpen's 2-D kernel instead of CK's 3-D backward-data."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        y = torch.ops.aten.convolution(x, weight, None, stride, padding, (1, 1, 1), False, (0, 0, 0), 1)
        ctx.save_for_backward(x, weight)
        ctx.conf = (tuple(stride), tuple(padding))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding = ctx.conf
        co, ci = weight.shape[:2]
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
        gx = gw = None
        import os
        mode = os.environ.get("NEXTOU_STRIDED_UNROLL", "both")          # both | wgrad | dgrad
        if mode != "both":
            mask = [ctx.needs_input_grad[0] and mode != "dgrad", ctx.needs_input_grad[1] and mode != "wgrad", False]
            if mask[0] or mask[1]:
                gx, gw, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, stride, padding, (1, 1, 1), False, (0, 0, 0), 1, mask)
        if ctx.needs_input_grad[0] and mode != "wgrad":
            w3 = weight.flip(2).permute(2, 0, 1, 3, 4).reshape(3 * co, ci, 3, 3)
            gx2, _, _ = torch.ops.aten.convolution_backward(_HIP.depth_unroll(gy), flat_depth(x), w3, None, stride[1:], padding[1:], (1, 1),
                                                            False, (0, 0), 1, [True, False, False])
            gx = unflat_depth(gx2, x.shape[0], x.shape[2])
        if ctx.needs_input_grad[1] and mode != "dgrad":
            w2 = weight.permute(0, 2, 1, 3, 4).reshape(co, 3 * ci, 3, 3)
            _, gw2, _ = torch.ops.aten.convolution_backward(flat_depth(gy), _HIP.depth_unroll(x), w2, None, stride[1:], padding[1:], (1, 1),
                                                            False, (0, 0), 1, [False, True, False])
            gw = gw2.reshape(co, 3, ci, 3, 3).permute(0, 2, 1, 3, 4)
        return gx, gw, None, None


def depth_unrolled_grads_eligible(conv: torch.nn.Module, x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Un-grouped [3,3,3] convolutions with depth stride 1, an in-plane stride > 1 and 'same' zero padding whose gradients
    :class:`_ConvDepthUnrolledGrads` computes as 2-D problems (same size rule as the stride-1 weight gradient); opt-in with
    ``NEXTOU_STRIDED_UNROLL=both|wgrad|dgrad``."""
    import os
    if os.environ.get("NEXTOU_STRIDED_UNROLL", "off") not in ("both", "wgrad", "dgrad"):
        return False            # opt-in: measured 181.1 vs 180.2 ms / step with it on
    if conv.transposed or conv.groups != 1 or isinstance(conv.padding, str) or getattr(conv, "padding_mode", "zeros") != "zeros":
        return False
    if torch.is_autocast_enabled("cuda") or weight.dtype != torch.float32 or weight.shape[0] % 4 != 0:
        return False
    stride = tuple(conv.stride)
    if len(stride) != 3 or stride[0] != 1 or max(stride[1:]) == 1 or any(d != 1 for d in conv.dilation):
        return False
    return wgrad_depth_unroll_eligible(x, weight, conv.padding)


def conv_depth_unrolled_grads(x, weight, stride, padding):
    return _ConvDepthUnrolledGrads.apply(x, weight, tuple(int(v) for v in stride), tuple(int(v) for v in padding))


def wgrad_depth_unroll_eligible(x: torch.Tensor, weight: torch.Tensor, padding) -> bool:
    """[3,3,3] kernels with 'same' padding on a dense channels-last fp32 device volume of at least 16 384 points per sample batch
    (below that the launch of the unroll costs what the 2-D kernel gains).  ``NEXTOU_WGRAD_DEPTH_UNROLL=0`` keeps the 3-D path."""
    import os
    if os.environ.get("NEXTOU_WGRAD_DEPTH_UNROLL", "1") == "0":
        return False
    if x.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or tuple(padding) != (1, 1, 1):
        return False
    if not x.is_cuda or x.dtype != torch.float32 or x.shape[1] % 4 != 0 or _dense_channels_last(x) is None:
        return False
    return x.shape[0] * x.shape[2] <= 65535 and x.numel() // x.shape[1] >= 16384


def dgrad_as_forward_eligible(conv: torch.nn.Module, x: torch.Tensor) -> bool:
    """Device fp32 tensors outside autocast, un-grouped stride-1 convolutions with odd kernels and 'same' zero padding."""
    import os
    if os.environ.get("NEXTOU_DGRAD_AS_FWD", "1") == "0":
        return False
    if not x.is_cuda or x.dtype != torch.float32 or conv.weight.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
        return False
    if conv.transposed or conv.groups != 1 or isinstance(conv.padding, str) or getattr(conv, "padding_mode", "zeros") != "zeros":
        return False
    return all(s == 1 for s in conv.stride) and all(d == 1 for d in conv.dilation) and \
        all(k % 2 == 1 and p == k // 2 for k, p in zip(conv.kernel_size, conv.padding))


def rows_gemm_eligible(conv: torch.nn.Module, x: torch.Tensor, weight: torch.Tensor) -> bool:
    """An un-grouped 1x1 / stride-1 / unpadded convolution of a SMALL dense channels-last device volume (the graph stages 4 / 5 of
    cfg 2: 2 688 and 336 points): a plain GEMM over the (points, channels) view of the same memory.  MIOpen's convolution kernels
    need 33-72 us per call at these sizes whatever the problem (and its find step picks between them run by run: the stage-4 / 5
    FFN measured between 0.55 and 1.07 ms forward + backward from one process to the next); the BLAS GEMM needs 20-47 us —
    324 -> 1296 at 2 688 points: 33 / 72 / 67 us (forward / data gradient / weight gradient) against 47 / 29 / 29 us, at 336 points
    44 / 70 / 66 against 20 / 20 / 20 us (profiles/r02_pw_gemm.md, columns conv2d and mm).  From 21 504 points (stage 3) MIOpen's 2-D
    kernels are the faster ones.  ``NEXTOU_PW_MM_MAX_POINTS`` (default 8192, 0 = off) is the switch."""
    import os
    limit = int(os.environ.get("NEXTOU_PW_MM_MAX_POINTS", "8192"))
    if limit <= 0 or not x.is_cuda or conv.transposed or conv.groups != 1 or isinstance(conv.padding, str):
        return False
    if any(k != 1 for k in weight.shape[2:]) or any(v != 1 for v in conv.stride) or any(v != 0 for v in conv.padding) or \
            any(v != 1 for v in conv.dilation):
        return False
    if x.shape[1] != weight.shape[1] or _dense_channels_last(x) is None:
        return False
    return x.numel() // x.shape[1] <= limit


def rows_gemm(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``conv1x1(x, weight)`` (no bias) of a dense channels-last volume as ``rows @ weight.T``; input and output rows are views of the
    channels-last memory, autograd's two gradient GEMMs come with ``linear``."""
    b, c = x.shape[:2]
    spatial = tuple(x.shape[2:])
    nd = x.dim()
    rows = x.permute(0, *range(2, nd), 1).reshape(-1, c)
    y = torch.nn.functional.linear(rows, weight.reshape(weight.shape[0], c))
    return y.view(b, *spatial, weight.shape[0]).permute(0, nd - 1, *range(1, nd - 1))


def grouped_cm_gemm_eligible(conv: torch.nn.Module, x: torch.Tensor, weight: torch.Tensor) -> bool:
    """The Pool MRConv's grouped 1x1 convolution (reference torch_nn.py:66-92 inside NexToU_Encoder_Decoder.py:401-418) on the graph
    kernels' channel-major (B, 2C, N, 1[, 1]) tensor: per sample and group ``Y = W_g X_g`` with the N points contiguous — a strided-batched
    BLAS GEMM over the (B, groups, 2C / groups, N) view of the same memory.  MIOpen runs this NCDHW grouped convolution through NHWC
    kernels wrapped in layout transposes; measured on MI355X as GPU kernel time, forward / forward + backward
    (tools/pool_basicconv_probe.py, profiles/r04_pool_basicconv_probe.md): cfg-2 Pool s2 (2 x 264 x 10 752, 6 groups) 24.5 / 142.9 us
    against 21.7 / 75.9 us for the batched GEMM, Pool s3 (528 channels) 38.8 / 251.1 against 47.3 / 138.3; level at 1 344 points
    (53.8 vs 52.5) and behind at 168 (49.4 vs 70.8) — hence from 4 096 points per sample (``NEXTOU_GROUPED_GEMM_MIN_POINTS``, 0 = off)."""
    import os
    limit = int(os.environ.get("NEXTOU_GROUPED_GEMM_MIN_POINTS", "4096"))
    if limit <= 0 or conv.groups < 2 or conv.transposed or isinstance(conv.padding, str):
        return False
    if not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
        return False
    if any(k != 1 for k in weight.shape[2:]) or any(v != 1 for v in conv.stride) or any(v != 0 for v in conv.padding) or \
            any(v != 1 for v in conv.dilation):
        return False
    if not x.is_contiguous() or x.shape[1] != conv.groups * weight.shape[1] or weight.shape[0] % conv.groups:
        return False
    return x.numel() // (x.shape[0] * x.shape[1]) >= limit


def grouped_cm_gemm(x: torch.Tensor, weight: torch.Tensor, groups: int) -> torch.Tensor:
    """``conv1x1(x, weight, groups)`` (no bias) of a channel-major tensor as ``W_g @ X_g`` per (sample, group); autograd's two gradient
    GEMMs come with ``matmul`` (the weight gradient sums over the samples)."""
    b, c = x.shape[:2]
    spatial = tuple(x.shape[2:])
    co, ci = weight.shape[0] // groups, weight.shape[1]
    y = torch.matmul(weight.reshape(groups, co, ci), x.reshape(b, groups, ci, -1))
    return y.reshape(b, groups * co, *spatial)


def flat_depth_eligible(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, output_padding=None) -> bool:
    """A 3-D convolution whose kernel has no extent along the depth axis (NexToU's stage 0: [1,3,3]; the (1,2,2)
    up-convolution; every 1x1x1 head) on a dense channels-last volume IS a 2-D convolution of the (B*D, C, H, W) view of
    the same memory.  MIOpen runs the 2-D problem faster, above all in the weight gradient (40 -> 40 at 64x224x192: 3.10
    -> 2.35 ms; 80 -> 40: 5.9 -> 4.4 ms; 4 -> 40: 1.54 -> 0.49 ms — tools/conv2d_probe.py, profiles/r02_conv2d_probe.txt)."""
    import os
    if os.environ.get("NEXTOU_FLAT_DEPTH_CONV", "1") == "0":
        return False
    if x.dim() != 5 or weight.dim() != 5 or weight.shape[2] != 1 or not x.is_cuda:
        return False
    if stride[0] != 1 or padding[0] != 0 or dilation[0] != 1 or (output_padding is not None and output_padding[0] != 0):
        return False
    return x.shape[1] > 1 and x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous()


def flat_depth(x: torch.Tensor) -> torch.Tensor:
    """(B, C, D, H, W) channels_last_3d -> the (B*D, C, H, W) channels_last view of the same memory."""
    b, c, d, h, w = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(b * d, c, h, w)


def unflat_depth(y: torch.Tensor, b: int, d: int) -> torch.Tensor:
    """(B*D, C, H, W) channels_last -> the (B, C, D, H, W) channels_last_3d view of the same memory."""
    _, c, h, w = y.shape
    return y.reshape(b, d, c, h, w).permute(0, 2, 1, 3, 4)


def conv_dgrad_as_forward(x, weight, padding):
    padding = tuple(int(p) for p in padding)
    if flat_depth_eligible(x, weight, (1, 1, 1), padding, (1, 1, 1)):
        y = _ConvDgradAsForward.apply(flat_depth(x), weight.squeeze(2), padding[1:])
        return unflat_depth(y, x.shape[0], x.shape[2])
    return _ConvDgradAsForward.apply(x, weight, padding)


class _CatBias(torch.autograd.Function):
    """``torch.cat((y + bias, skip), 1)`` of two dense channels-last fp32 volumes as one pass (nextou_cat_bias_rows)."""

    @staticmethod
    def forward(ctx, y, bias, skip):
        L_ = _lib.lib()
        c1, c2 = y.shape[1], skip.shape[1]
        P = y.numel() // c1
        out = _empty_channels_last((y.shape[0], c1 + c2) + tuple(y.shape[2:]), y.device)
        b = None if bias is None else bias.contiguous()
        with torch.cuda.device(y.device):
            rc = L_.nextou_cat_bias_rows(y.data_ptr(), _ptr(b), skip.data_ptr(), out.data_ptr(), P, c1, c2, _stream_ptr(y.device))
        _lib.check(rc, "cat_bias_rows")
        ctx.c1, ctx.has_bias = c1, bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        c1 = ctx.c1
        mf = {4: torch.channels_last, 5: torch.channels_last_3d}[g.dim()]
        want_gb = ctx.has_bias and ctx.needs_input_grad[1]
        both = _HIP.narrow_copy_sum(g, 0, c1) if (want_gb and g.is_cuda) else None       # copy + bias sums in one pass (ABI v13)
        if both is not None:
            gy, gb = both
        else:
            gy = g.narrow(1, 0, c1).contiguous(memory_format=mf)         # the copy the convolution's backward would make anyway
            gb = _HIP.channel_sum(gy, channels_last=True) if want_gb else None
        return gy, gb, g.narrow(1, c1, g.shape[1] - c1)


class _UpConvCat(torch.autograd.Function):
    """``torch.cat((conv_transpose(x, weight) + bias, skip), 1)`` for a transposed convolution whose kernel equals its stride (padding 0,
    dilation 1, one group — every transpconv of the decoder, reference NexToU_Encoder_Decoder.py:272-276) on dense channels-last fp32
    volumes.  Such a convolution has no overlapping taps: each input point produces its T = prod(stride) output points independently, so it
    is the GEMM x (P_in, Cin) . (Cin, T*Cout) followed by a pixel shuffle.  Forward: K7 (nextou_pw_rows) + ONE pass that shuffles, adds the
    bias and concatenates (nextou_upconv_cat_rows — the traffic of the concatenation that ran anyway).  Backward: one pass un-shuffles the
    first Cout channels of the gradient and forms the bias sums (nextou_upconv_cat_rows_bwd), then K7's data-gradient GEMM and weight
    gradient.  The library ran these as a backward-data kernel used forwards at 20 % of the fp32 MFMA peak (profiles/r05_conv_layer_table.md:
    1 029 + 1 033 us at the full-resolution stage of cfg 2, behind a 685-us concatenation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, skip, stride):
        cin, cout = weight.shape[0], weight.shape[1]
        T = weight.numel() // (cin * cout)
        # (Cin, Cout, *k) -> rows (t, co) x columns ci: the GEMM's (N, K) operand
        n = weight.dim() - 2
        w2 = weight.permute(*range(2, 2 + n), 1, 0).reshape(T * cout, cin).contiguous()
        b_ = None if bias is None else bias.contiguous()
        import os
        if all(int(v) in (1, 2, 4) for v in stride) and os.environ.get("NEXTOU_UPCONV_DIRECT", "0") == "1":
            # the shuffle in the GEMM's store, no (P_in, T*Cout) intermediate — OFF by default: measured level with the two-pass route
            # (165.79 / 165.71 vs 165.76 / 165.89 ms per cfg-2 step, one box, alternating; profiles/r06_step_ab.md).  Both halves of the
            # concatenation are then written as 160-byte pieces of 320-byte rows, which the memory system takes at ~2.7 TB/s where the
            # one pass that writes whole rows runs at 5.2
            out = _HIP.upconv_cat_direct(x, w2, b_, skip, stride, cout)
        else:
            out = _HIP.upconv_cat_rows(_HIP.pw_rows(x, w2, None, 1), b_, skip, stride)
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.has_bias, ctx.cout = tuple(int(v) for v in stride), bias is not None, cout
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        cin, cout = weight.shape[0], weight.shape[1]
        n = weight.dim() - 2
        T = weight.numel() // (cin * cout)
        mf = {4: torch.channels_last, 5: torch.channels_last_3d}[g.dim()]
        if _dense_channels_last(g) is None:
            g = g.contiguous(memory_format=mf)
        both = _HIP.upconv_cat_rows_bwd(g, cout, tuple(x.shape[2:]), ctx.stride)
        if both is None:        # wide rows (C1 > 128): ATen's strided copy, then the shuffle as a view + copy
            gy = g.narrow(1, 0, cout)
            B = g.shape[0]
            sp_in = tuple(x.shape[2:])
            shp = [B, cout]
            for d_, s_ in zip(sp_in, ctx.stride):
                shp += [d_, s_]
            gy = gy.reshape(shp)                                              # (B, co, d, sd, h, sh, w, sw)
            taps = [3 + 2 * i for i in range(n)]
            dims = [2 + 2 * i for i in range(n)]
            gy2 = gy.permute(0, *taps, 1, *dims).reshape((B, T * cout) + sp_in).contiguous(memory_format=mf)
            gb = gy2.reshape((B, T, cout) + sp_in).sum(dim=[0, 1] + list(range(3, 3 + n))) if ctx.has_bias else None
        else:
            gy2, gb = both
        gx = gw = None
        if ctx.needs_input_grad[0]:
            # gx[p, ci] = sum_{t, co} gy2[p, t*Cout + co] * weight[ci, co, t]: the (N = Cin, K = T*Cout) operand is the filter as stored channels-last
            w3 = weight.permute(0, *range(2, 2 + n), 1).reshape(cin, T * cout).contiguous()
            gx = _HIP.pw_rows(gy2, w3, None, 1)
        if ctx.needs_input_grad[1]:
            dw2 = _HIP.pw_wgrad(gy2, x, 1)                                   # (T*Cout, Cin)
            gw = dw2.reshape(tuple(weight.shape[2:]) + (cout, cin)).permute(n + 1, n, *range(n))
            if weight.dim() in (4, 5) and weight.is_contiguous(memory_format=mf) and not weight.is_contiguous():
                gw = gw.contiguous(memory_format=mf)                          # the layout the parameter is stored in (layout.filters_to_channels_last)
            else:
                gw = gw.contiguous()
        if not (ctx.has_bias and ctx.needs_input_grad[2]):
            gb = None
        return gx, gw, gb, g.narrow(1, cout, g.shape[1] - cout), None


def upconv_cat_eligible(x: torch.Tensor, weight: torch.Tensor, bias, skip: torch.Tensor, stride, padding, dilation, output_padding,
                        groups, kernel_size) -> bool:
    import os
    if os.environ.get("NEXTOU_UPCONV_GEMM", "1") == "0":
        return False
    if not (x.is_cuda and skip.is_cuda) or x.dtype != torch.float32 or skip.dtype != torch.float32 or weight.dtype != torch.float32:
        return False
    if torch.is_autocast_enabled("cuda") or groups != 1 or x.dim() not in (4, 5):
        return False
    if tuple(kernel_size) != tuple(stride) or any(int(p) for p in padding) or any(int(d) != 1 for d in dilation) or \
            any(int(o) for o in output_padding) or any(int(s_) < 1 or int(s_) > 4 for s_ in stride):
        return False
    cin, cout, c2 = weight.shape[0], weight.shape[1], skip.shape[1]
    if x.shape[1] != cin or cin % 4 or cout % 4 or c2 % 4 or cout + c2 > 1024 or x.shape[0] != skip.shape[0]:
        return False
    if tuple(skip.shape[2:]) != tuple(int(d_) * int(s_) for d_, s_ in zip(x.shape[2:], stride)):
        return False
    if _dense_channels_last(x) is None or _dense_channels_last(skip) is None:
        return False
    return bias is None or (bias.dtype == torch.float32 and bias.shape[0] == cout)


def upconv_cat(x, weight, bias, skip, stride):
    return _UpConvCat.apply(x, weight, bias, skip, tuple(stride))


def cat_bias_eligible(y: torch.Tensor, bias, skip: torch.Tensor) -> bool:
    import os
    if os.environ.get("NEXTOU_CAT_BIAS", "1") == "0":
        return False
    if not (y.is_cuda and skip.is_cuda) or y.dtype != torch.float32 or skip.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
        return False
    if y.shape[0] != skip.shape[0] or y.shape[2:] != skip.shape[2:] or y.shape[1] % 4 or skip.shape[1] % 4 or y.shape[1] + skip.shape[1] > 1024:
        return False
    if _dense_channels_last(y) is None or _dense_channels_last(skip) is None:
        return False
    return bias is None or (bias.dtype == torch.float32 and bias.shape[0] == y.shape[1])


def cat_bias(y: torch.Tensor, bias, skip: torch.Tensor) -> torch.Tensor:
    return _CatBias.apply(y, bias, skip)


def conv_own_bias_grad(x, weight, bias, stride, padding, dilation, transposed, output_padding, groups):
    """N-d (transposed) convolution with bias on the GPU; see :class:`_ConvOwnBiasGrad`."""
    if flat_depth_eligible(x, weight, stride, padding, dilation, output_padding):
        y = _ConvOwnBiasGrad.apply(flat_depth(x), weight.squeeze(2), bias, tuple(stride)[1:], tuple(padding)[1:],
                                   tuple(dilation)[1:], bool(transposed), tuple(output_padding)[1:], int(groups))
        return unflat_depth(y, x.shape[0], x.shape[2])
    return _ConvOwnBiasGrad.apply(x, weight, bias, tuple(stride), tuple(padding), tuple(dilation), bool(transposed),
                                  tuple(output_padding), int(groups))


@torch.no_grad()
def checked_label_map(target: torch.Tensor, n_classes: int, flag: torch.Tensor) -> torch.Tensor:
    """uint8 copy of a label map (float32 as nnU-Net hands it, int64 or uint8) in ONE pass, with the reference's range check
    (CrossEntropyLoss raises for targets outside [0, L), bti_loss.py:141) recorded in ``flag`` ((1,) int32 on the device, OR-ed) instead
    of a host read.  Other dtypes are converted by ATen first."""
    t = target.detach()
    if t.dtype not in (torch.float32, torch.int64, torch.uint8):
        t = t.float()
    t = t.contiguous()
    if not t.is_cuda:                   # CPU checker path (tests): plain torch
        flag |= int(bool(((t < 0) | (t >= n_classes)).any()))
        return t.to(torch.uint8)
    return _HIP.labels_u8(t, n_classes, flag)


@torch.no_grad()
def argmax_labels(logits: torch.Tensor) -> torch.Tensor:
    """uint8 arg-max over dim 1 (first index on ties) of (B,L,*spatial) float32 logits."""
    logits = _logits_in_place(logits.detach())
    return _backend_for(logits).argmax_labels(logits)


@torch.no_grad()
def bti_critical_map(labels: torch.Tensor, lut_a: torch.Tensor, lut_c: torch.Tensor,
                     connectivity: int, min_thick: int = 1) -> torch.Tensor:
    """uint8 critical-voxel map of a uint8 label volume (B,[D,]H,W); LUTs are uint32-as-int32."""
    labels = labels.contiguous()
    if labels.dtype != torch.uint8:
        raise TypeError("bti_critical_map: labels must be uint8, got %s" % labels.dtype)
    return _backend_for(labels).bti_critical(labels, lut_a.contiguous(), lut_c.contiguous(),
                                             int(connectivity), int(min_thick))


# ----------------------------------------------------------------------------------------------
# K3 / K4: window shift + partition / reverse and query max-pool / unpool, fused with the layout change between the dense
# stages' channels-last volumes and the graph kernels' channel-major rows
# ----------------------------------------------------------------------------------------------
def as_channels_last_rows(x: torch.Tensor) -> torch.Tensor:
    """float32 tensor whose MEMORY is (B, *spatial, C)-contiguous (a no-op for a dense channels-last tensor)."""
    mf = {4: torch.channels_last, 5: torch.channels_last_3d}[x.dim()]
    if x.dtype != torch.float32:
        x = x.float()
    if x.shape[1] == 1:          # one channel: NCDHW and NDHWC are the same bytes
        return x.contiguous()
    return x.contiguous(memory_format=mf)


class _WindowGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, window, shift):
        ctx.conf = (tuple(x.shape[2:]), window, shift)
        return _backend_for(x).window_gather(x, window, shift)

    @staticmethod
    def backward(ctx, g):
        spatial, window, shift = ctx.conf
        return _backend_for(g).window_scatter(g.contiguous(), None, spatial, window, shift), None, None


class _WindowScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, residual, spatial, window, shift):
        ctx.conf = (window, shift, residual is not None)
        return _backend_for(src).window_scatter(src, residual, spatial, window, shift)

    @staticmethod
    def backward(ctx, g):
        window, shift, has_res = ctx.conf
        g = as_channels_last_rows(g)
        return _backend_for(g).window_gather(g, window, shift), (g if has_res else None), None, None, None


def window_gather(x: torch.Tensor, window, shift) -> torch.Tensor:
    """``window_partition(torch.roll(x, -shift))`` (reference NexToU_Encoder_Decoder.py:781-790, :634-660) of a
    channels-last volume x (B,C,*spatial) -> (B * n_windows, C, prod(window)) channel-major rows."""
    return _WindowGather.apply(as_channels_last_rows(x), tuple(int(v) for v in window), tuple(int(v) for v in shift))


def window_scatter(windows: torch.Tensor, spatial, window, shift, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``torch.roll(window_reverse(windows), +shift) [+ residual]`` (reference :807-817, :662-693): (B * n_windows, C, Nw)
    -> channels-last (B,C,*spatial)."""
    res = None if residual is None else as_channels_last_rows(residual)
    return _WindowScatter.apply(_f32c(windows), res, tuple(int(v) for v in spatial), tuple(int(v) for v in window),
                                tuple(int(v) for v in shift))


class _PoolRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool, forced_cell):
        be = _backend_for(x)
        if forced_cell is None:
            values, cell = be.pool_rows(x, pool)
        else:                                   # teacher-forced arg-max cells (test hook): a plain gather
            values, cell = be.cell_gather(x, forced_cell, pool), forced_cell
        ctx.save_for_backward(cell)
        ctx.conf = (tuple(x.shape[2:]), pool)
        ctx.mark_non_differentiable(cell)
        return values, cell

    @staticmethod
    def backward(ctx, g, _):
        (cell,) = ctx.saved_tensors
        spatial, pool = ctx.conf
        return _backend_for(g).cell_scatter(_f32c(g), cell, spatial, pool), None, None


def pool_rows(x: torch.Tensor, pool, forced_cell: Optional[torch.Tensor] = None):
    """``MaxPool(pool, stride=pool, return_indices=True)`` (reference :524-530) of a channels-last volume ->
    (values (B,C,N) channel-major, cell (B,N,C) uint8 = winning position inside each pooling cell)."""
    return _PoolRows.apply(as_channels_last_rows(x), tuple(int(v) for v in pool), forced_cell)


class _CellScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, cell, spatial, pool):
        ctx.save_for_backward(cell)
        ctx.pool = pool
        return _backend_for(src).cell_scatter(src, cell, spatial, pool)

    @staticmethod
    def backward(ctx, g):
        (cell,) = ctx.saved_tensors
        g = as_channels_last_rows(g)
        return _backend_for(g).cell_gather(g, cell, ctx.pool), None, None, None


def cell_scatter(src: torch.Tensor, cell: torch.Tensor, spatial, pool) -> torch.Tensor:
    """``MaxUnpool(out, cat(indices, indices))`` (reference :536-549): src (B,C2,N), C2 = C or 2C, -> channels-last
    (B,C2,*spatial) with src at the recorded cell position of channel c2 mod C and zeros elsewhere."""
    return _CellScatter.apply(_f32c(src), cell, tuple(int(v) for v in spatial), tuple(int(v) for v in pool))


def cells_to_flat_indices(cell: torch.Tensor, spatial, pool) -> torch.Tensor:
    """cell (B,N,C) uint8 -> the reference's MaxPool ``indices`` (B,C,*pooled) int64 (flat position in the un-pooled
    spatial volume).  Only used to record / replay the index tape (test hook)."""
    spatial, pool = [int(v) for v in spatial], [int(v) for v in pool]
    pooled = [s // p for s, p in zip(spatial, pool)]
    B, N, C = cell.shape
    k = cell.long().permute(0, 2, 1).reshape(B, C, *pooled)
    flat, stride, rem_k = torch.zeros_like(k), 1, k
    coords = torch.meshgrid(*[torch.arange(n, device=cell.device) for n in pooled], indexing="ij")
    for axis in reversed(range(len(spatial))):
        kk = rem_k % pool[axis]
        rem_k = rem_k // pool[axis]
        flat = flat + (coords[axis] * pool[axis] + kk) * stride
        stride *= spatial[axis]
    return flat


def flat_indices_to_cells(indices: torch.Tensor, spatial, pool) -> torch.Tensor:
    """inverse of :func:`cells_to_flat_indices` -> cell (B,N,C) uint8."""
    spatial, pool = [int(v) for v in spatial], [int(v) for v in pool]
    B, C = indices.shape[:2]
    rem, k, mult = indices.long(), torch.zeros_like(indices, dtype=torch.long), 1
    for axis in reversed(range(len(spatial))):
        pos = rem % spatial[axis]
        rem = rem // spatial[axis]
        k = k + (pos % pool[axis]) * mult
        mult *= pool[axis]
    return k.reshape(B, C, -1).permute(0, 2, 1).contiguous().to(torch.uint8)
