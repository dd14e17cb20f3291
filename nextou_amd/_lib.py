"""ctypes binding of libnextou_hip.so (the C-ABI declared in include/nextou_hip.h).

The library is the product; there is no Python/CPU fallback.  ``lib()`` raises
``RuntimeError`` with the build command when the shared object is missing, and every wrapper
turns a non-zero return code into a ``RuntimeError`` carrying ``nextou_last_error()``.

torch is imported first on purpose: PyTorch-ROCm bundles its own ``libamdhip64.so.7``; loading
it before ``dlopen`` makes the dynamic loader reuse that one runtime (same SONAME) so device
pointers and streams are shared between torch and this library.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# NEXTOU_HIP_LIB points experiments (tools/ablate_knn.sh) at a side build; the product never sets it.
LIB_PATH = os.environ.get("NEXTOU_HIP_LIB") or os.path.join(_PKG_DIR, "libnextou_hip.so")

KNN_AUTO, KNN_FUSED, KNN_NAIVE = 0, 1, 2
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
ABI_VERSION = 14
EINVAL, ENOSPACE, ENOTSUP = -1, -2, -3       # include/nextou_hip.h

# name -> (restype, argtypes); mirrors include/nextou_hip.h one to one
_SIGNATURES = {
    "nextou_abi_version": (c_int, []),
    "nextou_last_error": (c_char_p, []),
    "nextou_profile_enable": (c_int, [c_int]),
    "nextou_profile_report": (c_size_t, [c_char_p, c_size_t]),
    "nextou_profile_dropped": (c_int, []),
    "nextou_knn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "nextou_knn_graph": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "nextou_pairwise_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "nextou_pairwise_distance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "nextou_edge_index_i64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "nextou_mr_aggregate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "nextou_mr_aggregate_has_arg": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "nextou_mr_grouped_rows_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "nextou_mr_grouped_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p]),
    "nextou_mr_grouped_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int] +
                               [c_int] * 12 + [c_void_p]),
    "nextou_mr_aggregate_bwd_arg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_void_p]),
    "nextou_mr_aggregate_bwd_wants_idx": (c_int, [c_int, c_int, c_int, c_int]),
    "nextou_mr_aggregate_bwd_arg_idx": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_void_p]),
    "nextou_mr_aggregate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p]),
    "nextou_gather_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "nextou_gather_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "nextou_argmax_labels": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "nextou_labels_u8": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "nextou_bti_ce_partials": (c_int, []),
    "nextou_bti_ce_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "nextou_bti_ce_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64,
                                  c_void_p]),
    "nextou_dice_stats_partials": (c_int, []),
    "nextou_dice_stats_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "nextou_dice_stats_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64,
                                      c_void_p]),
    "nextou_ce_mean_partials": (c_int, []),
    "nextou_ce_mean_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "nextou_ce_mean_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "nextou_bti_critical_map": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_void_p]),
    "nextou_norm_act_workspace_bytes": (c_size_t, [c_int, c_int, c_int64, c_int]),
    "nextou_norm_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_int, c_int, c_int64, c_int, c_int, c_int, c_int,
                                    c_float, c_float, c_float, c_void_p]),
    "nextou_norm_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_int, c_int, c_int64, c_int, c_int, c_int, c_int,
                                    c_float, c_void_p]),
    "nextou_norm_act_bwd_two": (c_int, [c_void_p, c_void_p, c_void_p, c_int64] + [c_void_p] * 8 + [c_size_t, c_int, c_int, c_int64, c_int, c_float,
                                        c_void_p]),
    "nextou_channel_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int64, c_int, c_int,
                                   c_void_p]),
    "nextou_window_gather": (c_int, [c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "nextou_window_scatter": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "nextou_pool_rows": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "nextou_cell_gather": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "nextou_cell_scatter": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "nextou_cat_bias_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "nextou_narrow_copy_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int64, c_int, c_void_p]),
    "nextou_upconv_cat_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "nextou_upconv_cat_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t] + [c_int] * 9 + [c_void_p]),
    "nextou_device_write_i64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nextou_grad_norm_clip_coef": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_void_p, c_float, c_void_p, c_void_p]),
    "nextou_clip_sgd_update": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_void_p, c_float, c_void_p, c_float, c_float,
                                       c_int, c_void_p]),
    "nextou_filter_flip_t": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int64,
                                     c_void_p]),
    "nextou_depth_unroll": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "nextou_pw_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "nextou_pw_rows_up": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64] + [c_int] * 8 + [c_void_p]),
    "nextou_pw_wgrad_workspace": (c_int, [c_int64, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "nextou_pw_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_int64, c_int64,
                                c_int, c_void_p]),
    "nextou_pw_rows_tiles": (c_int, [c_int64, c_int, c_int, c_int, c_int64, c_int64, c_int, c_int]),
    "nextou_pw_rows_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int64,
                                     c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_float, c_void_p]),
    "nextou_pw_wgrad_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_int64, c_int64,
                                      c_int, c_void_p, c_void_p, c_float, c_void_p]),
    "nextou_norm_finalize": (c_int, [c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p]),
    "nextou_norm_apply_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                                       c_void_p]),
    "nextou_norm_bwd_finalize": (c_int, [c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "nextou_norm_bwd_apply_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                           c_int, c_float, c_void_p]),
    "nextou_mr_grouped_cm_tiles": (c_int, [c_int] * 7),
    "nextou_mr_grouped_cm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "nextou_norm_act_fwd_partials": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_int, c_int, c_int64, c_int, c_float, c_float, c_float, c_void_p]),
    "nextou_head_rows_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_void_p]),
    "nextou_head_rows_bwd_workspace": (c_int, [c_int64, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "nextou_head_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int,
                                     c_int64, c_int64, c_void_p]),
    "nextou_stem_workspace_bytes": (c_size_t, [c_int] * 5),
    "nextou_stem_fwd": (c_int, [c_void_p] * 13 + [c_size_t] + [c_int] * 7 + [c_float, c_float, c_float, c_void_p]),
    "nextou_stem_bwd": (c_int, [c_void_p] * 12 + [c_size_t] + [c_int] * 6 + [c_float, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; fail loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "nextou_amd: %s is missing — the HIP extension is the only implementation of the graph "
            "hot path (there is no CPU/PyTorch fallback). Build it with `python -m nextou_amd.build` "
            "(hipcc --offload-arch=gfx950)." % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    got = handle.nextou_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("nextou_amd: libnextou_hip.so ABI %d != expected %d; rebuild" % (got, ABI_VERSION))
    _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().nextou_last_error()
        raise RuntimeError("nextou_amd.%s failed (code %d): %s" % (what, code, (msg or b"").decode()))
