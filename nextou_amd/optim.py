"""``ClipSGD``: nnU-Net's optimizer step — ``clip_grad_norm_(parameters, 12)`` then ``torch.optim.SGD(momentum, nesterov,
weight_decay).step()`` — on this library's step-glue kernels (csrc/step_glue.hip, ABI v13).

The reference sets no optimizer of its own (nnUNetTrainer_NexToU.py:17-91 overrides only build_network_architecture and inherits nnUNetTrainer's ``configure_optimizers`` and
``train_step``); the plug-ins' step therefore ends with torch's multi-tensor clip and SGD — a few dozen launches of a few dozen workgroups
each for the 358 trainable tensors of a cfg-2 network (575 us replayed, profiles/r05_step_glue2.md).  ``ClipSGD`` IS a ``torch.optim.SGD`` (same constructor, ``param_groups``,
``state`` with ``momentum_buffer`` tensors, ``state_dict`` / ``load_state_dict``, LR schedulers) whose ``step()`` names its
tensors to the kernels through a table in device memory: one launch for the update, two more for the clip
(:meth:`clip_and_step`), whatever the tensor count.

Same update rule and operation order as ``torch.optim.SGD(foreach=True)``; results agree to the last bit or two (a fused
multiply-add here, a separately rounded product there — the same spread as between torch's own foreach / fused / single-tensor
variants).  Anything the kernels do not take (CPU or non-float32 parameters, sparse gradients, dampening, maximize) goes through
torch's implementation, unchanged — ``last_path`` / ``last_reason`` say which ran and why; a gradient stored in another element order
than its parameter is copied into the parameter's order first.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import _lib

__all__ = ["ClipSGD"]

_CHUNK = 16384          # elements per workgroup (64 KB of each tensor)


def _same_dense_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Same element order in memory: equal shapes and equal strides in every dimension LONGER THAN ONE.  (The stride of a size-1
    dimension is arbitrary — autograd's layout contract for ``.grad`` ignores it too — and the gradients of (N, K, 1, 1, 1) filters
    routinely arrive with other values there than the parameter has.)"""
    if a.shape != b.shape:
        return False
    sa, sb = a.stride(), b.stride()
    return sa == sb or all(x == y for x, y, n in zip(sa, sb, a.shape) if n > 1)


def _dense(p: torch.Tensor) -> bool:
    """True when ``p``'s elements occupy exactly ``numel`` consecutive slots in some dimension order (contiguous, channels-last ...):
    the tensor can be walked as a flat array."""
    dims = sorted((d for d in range(p.dim()) if p.shape[d] > 1), key=lambda d: p.stride(d))
    expect = 1
    for d in dims:
        if p.stride(d) != expect:
            return False
        expect *= p.shape[d]
    return True


def _stream_of(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0


def _capturing(device: torch.device) -> bool:
    return device.type == "cuda" and torch.cuda.is_current_stream_capturing()


class _on_device:
    def __init__(self, device):
        self._ctx = torch.cuda.device(device) if device.type == "cuda" else None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


class _Table:
    """Device table + chunk list + workspace of one set of (parameter, gradient, buffer) tensors.  Immutable once built: the values
    travel as kernel arguments (nextou_device_write_i64), so a captured hipGraph keeps rewriting — and its kernels keep reading — the
    table it was captured with; nothing on the host is read at replay and no pinned allocation happens inside a capture."""

    def __init__(self, rows: List[tuple], device: torch.device):
        import ctypes
        flat, chunks, total = [], [], 0
        for i, (p, g, m, n) in enumerate(rows):
            flat.extend((p, g, m, n))
            for c in range((n + _CHUNK - 1) // _CHUNK):
                chunks.append(i | (c << 32))              # int32 pair (row, chunk) as one little-endian int64
            total += n
        self.n_tensors, self.n_chunks, self.total = len(rows), len(chunks), total
        self.table = torch.empty((len(flat),), dtype=torch.int64, device=device)
        self.chunks = torch.empty((len(chunks),), dtype=torch.int64, device=device)
        L = _lib.lib()
        with _on_device(device):
            stream = _stream_of(device)
            for dst, vals in ((self.table, flat), (self.chunks, chunks)):
                host = (ctypes.c_int64 * len(vals))(*vals)
                _lib.check(L.nextou_device_write_i64(dst.data_ptr(), ctypes.cast(host, ctypes.c_void_p), len(vals), stream),
                           "device_write_i64")
        self.partial = torch.empty((self.n_chunks,), dtype=torch.float64, device=device)
        self.norm_coef = torch.empty((2,), dtype=torch.float32, device=device)


class ClipSGD(torch.optim.SGD):
    """``torch.optim.SGD`` with the step (and, through :meth:`clip_and_step`, the gradient clip) on own kernels."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, *, maximize=False,
                 foreach=None, differentiable=False, fused=None):
        # torch.optim.SGD's constructor, keyword for keyword: `foreach` / `fused` choose what the FALLBACK path runs (torch's
        # default on the GPU is foreach); the own kernels do not need either
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                         maximize=maximize, foreach=foreach, differentiable=differentiable, fused=fused)
        self._clip_request = None        # clip_and_step() -> step(): the max norm of this one step
        self._clip_norm = None
        self._tables = {}
        self._retired = []
        self._static = {}
        self._device_type = "cuda"       # (the CPU tests of the table plumbing run the same code on host tensors against a stand-in library)
        self.last_path = None            # "own" | "torch": which implementation the last step took (tests, bench line)
        self.last_reason = None          # why torch's implementation took it

    def __setstate__(self, state):
        # torch.optim.Optimizer pickles defaults / state / param_groups only: the tables (device pointers of another process) start empty
        super().__setstate__(state)
        self._tables, self._retired, self._static = {}, [], {}
        self.__dict__.setdefault("_device_type", "cuda")
        self.__dict__.setdefault("last_path", None)
        self.__dict__.setdefault("last_reason", None)
        self.__dict__.setdefault("_clip_request", None)
        self.__dict__.setdefault("_clip_norm", None)

    # ---------------------------------------------------------------------------------------------------------------
    def _plan(self, group):
        """(rows, device) of one param group for the kernels, or a string — why torch's implementation has to run it."""
        if os.environ.get("NEXTOU_CLIP_SGD", "1") == "0":
            return "NEXTOU_CLIP_SGD=0"
        if group["dampening"] != 0 or group["maximize"] or group.get("differentiable", False):
            return "dampening / maximize / differentiable"
        if group["nesterov"] and group["momentum"] <= 0:
            return "nesterov without momentum"
        lr = group["lr"]
        if isinstance(lr, torch.Tensor) and not (lr.device.type == self._device_type and lr.dtype == torch.float32 and lr.numel() == 1):
            return "learning-rate tensor not a float32 scalar on the device"
        rows, device = [], None
        f32, want_m, state, static = torch.float32, group["momentum"] != 0, self.state, self._static
        for p in group["params"]:
            g = p.grad
            if g is None:
                continue
            ptr = p.data_ptr()
            rec = static.get(id(p))
            if rec is None or rec[0] != ptr:
                # what does not change from step to step (checked again when the parameter's storage moves): (pointer, strides, count, device)
                if not (p.device.type == self._device_type and p.dtype == f32 and _dense(p)):
                    return "parameter %s %s on %s, strides %s: not a dense float32 %s tensor" % (
                        tuple(p.shape), p.dtype, p.device, p.stride(), self._device_type)
                rec = static[id(p)] = (ptr, p.stride(), p.numel(), p.device, p.shape)
            if device is None:
                device = rec[3]
            if g.dtype != f32 or g.is_sparse or g.device != device or g.shape != rec[4]:
                return "gradient %s %s on %s of parameter %s" % (tuple(g.shape), g.dtype, g.device, tuple(p.shape))
            if g.stride() != rec[1] and not _same_dense_layout(p, g):
                # a gradient in another element order than its parameter (autograd's layout contract makes this rare: a .grad assigned
                # by hand, bucket views of another layout): one copy into the parameter's order, then it is walked flat like the rest
                g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                p.grad = g
            m = 0
            if want_m:
                st = state[p]
                buf = st.get("momentum_buffer")
                if buf is None:
                    # zeros: momentum * 0 + d = d is torch's "buf = clone(d)" of the first step
                    buf = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if buf.dtype != f32 or buf.device != device or buf.shape != rec[4]:
                    return "momentum buffer %s %s on %s of parameter %s" % (tuple(buf.shape), buf.dtype, buf.device, tuple(p.shape))
                if buf.stride() != rec[1] and not _same_dense_layout(p, buf):
                    buf = st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(buf)
                m = buf.data_ptr()
            if rec[2]:
                rows.append((ptr, g.data_ptr(), m, rec[2]))
        if not rows:
            return "no gradients"
        return rows, device

    def _table(self, gi: int, rows, device) -> _Table:
        key = tuple(rows)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            t = cached[1]
        else:
            t = _Table(rows, device)
            self._tables[gi] = (key, t)
        if _capturing(device) and not any(r is t for r in self._retired):
            self._retired.append(t)               # the captured graph's kernels keep reading this table after a later step replaced it
        return t

    # ---------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _own_step(self, max_norm: Optional[float]):
        """The whole step on the kernels; returns the total-norm tensor (None without a clip), or NotImplemented when any group
        needs torch's implementation (nothing but zero-filled momentum buffers has been created then — torch's update of a
        zero buffer is its first-step ``buf = clone(d)``)."""
        groups_with_grads = [gi for gi, g in enumerate(self.param_groups) if any(p.grad is not None for p in g["params"])]
        if not groups_with_grads:
            self.last_reason = None
            return None if max_norm is None else torch.zeros(())
        plans = {gi: self._plan(self.param_groups[gi]) for gi in groups_with_grads}
        why = [pl for pl in plans.values() if isinstance(pl, str)]
        if not why and max_norm is not None and len(plans) != 1:
            why = ["the clip's norm spans %d parameter groups (the kernels take one table)" % len(plans)]
        self.last_reason = why[0] if why else None
        if why:
            return NotImplemented
        live = sorted(plans.items())
        L = _lib.lib()
        norm = None
        for gi, (rows, device) in live:
            group = self.param_groups[gi]
            t = self._table(gi, rows, device)
            lr = group["lr"]
            lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) else None
            with _on_device(device):
                stream = _stream_of(device)
                if max_norm is not None:
                    _lib.check(L.nextou_grad_norm_clip_coef(t.table.data_ptr(), t.n_tensors, t.chunks.data_ptr(), t.n_chunks, _CHUNK,
                                                            t.total, t.partial.data_ptr(), float(max_norm), t.norm_coef.data_ptr(),
                                                            stream), "grad_norm_clip_coef")
                    norm = t.norm_coef[0]
                _lib.check(L.nextou_clip_sgd_update(t.table.data_ptr(), t.n_tensors, t.chunks.data_ptr(), t.n_chunks, _CHUNK, t.total,
                                                    t.norm_coef.data_ptr() if max_norm is not None else None,
                                                    0.0 if lr_dev else float(lr), lr_dev, float(group["momentum"]),
                                                    float(group["weight_decay"]), int(bool(group["nesterov"])), stream),
                           "clip_sgd_update")
        return norm

    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        max_norm, self._clip_request = self._clip_request, None
        norm = self._own_step(max_norm)
        if norm is NotImplemented:
            self.last_path = "torch"
            if max_norm is not None:
                params = [p for g in self.param_groups for p in g["params"]]
                norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
            else:
                norm = None
            super().step()
        else:
            self.last_path = "own"
        self._clip_norm = norm
        return loss

    def clip_and_step(self, max_norm: float) -> torch.Tensor:
        """``torch.nn.utils.clip_grad_norm_(all parameters of this optimizer, max_norm)`` followed by :meth:`step` — through ``self.step``,
        i.e. with the optimizer's step pre / post hooks, the profiler's record and the LR schedulers' bookkeeping; returns the
        total norm (a 0-dim tensor, no host synchronisation).  The gradients are left scaled, as the clip leaves them.  The returned
        tensor is the caller's own copy outside a stream capture; inside one it is the table's buffer, rewritten by every replay."""
        self._clip_request = float(max_norm)
        try:
            self.step()
        finally:
            self._clip_request = None
        norm, self._clip_norm = self._clip_norm, None
        if isinstance(norm, torch.Tensor) and norm.is_cuda and not _capturing(norm.device):
            norm = norm.clone()
        return norm
