"""Data-parallel gradient exchange for the NexToU hot path: one process per GPU, RCCL over xGMI.

The reference has no distributed code of its own; nnU-Net wraps the network in
torch's DistributedDataParallel.  Patches are independent samples, so the path shards with exactly
one exchange step: the average of the 30.7 M trainable fp32 gradients (122.7 MB for cfg 2).  The
frozen ``relative_pos`` tables (145 MB) are ``requires_grad=False`` and never enter a bucket.

Design for xGMI (7 point-to-point links x ~153 GB/s per GPU): the whole gradient is
~1.4 ms of ring time, far below the backward pass, so the job is *overlap*, not bandwidth —
few, large, flat buckets (default 32 MiB -> 4 collectives per step instead of DDP's ~5 x 25 MiB +
first-bucket 1 MiB), filled in reverse parameter order (the order backward produces gradients) from
``post_accumulate_grad`` hooks, each all-reduced asynchronously on RCCL's own HIP stream the moment
its last gradient lands; ``finalize()`` joins the streams once, after backward.  BatchNorm
statistics stay per replica (no SyncBN), as in nnU-Net.

``backend='nccl'`` is RCCL on ROCm; the same code runs on ``gloo`` (CPU) for the world-size-2 tests.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    """One flat all-reduce unit: the gradients of ``params`` back to back, then one float per parameter that says
    whether any rank produced a gradient for it this step (summed by the same collective — no extra launch)."""
    __slots__ = ("flat", "params", "offsets", "views", "flags", "pending", "filled", "work", "n_grad", "ready")

    @staticmethod
    def _view_like(flat: torch.Tensor, offset: int, p: torch.Tensor) -> torch.Tensor:
        import os
        if os.environ.get("NEXTOU_DDP_STRIDED_VIEWS", "1") != "0" and _dense_strides(p) and not p.is_contiguous():
            return flat.as_strided(p.shape, p.stride(), flat.storage_offset() + offset)
        return flat[offset:offset + p.numel()].view_as(p)

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += p.numel()
        self.n_grad = total
        self.flat = torch.zeros(total + len(params), dtype=params[0].dtype, device=params[0].device)
        # views with the PARAMETER's strides where it is dense but not contiguous (a channels-last convolution weight keeps its permuted
        # element order inside its slice of the flat buffer): optimizers that walk parameter, gradient and momentum as flat arrays
        # (ClipSGD's kernels, torch.optim.SGD(fused=True)) pair the right elements without first copying ~100 filter gradients out of the
        # buckets every step.  NEXTOU_DDP_STRIDED_VIEWS=0: plain reshaped slices (rounds 2-5)
        self.views = [self._view_like(self.flat, o, p) for o, p in zip(self.offsets, params)]
        self.flags = self.flat[total:]
        self.pending = len(params)
        self.filled = [False] * len(params)
        self.work = None
        self.ready = False           # filled (copies + flags written) in this step


def _dense_strides(p: torch.Tensor) -> bool:
    """True when ``p``'s elements occupy exactly ``numel`` consecutive slots in some dimension order (contiguous, channels-last ...)."""
    if p.numel() == 0:
        return False
    dims = sorted((d for d in range(p.dim()) if p.shape[d] > 1), key=lambda d: p.stride(d))
    expect = 1
    for d in dims:
        if p.stride(d) != expect:
            return False
        expect *= p.shape[d]
    return True


class BucketedGradientAverager:
    """Overlapped, bucketed all-reduce (mean) of ``module``'s gradients.

    usage per step::

        averager.zero_grad()     # instead of optimizer.zero_grad(): drops the grads without freeing the buckets
        loss.backward()          # hooks launch one async all-reduce per completed bucket
        averager.finalize()      # wait, scale by 1/world, point p.grad at the reduced bucket views
        optimizer.step()

    ``p.grad`` of every parameter is a persistent VIEW of its flat bucket after ``finalize()`` (nothing is copied back).
    On the way in, autograd hands each parameter a freshly produced gradient tensor (it steals the buffer when
    ``p.grad is None``); the hook only counts it, and when the last gradient of a bucket has arrived ONE multi-tensor
    copy (``torch._foreach_copy_``) moves the bucket's gradients into the flat buffer — ~10 launches per step instead of
    one per parameter (362 for cfg 2), 2 x 122.7 MB of traffic = ~50 us at HBM speed against a ~230 ms backward.
    A parameter no rank produced a gradient for (the zero-weighted lowest deep-supervision head) keeps ``p.grad = None``,
    exactly as in a single-process step, so momentum / weight decay treat it the same at every world size.
    """

    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 32 << 20,
                 process_group: Optional[dist.ProcessGroup] = None, broadcast_from_rank0: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before building the averager")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.module = module
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()  # gradients become ready roughly in reverse registration order
        self.buckets: List[_Bucket] = []
        current, size = [], 0
        for p in params:
            nbytes = p.numel() * p.element_size()
            if current and (size + nbytes > bucket_bytes or p.dtype != current[0].dtype):
                self.buckets.append(_Bucket(current))
                current, size = [], 0
            current.append(p)
            size += nbytes
        if current:
            self.buckets.append(_Bucket(current))
        self._slot = {}
        self._hooks = []
        self._produced_cache = {}
        self._missing_cache = {}
        self._mismatch = torch.zeros((), dtype=torch.long, device=params[0].device) if params else None
        # True: the hooks only FILL the buckets (copies, flags); the collectives are launched by reduce_all() after backward — the form a step
        # captured as two hipGraphs around eager collectives needs (harness.SplitGraphedTrainStep); False: each bucket is all-reduced the
        # moment it is full, overlapping the rest of backward
        self.defer_collectives = False
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._slot[p] = (bi, pi)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        if broadcast_from_rank0:
            self.broadcast_state()

    # ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def broadcast_state(self) -> None:
        """One-time sync of parameters (incl. the frozen position tables) and buffers from rank 0."""
        tensors = [p.data for p in self.module.parameters()] + [b.data for b in self.module.buffers()]
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for group in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in group])
            dist.broadcast(flat, src=0, group=self.group)
            off = 0
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    @torch.no_grad()
    def zero_grad(self) -> None:
        """``optimizer.zero_grad(set_to_none=True)`` for the averaged parameters: the next backward writes fresh
        gradients (no accumulate-into-zeros pass), the flat buckets stay allocated."""
        for b in self.buckets:
            for p in b.params:
                p.grad = None

    @torch.no_grad()
    def _missing_index(self, b: _Bucket, pattern) -> torch.Tensor:
        """Device int64 positions of the parameters without a local gradient, one tensor per distinct pattern (built once:
        the host-to-device copy behind it blocks the host, which must not happen in the middle of every backward — and
        cannot happen inside a captured step, where the warm-up steps have filled this cache)."""
        key = (id(b), pattern)
        idx = self._missing_cache.get(key)
        if idx is None:
            idx = torch.tensor([pi for pi, f in enumerate(pattern) if not f], dtype=torch.long).to(b.flat.device)
            self._missing_cache[key] = idx
        return idx

    @torch.no_grad()
    def _launch(self, b: _Bucket) -> None:
        self._fill(b)
        if not self.defer_collectives:
            self._reduce(b)

    @torch.no_grad()
    def _reduce(self, b: _Bucket) -> None:
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @torch.no_grad()
    def _fill(self, b: _Bucket) -> None:
        src, dst, empty = [], [], []
        for pi, p in enumerate(b.params):
            if b.filled[pi]:
                if p.grad.data_ptr() != b.views[pi].data_ptr():     # accumulated in place into the view: nothing to move
                    src.append(p.grad.reshape(b.views[pi].shape) if p.grad.shape != b.views[pi].shape else p.grad)
                    dst.append(b.views[pi])
            else:
                empty.append(b.views[pi])
        if dst:
            torch._foreach_copy_(dst, src)
        # "was produced" flags, written with device-side fills only (two launches whatever the pattern): a host-to-device
        # copy from pageable memory would block the host until the stream reaches it — in the middle of backward — and
        # starve the launch queue behind it
        b.flags.fill_(1.0)
        if empty:
            torch._foreach_zero_(empty)                              # one multi-tensor launch, not one per parameter
            b.flags.index_fill_(0, self._missing_index(b, tuple(b.filled)), 0.0)
        b.ready = True

    @torch.no_grad()
    def _on_grad_ready(self, p: torch.nn.Parameter) -> None:
        bi, pi = self._slot[p]
        b = self.buckets[bi]
        if not b.filled[pi]:
            b.filled[pi] = True
            b.pending -= 1
        if b.pending == 0 and b.work is None and not b.ready:
            self._launch(b)

    @torch.no_grad()
    def fill_missing(self) -> None:
        """After backward: fill the buckets whose last gradient never came (a parameter without a gradient on this rank, e.g. the
        zero-weighted lowest deep-supervision head, gets zeros in its place)."""
        for b in self.buckets:
            if not b.ready:
                self._fill(b)

    @torch.no_grad()
    def reduce_all(self) -> None:
        """Launch the collective of every bucket that has none in flight (all of them under ``defer_collectives``).  Stateless with respect
        to the hooks' bookkeeping: a step replayed from hipGraphs never runs the hooks again, and calls this between the two graphs."""
        for b in self.buckets:
            if b.work is None:
                self._reduce(b)

    @torch.no_grad()
    def wait_all(self) -> None:
        """The current stream waits for the collectives (no host synchronisation with RCCL)."""
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None

    @torch.no_grad()
    def finish_local(self) -> None:
        """What follows the collectives, device-side only (capturable): the rank-consistency count, the 1 / world scale, ``p.grad`` pointed
        at the reduced bucket views (``None`` for parameters NO rank produced a gradient for), the hooks' bookkeeping reset.

        No host synchronisation after the first step of a given pattern: which parameters no rank produced a gradient for is read back once
        per distinct local pattern and remembered.  That is sound only while the pattern is STRUCTURAL, i.e. the same on every rank and every
        step (the zero-weighted head).  The summed flags show a violation on the device (0 < flag < world: some ranks produced the gradient,
        others did not); every step adds those to a device counter, and :meth:`check_consistency` — one host read, called by the caller
        wherever it synchronises anyway — raises if the contract was ever broken."""
        inv = 1.0 / self.world
        for bi, b in enumerate(self.buckets):
            if self.world > 1:
                self._mismatch.add_(((b.flags > 0) & (b.flags < self.world)).sum())
            b.flat[:b.n_grad].mul_(inv)
            produced = None
            if not all(b.filled):
                key = (bi, tuple(b.filled))
                if key not in self._produced_cache:
                    flags = b.flags.tolist()                         # the one host read of this pattern
                    if any(0 < f < self.world for f in flags):
                        raise RuntimeError("BucketedGradientAverager: ranks disagree on which parameters of bucket %d received a "
                                           "gradient (flags %s of world %d); only structurally unused parameters are supported"
                                           % (bi, [f for f in flags if 0 < f < self.world][:8], self.world))
                    self._produced_cache[key] = [f > 0 for f in flags]
                produced = self._produced_cache[key]
            for pi, p in enumerate(b.params):
                p.grad = b.views[pi] if (produced is None or produced[pi]) else None
            b.pending = len(b.params)
            b.filled = [False] * len(b.params)
            b.ready = False

    @torch.no_grad()
    def finalize(self) -> None:
        """Join the collectives: ``fill_missing`` -> ``reduce_all`` -> ``wait_all`` -> ``finish_local``.  With the default
        (``defer_collectives = False``) most buckets are already in flight — launched from the hooks during backward — and only the
        stragglers start here."""
        self.fill_missing()
        self.reduce_all()
        self.wait_all()
        self.finish_local()

    def check_consistency(self) -> None:
        """Host read of the device-side mismatch counter (see :meth:`finalize`); raises if, on any step so far, some ranks
        produced a gradient for a parameter and others did not."""
        n = int(self._mismatch.item()) if self.world > 1 else 0
        if n:
            raise RuntimeError("BucketedGradientAverager: %d (parameter, step) pairs received a gradient on some ranks only; the "
                               "remembered grad-is-None decisions may be stale" % n)

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def bytes_per_step(self) -> int:
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)


def _capture_safe_process_group_env(backend: str) -> None:
    """Settings of PyTorch's NCCL (= RCCL) process group for steps that are captured into a hipGraph, applied before the group exists
    and only where the user has not set the variable.

    ``TORCH_NCCL_CUDA_EVENT_CACHE=0``: by default the group recycles the events of finished collectives.  A collective captured into
    a graph records its end event in the CAPTURING stream; once that work object dies the event goes back to the pool and a later
    eager collective may receive it.  Seen on ROCm 7.2 / PyTorch 2.10, about once in ten runs of ``bench.py --force-averager --graph
    on``: the group's watchdog thread polling a work item's event got ``hipErrorCapturedEvent`` ("operation not permitted on an event
    last recorded in a capturing stream") and terminated the process.  With fresh events per collective no event ever crosses from a
    captured work item to an eager one (an event per collective costs microseconds; a step launches a handful)."""
    import os
    if backend == "nccl":
        os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")


def init_single_process_group(backend: Optional[str] = None) -> None:
    """A world-size-1 default group (``bench.py --force-averager``): the hooks, the bucket copies and the collective
    launches of the N > 1 path run on one GPU so that their overhead can be timed without an 8-GPU node."""
    import os
    import socket
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    _capture_safe_process_group_env(backend)
    dist.init_process_group(backend=backend, rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)


def init_process_group_from_env(backend: Optional[str] = None):
    """(rank, local_rank, world) from torchrun's environment; initialises the default group when
    WORLD_SIZE > 1.  ``backend`` defaults to nccl (= RCCL) with a GPU, gloo without."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl" and torch.cuda.device_count() > local_rank:
            torch.cuda.set_device(local_rank)
        _capture_safe_process_group_env(backend)
        # the first step of every rank runs the convolution library's find pass (tens of seconds alone, longer with N ranks sharing the
        # host and the library's user database); ranks drift apart by that much before their first collective meets.  The default
        # watchdog limit (10 min for nccl) is enough on an idle node — a generous one costs nothing and keeps a slow box from being
        # reported as a hang
        import datetime
        minutes = float(os.environ.get("NEXTOU_DIST_TIMEOUT_MIN", "30"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=minutes))
    return rank, local_rank, world
