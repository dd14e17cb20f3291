"""Guard-page device buffers: a tensor whose storage is bracketed by UNMAPPED virtual address ranges.

HIP's virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess) lets a test place a buffer
so that the byte after its last element (and the byte before its first) belongs to no mapping: a kernel that reads or writes
past an operand takes a `Memory access fault by GPU` deterministically, instead of only when the caching allocator happens
to have left the neighbouring page unmapped.  Used by tools/conv_bwd_fault_repro.py (the N > 1 step's fault, VERDICT r4
item 1a) and by tests/test_gpu_guard.py (no kernel of libnextou_hip.so reads past its operands).

Test / tool infrastructure only: nothing under nextou_amd/ imports this.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, byref, c_int, c_size_t, c_ubyte, c_ulonglong, c_ushort, c_void_p

import numpy as np
import torch


class _Location(Structure):
    _fields_ = [("type", c_int), ("id", c_int)]


class _AllocFlags(Structure):
    _fields_ = [("compressionType", c_ubyte), ("gpuDirectRDMACapable", c_ubyte), ("usage", c_ushort)]


class _AllocationProp(Structure):            # hipMemAllocationProp (hip_runtime_api.h)
    _fields_ = [("type", c_int), ("requestedHandleType", c_int), ("location", _Location),
                ("win32HandleMetaData", c_void_p), ("allocFlags", _AllocFlags)]


class _AccessDesc(Structure):                # hipMemAccessDesc
    _fields_ = [("location", _Location), ("flags", c_int)]


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        # torch has already loaded its libamdhip64: dlopen by SONAME returns that runtime
        for name in ("libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"):
            try:
                _HIP = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _HIP is None:
            raise RuntimeError("libamdhip64 not found")
    return _HIP


def _ck(code, what):
    if code != 0:
        raise RuntimeError("%s failed with hipError %d" % (what, code))


class GuardedBuffer:
    """``nbytes`` of device memory whose END (``flush="end"``) or START (``flush="start"``) touches an unmapped range.
    ``align``: alignment kept for the data pointer when the buffer is flushed to the end (the gap to the guard is then
    ``< align`` bytes)."""

    def __init__(self, nbytes: int, device: int = 0, flush: str = "end", align: int = 4):
        hip = _hip()
        torch.cuda.init()
        prop = _AllocationProp()
        prop.type = 1                      # hipMemAllocationTypePinned
        prop.requestedHandleType = 0       # hipMemHandleTypeNone
        prop.location = _Location(1, device)   # hipMemLocationTypeDevice
        gran = c_size_t(0)
        _ck(hip.hipMemGetAllocationGranularity(byref(gran), byref(prop), c_int(0)), "hipMemGetAllocationGranularity")
        self.gran = int(gran.value) or (2 << 20)
        self.nbytes = int(nbytes)
        self.mapped = max(self.gran, (self.nbytes + self.gran - 1) // self.gran * self.gran)
        self.reserved = self.mapped + 2 * self.gran
        base = c_void_p(0)
        _ck(hip.hipMemAddressReserve(byref(base), c_size_t(self.reserved), c_size_t(self.gran), c_void_p(0), c_ulonglong(0)),
            "hipMemAddressReserve")
        self.base = int(base.value)
        self.handle = c_void_p(0)
        _ck(hip.hipMemCreate(byref(self.handle), c_size_t(self.mapped), byref(prop), c_ulonglong(0)), "hipMemCreate")
        self.map_ptr = self.base + self.gran
        _ck(hip.hipMemMap(c_void_p(self.map_ptr), c_size_t(self.mapped), c_size_t(0), self.handle, c_ulonglong(0)), "hipMemMap")
        desc = _AccessDesc(_Location(1, device), 3)     # hipMemAccessFlagsProtReadWrite
        _ck(hip.hipMemSetAccess(c_void_p(self.map_ptr), c_size_t(self.mapped), byref(desc), c_size_t(1)), "hipMemSetAccess")
        if flush == "end":
            self.ptr = (self.map_ptr + self.mapped - self.nbytes) // align * align
        else:
            self.ptr = self.map_ptr
        self.device = device
        self._alive = True

    # numpy-style description torch.as_tensor understands
    def _view(self, shape, dtype: torch.dtype, strides_elems=None):
        np_dtype = {torch.float32: "<f4", torch.float64: "<f8", torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1",
                    torch.int16: "<i2", torch.float16: "<f2", torch.uint16: "<u2"}[dtype]
        item = np.dtype(np_dtype).itemsize
        iface = {"shape": tuple(int(s) for s in shape), "typestr": np_dtype, "data": (self.ptr, False), "version": 2,
                 "strides": None if strides_elems is None else tuple(int(s) * item for s in strides_elems)}
        holder = type("_Iface", (), {"__cuda_array_interface__": iface})()
        t = torch.as_tensor(holder, device="cuda:%d" % self.device)
        t._guard_owner = self            # keep the mapping alive as long as the tensor
        return t

    def tensor(self, shape, dtype=torch.float32, strides=None):
        n = 1
        if strides is None:
            for s in shape:
                n *= int(s)
        else:
            n = 1 + sum((int(s) - 1) * int(st) for s, st in zip(shape, strides))
        item = torch.empty((), dtype=dtype).element_size()
        if n * item > self.nbytes:
            raise ValueError("view of %d bytes over a %d-byte buffer" % (n * item, self.nbytes))
        return self._view(shape, dtype, strides)

    def free(self):
        if not self._alive:
            return
        hip = _hip()
        torch.cuda.synchronize()
        hip.hipMemUnmap(c_void_p(self.map_ptr), c_size_t(self.mapped))
        hip.hipMemRelease(self.handle)
        # The address range stays RESERVED for the life of the process: a later buffer mapped at a recycled virtual address was seen
        # to read back stale bytes of the earlier mapping (MI355X, ROCm 7.2: rows of a kernel's output replaced by the old tenant's
        # data in ~1 of 3 launches; never with fresh addresses) — an artefact of unmap -> map at the same address, not of the kernels
        # under test.  Virtual address space is the only thing this leaks.
        self._alive = False


def guarded_like(t: torch.Tensor, flush: str = "end", align: int = 4) -> torch.Tensor:
    """A copy of the dense tensor ``t`` (any memory format) in a guard-page buffer: same sizes, strides and values."""
    if not t.is_cuda:
        raise ValueError("guarded_like: device tensor expected")
    span = 1 + sum((s - 1) * st for s, st in zip(t.shape, t.stride())) if t.numel() else 0
    if span != t.numel():
        raise ValueError("guarded_like: dense tensors only")
    buf = GuardedBuffer(max(span, 1) * t.element_size(), t.device.index or 0, flush, align)
    g = buf.tensor(tuple(t.shape), t.dtype, tuple(t.stride()))
    g.copy_(t)
    return g



class GuardScope:
    """Guard-page tensors with a common lifetime: ``like`` copies a dense tensor into one, ``patched_outputs`` makes every
    ``torch.empty`` / ``torch.empty_like`` of a device tensor issued by Python code inside the block (the ctypes wrappers of
    nextou_amd/graph_ops.py allocate their outputs and workspaces that way) a guard-page tensor too, ``close`` unmaps them all."""

    def __init__(self, flush: str = "end", align: int = 16):
        self.flush, self.align, self.buffers = flush, align, []

    def empty(self, shape, dtype=torch.float32, strides=None, device=0):
        shape = tuple(int(s) for s in shape)
        if strides is None:
            strides, n = [], 1
            for s in reversed(shape):
                strides.append(n)
                n *= max(s, 1)
            strides = tuple(reversed(strides))
        span = 1 + sum((s - 1) * st for s, st in zip(shape, strides)) if all(shape) else 0
        item = torch.empty((), dtype=dtype).element_size()
        buf = GuardedBuffer(max(span, 1) * item, device, self.flush, self.align)     # end-flush: the gap to the guard is < align bytes
        self.buffers.append(buf)
        return buf.tensor(shape, dtype, strides)

    def like(self, t: torch.Tensor) -> torch.Tensor:
        g = self.empty(tuple(t.shape), t.dtype, tuple(t.stride()), t.device.index or 0)
        g.copy_(t)
        return g

    def patched_outputs(self):
        import contextlib
        scope = self

        @contextlib.contextmanager
        def ctx():
            real_empty, real_like = torch.empty, torch.empty_like

            def empty(*size, **kw):
                dev = kw.get("device")
                if dev is None or torch.device(dev).type != "cuda":
                    return real_empty(*size, **kw)
                meta = real_empty(*size, **dict(kw, device="meta"))
                return scope.empty(tuple(meta.shape), meta.dtype, tuple(meta.stride()), torch.device(dev).index or 0)

            def empty_like(t, **kw):
                if not t.is_cuda or kw:
                    return real_like(t, **kw)
                meta = real_like(t, device="meta")
                return scope.empty(tuple(meta.shape), meta.dtype, tuple(meta.stride()), t.device.index or 0)

            torch.empty, torch.empty_like = empty, empty_like
            try:
                yield scope
            finally:
                torch.empty, torch.empty_like = real_empty, real_like
        return ctx()

    def close(self):
        for b in self.buffers:
            b.free()
        self.buffers = []


for _fn, _args in (("hipMemGetAllocationGranularity", [POINTER(c_size_t), POINTER(_AllocationProp), c_int]),
                   ("hipMemAddressReserve", [POINTER(c_void_p), c_size_t, c_size_t, c_void_p, c_ulonglong]),
                   ("hipMemCreate", [POINTER(c_void_p), c_size_t, POINTER(_AllocationProp), c_ulonglong]),
                   ("hipMemMap", [c_void_p, c_size_t, c_size_t, c_void_p, c_ulonglong]),
                   ("hipMemSetAccess", [c_void_p, c_size_t, POINTER(_AccessDesc), c_size_t]),
                   ("hipMemUnmap", [c_void_p, c_size_t]), ("hipMemRelease", [c_void_p]),
                   ("hipMemAddressFree", [c_void_p, c_size_t])):
    try:
        getattr(_hip(), _fn).argtypes = _args
        getattr(_hip(), _fn).restype = c_int
    except (RuntimeError, AttributeError):
        pass
