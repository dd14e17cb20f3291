"""Host-side plumbing of the step glue, on CPU tensors against a stand-in library (no GPU, no HIP code involved):

* ClipSGD: the device table (parameter / gradient / buffer pointers, element counts), the chunk list and the argument order of the
  three C-ABI calls.  The stand-in reads the table through the pointers it is handed — exactly what the kernels do — and applies the
  documented update rule with numpy, so a wrong row, chunk, stride check or argument position shows up as a wrong parameter.
* _UpConvCat: the filter permutations, the GEMM shapes and the tap order of the transposed convolution written as a GEMM + pixel shuffle,
  with the two shuffle passes emulated in torch from the header's formula, against torch.nn.functional.conv_transpose{2,3}d.

The kernels themselves are tested on the GPU (tests/test_gpu_step_glue.py, tests/test_gpu_guard.py).
"""
import ctypes

import numpy as np
import pytest
import torch


def _arr(ptr, n, ct):
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(n,))


class _StandInLib:
    """nextou_device_write_i64 / nextou_grad_norm_clip_coef / nextou_clip_sgd_update on host memory, per include/nextou_hip.h."""

    def __init__(self):
        self.calls = []

    def nextou_device_write_i64(self, dst, host_values, n, stream):
        _arr(dst, n, ctypes.c_int64)[:] = _arr(host_values.value if hasattr(host_values, "value") else host_values, n, ctypes.c_int64)
        return 0

    def _walk(self, table, n_tensors, chunks, n_chunks, chunk_elems):
        tab = _arr(table, 4 * n_tensors, ctypes.c_int64).reshape(n_tensors, 4)
        ck = _arr(chunks, 2 * n_chunks, ctypes.c_int32).reshape(n_chunks, 2)
        seen = [set() for _ in range(n_tensors)]
        for row, c in ck:
            p, g, m, n = (int(v) for v in tab[row])
            lo = int(c) * chunk_elems
            cnt = min(chunk_elems, n - lo)
            assert 0 < cnt and c not in seen[row]
            seen[row].add(int(c))
            yield p, g, m, lo, cnt
        for row in range(n_tensors):
            assert len(seen[row]) == -(-int(tab[row][3]) // chunk_elems), "every element in exactly one chunk"

    def nextou_grad_norm_clip_coef(self, table, n_tensors, chunks, n_chunks, chunk_elems, total, partial, max_norm, norm_coef, stream):
        self.calls.append("norm")
        acc, elems = 0.0, 0
        part = _arr(partial, n_chunks, ctypes.c_double)
        for i, (p, g, m, lo, cnt) in enumerate(self._walk(table, n_tensors, chunks, n_chunks, chunk_elems)):
            gv = _arr(g, lo + cnt, ctypes.c_float)[lo:].astype(np.float64)
            part[i] = float((gv * gv).sum())
            acc += part[i]
            elems += cnt
        assert elems == total
        out = _arr(norm_coef, 2, ctypes.c_float)
        out[0] = np.float32(np.sqrt(acc))
        out[1] = min(np.float32(max_norm) / (out[0] + np.float32(1e-6)), np.float32(1.0))
        return 0

    def nextou_clip_sgd_update(self, table, n_tensors, chunks, n_chunks, chunk_elems, total, norm_coef, lr, lr_dev, momentum, wd, nesterov,
                               stream):
        self.calls.append("update")
        coef = _arr(norm_coef, 2, ctypes.c_float)[1] if norm_coef else None
        if lr_dev:
            lr = float(_arr(lr_dev, 1, ctypes.c_float)[0])
        f = np.float32
        for p, g, m, lo, cnt in self._walk(table, n_tensors, chunks, n_chunks, chunk_elems):
            pv = _arr(p, lo + cnt, ctypes.c_float)[lo:]
            gv = _arr(g, lo + cnt, ctypes.c_float)[lo:]
            if coef is not None:
                gv[:] = gv * coef
            d = gv.copy()
            if wd:
                d = (d.astype(np.float64) + np.float64(f(wd)) * pv).astype(f)
            if m and momentum:
                mv = _arr(m, lo + cnt, ctypes.c_float)[lo:]
                mv[:] = mv * f(momentum)
                mv[:] = mv + d
                d = (d.astype(np.float64) + np.float64(f(momentum)) * mv).astype(f) if nesterov else mv.copy()
            pv[:] = (pv.astype(np.float64) - np.float64(f(lr)) * d).astype(f)
        return 0

    def nextou_last_error(self):
        return b""


@pytest.fixture()
def stand_in(monkeypatch):
    from nextou_amd import _lib
    fake = _StandInLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    return fake


def _params():
    g = torch.Generator().manual_seed(3)
    shapes = [(1,), (37,), (14, 33, 1, 1, 1), (6, 8, 3, 3, 3), (40000,), (16384,), (16385,)]
    ts = [torch.randn(s, generator=g) for s in shapes]
    ts[3] = ts[3].contiguous(memory_format=torch.channels_last_3d)
    base = torch.randn((1005,), generator=g)
    ts.append(base[1:1002])
    return ts


@pytest.mark.parametrize("momentum,nesterov,wd", [(0.99, True, 3e-5), (0.9, False, 0.0), (0.0, False, 1e-4)])
def test_clip_sgd_table_plumbing_against_torch(stand_in, momentum, nesterov, wd):
    from nextou_amd.optim import ClipSGD
    ps = [torch.nn.Parameter(t.clone(memory_format=torch.preserve_format)) for t in _params()]
    qs = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in ps]
    own = ClipSGD(ps, 0.01, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    own._device_type = "cpu"
    ref = torch.optim.SGD(qs, 0.01, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    g = torch.Generator().manual_seed(9)
    for step in range(4):
        for i, (p, q) in enumerate(zip(ps, qs)):
            if i == 4 and step < 2:
                p.grad = q.grad = None
                continue
            gr = torch.randn(p.shape, generator=g) * (3.0 if step % 2 else 0.01)
            p.grad = torch.empty_like(p).copy_(gr)
            q.grad = torch.empty_like(q).copy_(gr)
        clip = step != 2
        if clip:
            n_own = own.clip_and_step(12.0)
            n_ref = torch.nn.utils.clip_grad_norm_(qs, 12.0)
            assert abs(float(n_own) - float(n_ref)) <= 1e-6 * float(n_ref)
        else:
            own.step()
        ref.step()
        assert own.last_path == "own"
        for i, (p, q) in enumerate(zip(ps, qs)):
            assert p.stride() == q.stride()
            tol = 2e-6 * float(q.detach().abs().max())
            assert float((p.detach() - q.detach()).abs().max()) <= tol, (i, step)
            if q.grad is not None:
                assert float((p.grad - q.grad).abs().max()) <= 2e-6 * float(q.grad.abs().max()), (i, step)
            if momentum and q in ref.state and "momentum_buffer" in ref.state[q] and ref.state[q]["momentum_buffer"] is not None:
                mo = own.state[p]["momentum_buffer"]
                assert mo.stride() == p.stride()
                assert float((mo - ref.state[q]["momentum_buffer"]).abs().max()) <= 2e-6 * float(ref.state[q]["momentum_buffer"].abs().max())
    assert stand_in.calls.count("update") == 4 and stand_in.calls.count("norm") == 3
    # the table is rebuilt only when a pointer moved: same gradients tensors -> cache hit
    tables = own._tables[0][1]
    own.step()
    assert own._tables[0][1] is tables
    # torch.optim.SGD's state layout
    if momentum:
        assert set(own.state_dict()["state"][1].keys()) == {"momentum_buffer"}


def test_clip_sgd_hands_over_what_the_kernels_do_not_take(stand_in):
    from nextou_amd.optim import ClipSGD
    ps = [torch.nn.Parameter(t.clone(memory_format=torch.preserve_format)) for t in _params()[:5]]
    own = ClipSGD(ps, 0.01, momentum=0.9, nesterov=True)
    own._device_type = "cpu"
    for p in ps:
        p.grad = torch.empty_like(p).normal_()
    # another element order than the channels-last parameter: copied into the parameter's order, then the kernels take the step
    qs = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in ps]
    for p, q in zip(ps, qs):
        q.grad = p.grad.clone(memory_format=torch.preserve_format)
    ps[3].grad = ps[3].grad.contiguous()
    assert ps[3].grad.stride() != ps[3].stride()
    # the stride of a size-1 dimension is arbitrary (autograd's layout contract ignores it): still the same element order
    ps[2].grad = torch.as_strided(ps[2].grad.clone(), ps[2].shape, (33, 1, 462, 462, 462))
    own.clip_and_step(5.0)
    assert own.last_path == "own" and own.last_reason is None and ps[3].grad.stride() == ps[3].stride()
    ref = torch.optim.SGD(qs, 0.01, momentum=0.9, nesterov=True)
    torch.nn.utils.clip_grad_norm_(qs, 5.0)
    ref.step()
    for p, q in zip(ps, qs):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * float(q.detach().abs().max())
    stand_in.calls.clear()
    own2 = ClipSGD([torch.nn.Parameter(torch.randn(5).double())], 0.01)
    own2._device_type = "cpu"
    own2.param_groups[0]["params"][0].grad = torch.randn(5).double()
    own2.step()
    assert own2.last_path == "torch" and "float32" in own2.last_reason
    own3 = ClipSGD([torch.nn.Parameter(torch.randn(5))], 0.01, momentum=0.9, dampening=0.1)
    own3._device_type = "cpu"
    own3.param_groups[0]["params"][0].grad = torch.randn(5)
    own3.step()
    assert own3.last_path == "torch" and "dampening" in own3.last_reason and stand_in.calls == []


def test_clip_sgd_survives_pickle_and_deepcopy(stand_in):
    import copy
    import pickle
    from nextou_amd.optim import ClipSGD
    ps = [torch.nn.Parameter(t.clone()) for t in _params()[:3]]
    own = ClipSGD(ps, 0.01, momentum=0.9, nesterov=True)
    own._device_type = "cpu"
    for p in ps:
        p.grad = torch.randn_like(p)
    own.step()
    assert own.last_path == "own"
    for other in (pickle.loads(pickle.dumps(own)), copy.deepcopy(own)):
        other._device_type = "cpu"
        assert isinstance(other, torch.optim.SGD) and other._tables == {} or other is not own
        for p in other.param_groups[0]["params"]:
            p.grad = torch.randn_like(p)
        other.step()                                   # builds its own table for its own tensors
        assert other.last_path == "own"
        assert all("momentum_buffer" in other.state[p] for p in other.param_groups[0]["params"])


# ------------------------------------------------------------------------------------------------ _UpConvCat on emulated passes
def _upconv_row(row, D2, H2, W2, sd, sh, sw):
    """include/nextou_hip.h, nextou_upconv_cat_rows: output row -> p_in * T + t."""
    w2, r1 = row % W2, row // W2
    h2, r2 = r1 % H2, r1 // H2
    d2, b = r2 % D2, r2 // D2
    t = ((d2 % sd) * sh + h2 % sh) * sw + w2 % sw
    p_in = ((b * (D2 // sd) + d2 // sd) * (H2 // sh) + h2 // sh) * (W2 // sw) + w2 // sw
    return p_in * (sd * sh * sw) + t


class _EmulatedHip:
    """The four launches _UpConvCat makes, in torch on the host, from their C-ABI descriptions (rows = channels-last memory)."""

    @staticmethod
    def _rows(t):
        return t.permute(0, *range(2, t.dim()), 1).reshape(-1, t.shape[1])

    @staticmethod
    def _from_rows(rows, shape):
        b, c = shape[0], shape[1]
        sp = tuple(shape[2:])
        mf = {4: torch.channels_last, 5: torch.channels_last_3d}[len(shape)]
        return rows.reshape((b,) + sp + (c,)).permute(0, len(shape) - 1, *range(1, len(shape) - 1)).contiguous(memory_format=mf)

    def pw_rows(self, x_cl, w2, bias, groups):
        assert groups == 1 and bias is None and w2.is_contiguous()
        return self._from_rows(self._rows(x_cl) @ w2.t(), (x_cl.shape[0], w2.shape[0]) + tuple(x_cl.shape[2:]))

    def pw_wgrad(self, gy_cl, x_cl, groups):
        return self._rows(gy_cl).t() @ self._rows(x_cl)

    def _index(self, B, sp_in, stride):
        d, h, w = ((1,) + tuple(sp_in))[-3:]
        sd, sh, sw = ((1,) + tuple(stride))[-3:]
        P = B * d * sd * h * sh * w * sw
        return torch.tensor([_upconv_row(r, d * sd, h * sh, w * sw, sd, sh, sw) for r in range(P)])

    def upconv_cat_rows(self, y2, bias, skip, stride):
        T = int(np.prod(stride))
        c1 = y2.shape[1] // T
        idx = self._index(y2.shape[0], y2.shape[2:], stride)
        a = self._rows(y2).reshape(-1, c1)[idx]
        if bias is not None:
            a = a + bias
        out_rows = torch.cat((a, self._rows(skip)), 1)
        return self._from_rows(out_rows, (skip.shape[0], c1 + skip.shape[1]) + tuple(skip.shape[2:]))

    def upconv_cat_rows_bwd(self, g, c1, sp_in, stride):
        if c1 > 128:
            return None
        T = int(np.prod(stride))
        idx = self._index(g.shape[0], sp_in, stride)
        rows = self._rows(g)[:, :c1]
        gy2 = torch.empty((rows.shape[0], c1), dtype=rows.dtype)
        gy2[idx] = rows
        return self._from_rows(gy2.reshape(-1, T * c1), (g.shape[0], T * c1) + tuple(sp_in)), rows.sum(0)


@pytest.mark.parametrize("cin,cout,c2,sp,stride", [
    (8, 4, 4, (3, 4, 5), (1, 2, 2)), (12, 8, 8, (2, 3, 2), (2, 2, 2)), (8, 132, 4, (1, 2, 2), (2, 2, 2)), (8, 4, 8, (5, 3), (2, 2)),
    (4, 4, 4, (2, 2, 3), (2, 1, 2))])
def test_upconv_cat_is_the_transposed_convolution(monkeypatch, cin, cout, c2, sp, stride):
    from nextou_amd import graph_ops
    monkeypatch.setattr(graph_ops, "_HIP", _EmulatedHip())
    n = len(sp)
    mf = {2: torch.channels_last, 3: torch.channels_last_3d}[n]
    g = torch.Generator().manual_seed(cin * cout)
    x = torch.randn((2, cin) + sp, generator=g, dtype=torch.float64).contiguous(memory_format=mf).requires_grad_(True)
    w = torch.randn((cin, cout) + stride, generator=g, dtype=torch.float64).contiguous(memory_format=mf).requires_grad_(True)
    b = torch.randn(cout, generator=g, dtype=torch.float64).requires_grad_(True)
    sp_out = tuple(d * s for d, s in zip(sp, stride))
    skip = torch.randn((2, c2) + sp_out, generator=g, dtype=torch.float64).contiguous(memory_format=mf).requires_grad_(True)
    go = torch.randn((2, cout + c2) + sp_out, generator=g, dtype=torch.float64).contiguous(memory_format=mf)
    conv = torch.nn.functional.conv_transpose3d if n == 3 else torch.nn.functional.conv_transpose2d
    for bias in (b, None):
        out = graph_ops.upconv_cat(x, w, bias, skip, stride)
        ref = torch.cat((conv(x, w, bias, stride), skip), 1)
        assert out.shape == ref.shape and float((out - ref).abs().max()) <= 1e-12
        ins = [x, w, skip] + ([bias] if bias is not None else [])
        got = torch.autograd.grad(out, ins, go)
        want = torch.autograd.grad(ref, ins, go)
        for a, e in zip(got, want):
            assert a.shape == e.shape and float((a - e).abs().max()) <= 1e-11 * (1 + float(e.abs().max()))
    # a contiguous (not channels-last) filter gives the same result
    wc = w.detach().contiguous().requires_grad_(True)
    out = graph_ops.upconv_cat(x, wc, b, skip, stride)
    assert float((out - torch.cat((conv(x, wc, b, stride), skip), 1)).abs().max()) <= 1e-12
    gw = torch.autograd.grad(out, [wc], go)[0]
    assert float((gw - torch.autograd.grad(torch.cat((conv(x, wc, b, stride), skip), 1), [wc], go)[0]).abs().max()) <= 1e-10


@pytest.mark.parametrize("strided_views", ["1", "0"])
def test_clip_sgd_on_the_gradient_averager_s_bucket_views(stand_in, monkeypatch, strided_views):
    """Under the gradient averager ``p.grad`` is a view of a flat bucket.  Default (round 6): the view carries the PARAMETER's
    strides, so a channels-last filter's gradient sits in the bucket in the filter's own element order and ClipSGD walks it in place —
    ``p.grad`` stays inside the bucket.  ``NEXTOU_DDP_STRIDED_VIEWS=0``: a plain slice, another element order; ClipSGD copies such a
    gradient into the parameter's order first.  Either way the step is the kernels' (world-size-1 gloo group, real
    BucketedGradientAverager, the kernels' stand-in) and equals torch's."""
    monkeypatch.setenv("NEXTOU_DDP_STRIDED_VIEWS", strided_views)
    import socket

    import torch.distributed as dist
    from nextou_amd.ddp import BucketedGradientAverager
    from nextou_amd.optim import ClipSGD
    if dist.is_initialized():
        pytest.skip("another test left a process group up")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Conv3d(4, 6, 3, padding=1), torch.nn.Conv3d(6, 2, 1))
        net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last_3d)       # layout.filters_to_channels_last
        ref = torch.nn.Sequential(torch.nn.Conv3d(4, 6, 3, padding=1), torch.nn.Conv3d(6, 2, 1))
        ref.load_state_dict(net.state_dict())
        averager = BucketedGradientAverager(net, bucket_bytes=1 << 10)
        own = ClipSGD(net.parameters(), 0.05, momentum=0.9, nesterov=True, weight_decay=1e-4)
        own._device_type = "cpu"
        plain = torch.optim.SGD(ref.parameters(), 0.05, momentum=0.9, nesterov=True, weight_decay=1e-4)
        x = torch.randn(2, 4, 3, 5, 5)
        for _ in range(3):
            averager.zero_grad()
            net(x).square().mean().backward()
            averager.finalize()
            bucket = averager.buckets[averager._slot[net[0].weight][0]].flat
            inside = lambda t: bucket.data_ptr() <= t.data_ptr() < bucket.data_ptr() + bucket.numel() * 4      # noqa: E731
            assert inside(net[0].weight.grad)
            assert (net[0].weight.grad.stride() == net[0].weight.stride()) is (strided_views == "1")
            own.clip_and_step(0.05)
            assert own.last_path == "own", own.last_reason
            assert net[0].weight.grad.stride() == net[0].weight.stride()
            assert inside(net[0].weight.grad) is (strided_views == "1")          # "0": ClipSGD re-pointed p.grad at its copy
            plain.zero_grad(set_to_none=True)
            ref(x).square().mean().backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
            plain.step()
            for p, q in zip(net.parameters(), ref.parameters()):
                assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * float(q.detach().abs().max())
        averager.remove_hooks()
    finally:
        dist.destroy_process_group()
