"""K6 for small tensors in one launch each way (csrc/norm_act.hip bn_one_cl_kernel / bn_one_rows_kernel, round 6): statistics, finalisation and
apply by workgroups that own whole channels — the graph-stage norms of stages 4 / 5 (reference torch_nn.py:84-90,
NexToU_Encoder_Decoder.py:384-390, 710-720, 833-842).  Bars: against the multi-launch kernels (the same arithmetic, another order of the float64
partial sums) 2e-6 of each tensor's scale; against float64 autograd 2e-5 / 1e-4; bit-reproducible; running statistics as F.batch_norm updates them."""
import ctypes
import json

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from nextou_amd import _lib, graph_ops
    _lib.lib()
    assert "libnextou_hip.so" in open("/proc/self/maps").read(), "HIP extension not loaded into this process"
    return graph_ops


def _labels(fn):
    from nextou_amd import _lib
    L = _lib.lib()
    L.nextou_profile_enable(64)
    out = fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.nextou_profile_report(buf, len(buf))
    L.nextou_profile_enable(0)
    return out, [r["kernel"] for r in json.loads(buf.value.decode())] if n else []


CASES = [
    # shape, channels-last, instance period
    ((2, 324, 4, 7, 6), True, 0),          # cfg-2 stage 5: fc1 / fc2 / FFN output
    ((2, 1296, 4, 7, 6), True, 0),         # the FFN's hidden tensor: two float4 columns per workgroup
    ((1, 520, 8, 14, 8), True, 0),         # 896 rows, a ragged last column block (520 = 65 x 8)
    ((1, 8, 3, 5, 7), True, 0),            # ragged rows, fewer rows than row lanes
    ((3, 12, 9, 11), True, 0),             # 2-D
    ((1, 2 * 648, 168), False, 648),       # Pool MRConv's InstanceNorm at stage 5: the caller's (1, B C, S) view
    ((2, 10, 160), False, 0),              # channel-major batch statistics over two rows per channel
    ((1, 6, 4), False, 3),                 # one float4 per row
]


# (instance norm always uses the statistics of its input: no eval variant of the two period cases)
@pytest.mark.parametrize("shape,cl,period,training", [c + (t,) for c in CASES for t in (True, False) if t or not c[2]])
def test_one_launch_norm_equals_the_multi_launch_path(ops, monkeypatch, shape, cl, period, training):
    g = torch.Generator().manual_seed(len(shape) * 100 + shape[1])
    x = torch.randn(shape, generator=g).to(DEV) * 1.5 + 0.3
    if cl:
        x = x.contiguous(memory_format=torch.channels_last if len(shape) == 4 else torch.channels_last_3d)
    gy = torch.randn(shape, generator=g).to(DEV)
    gy = gy.contiguous(memory_format=torch.channels_last if len(shape) == 4 else torch.channels_last_3d) if cl else gy
    P = period or shape[1]
    w, b, pre = (1 + 0.2 * torch.randn(P, generator=g)).to(DEV), (0.3 * torch.randn(P, generator=g)).to(DEV), (0.1 * torch.randn(P, generator=g)).to(DEV)
    rm0, rv0 = torch.randn(shape[1], generator=g).to(DEV), (0.5 + torch.rand(shape[1], generator=g)).to(DEV)
    H = ops._HIP

    def run():
        rm, rv = (None, None) if period else (rm0.clone(), rv0.clone())
        y, mean, invstd = H.norm_act_fwd(x, w, b, rm, rv, training, 0.1, 1e-5, 0.01, period, None if period else pre, channels_last=cl)
        gx, gw, gb = H.norm_act_bwd(x, gy, w, b, mean, invstd, training, 0.01, period, 1e-5, channels_last=cl)
        return [y, mean, invstd, gx, gw, gb] + ([] if period else [rm, rv])

    one, labels = _labels(run)
    assert [l.split("[")[0] for l in labels] == (["bn_one_cl_kernel<fwd,%d>" % (2 if shape[1] >= 512 else 1), "bn_one_cl_kernel<bwd,%d>" % (2 if shape[1] >= 512 else 1)]
                                                 if cl else ["bn_one_rows_kernel<fwd>", "bn_one_rows_kernel<bwd>"]), labels
    again = run()
    for a_, b_ in zip(one, again):
        assert torch.equal(a_, b_), "not bit-reproducible"
    monkeypatch.setenv("NEXTOU_K6_ONE_MAX", "0")               # read per call: every tensor is "too large" -> the multi-launch kernels
    multi, labels2 = _labels(run)
    assert not any(l.startswith("bn_one_") for l in labels2) and len(labels2) >= (3 if training else 2), labels2
    for name, a_, e_ in zip(("y", "mean", "invstd", "gx", "gweight", "gbias", "running_mean", "running_var"), one, multi):
        assert a_.shape == e_.shape and a_.stride() == e_.stride(), name
        assert float((a_ - e_).abs().max()) <= 2e-6 * float(e_.abs().max()) + 1e-30, name


@pytest.mark.parametrize("shape", [(2, 324, 4, 7, 6), (2, 520, 3, 5, 4)])     # (both below the 2-MB bound of plan_one)
def test_one_launch_norm_against_float64_autograd(ops, shape):
    g = torch.Generator().manual_seed(shape[1])
    mf = torch.channels_last_3d
    x = (torch.randn(shape, generator=g) * 2 - 0.5).to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(shape[1], generator=g)).to(DEV).requires_grad_(True)
    b = (0.3 * torch.randn(shape[1], generator=g)).to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(shape[1], device=DEV), torch.ones(shape[1], device=DEV)
    gy = torch.randn(shape, generator=g).to(DEV).contiguous(memory_format=mf)
    (y, grads), labels = _labels(lambda: (lambda yy: (yy, torch.autograd.grad(yy, [x, w, b], gy)))(
        ops.norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, 0.01)))
    assert any(l.startswith("bn_one_cl_kernel<fwd") for l in labels) and any(l.startswith("bn_one_cl_kernel<bwd") for l in labels), labels
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    rmd, rvd = torch.zeros(shape[1], device=DEV, dtype=torch.float64), torch.ones(shape[1], device=DEV, dtype=torch.float64)
    yd = F.leaky_relu(F.batch_norm(xd, rmd, rvd, wd, bd, True, 0.1, 1e-5), 0.01)
    gd = torch.autograd.grad(yd, [xd, wd, bd], gy.double())
    assert float((y.double() - yd).abs().max()) <= 2e-5 * float(yd.abs().max())
    for a_, e_ in zip(grads, gd):
        assert float((a_.double() - e_).abs().max()) <= 1e-4 * float(e_.abs().max())
    assert torch.allclose(rm.double(), rmd, rtol=1e-5, atol=1e-7) and torch.allclose(rv.double(), rvd, rtol=1e-5, atol=1e-7)


def test_one_launch_norm_on_guard_pages(ops):
    from tools.guard_alloc import GuardScope
    H = ops._HIP
    for shape, cl, period in (((2, 324, 2, 5, 3), True, 0), ((1, 2 * 12, 40), False, 12)):
        g = torch.Generator().manual_seed(1)
        x, gy = torch.randn(shape, generator=g).to(DEV), torch.randn(shape, generator=g).to(DEV)
        if cl:
            x, gy = (t.contiguous(memory_format=torch.channels_last_3d) for t in (x, gy))
        P = period or shape[1]
        w, b = torch.randn(P, generator=g).to(DEV), torch.randn(P, generator=g).to(DEV)

        def launch(x, gy, w, b):
            y, mean, invstd = H.norm_act_fwd(x, w, b, None, None, True, 0.1, 1e-5, 0.01, period, None, channels_last=cl)
            return (y, mean, invstd) + tuple(H.norm_act_bwd(x, gy, w, b, mean, invstd, True, 0.01, period, 1e-5, channels_last=cl))

        want = launch(x, gy, w, b)
        torch.cuda.synchronize()
        for flush in ("end", "start"):
            scope = GuardScope(flush=flush, align=16)
            try:
                gin = [scope.like(t.clone(memory_format=torch.preserve_format)) for t in (x, gy, w, b)]
                with scope.patched_outputs():
                    got = launch(*gin)
                torch.cuda.synchronize()
                for a_, e_ in zip(got, want):
                    assert torch.equal(a_, e_)
            finally:
                torch.cuda.synchronize()
                scope.close()
