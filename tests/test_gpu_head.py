"""K8 — the segmentation heads on own kernels (csrc/head_rows.hip, include/nextou_hip.h "K8"): biased 1x1 convolutions from a
stage's features to the class logits (reference NexToU_Encoder_Decoder.py:253-258, :311-337) and their autograd, against the
float64 convolution of the same operands; bit-reproducibility; operands in guard-page buffers (tools/guard_alloc.py: the byte
after an operand's last element is unmapped — the library convolution this replaces fails exactly that test,
tools/conv_bwd_fault_repro.py); the model takes the kernels at every head."""
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


def _mf(t):
    return {4: torch.channels_last, 5: torch.channels_last_3d}[t.dim()]


_SHAPES = [
    # (B, C, spatial, L)
    (2, 40, (4, 16, 12), 14),       # cfg 2's full-resolution head: padded 33 -> 40 features, 14 classes
    (2, 72, (3, 9, 8), 14),
    (1, 8, (3, 5, 7), 14),          # the tiny workload's head (P % 16 != 0)
    (2, 324, (4, 7, 6), 14),        # bottleneck-side head: 21 channel tiles
    (1, 33, (2, 5, 9), 3),          # unpadded channel count: scalar loads / stores, odd class count
    (1, 132, (1, 1, 5), 17),        # two class tiles, P < 16
    (2, 24, (9, 11), 5),            # 2-D model
    (1, 16, (33, 31), 64),          # four class tiles (the data gradient's limit)
    (1, 4, (2, 2, 2), 1),
]


def _reference(x, w, b, gy):
    xd = x.double().cpu().requires_grad_(True)
    wd = w.double().cpu().requires_grad_(True)
    bd = b.double().cpu().requires_grad_(True)
    conv = F.conv3d if x.dim() == 5 else F.conv2d
    y = conv(xd, wd, bd)
    gx, gw, gb = torch.autograd.grad(y, (xd, wd, bd), gy.double().cpu())
    return y.detach(), gx, gw, gb


@pytest.mark.parametrize("B,C,sp,L", _SHAPES)
def test_head_rows_vs_float64_convolution(ops, B, C, sp, L):
    """forward within 2 ulp-sums of the float64 convolution (|y - y64| <= 1e-6 * sum_c |x_c w_lc|), gradients likewise; the weight and bias
    gradients (sums over all points) rtol 2e-5 of their scale; repeated launches bit-identical."""
    g = torch.Generator().manual_seed(C * 100 + L)
    x = (torch.randn((B, C) + sp, generator=g) * 1.3 + 0.2).to(DEV)
    x = x.contiguous(memory_format=_mf(x)).requires_grad_(True)
    w = (torch.randn((L, C) + (1,) * len(sp), generator=g) * 0.2).to(DEV).requires_grad_(True)
    b = torch.randn(L, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn((B, L) + sp, generator=g).to(DEV)
    gy = gy.contiguous(memory_format=_mf(gy))

    y = ops.head_rows(x, w, b)
    assert y.shape == (B, L) + sp and y.is_contiguous(memory_format=_mf(y))
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
    y64, gx64, gw64, gb64 = _reference(x.detach(), w.detach(), b.detach(), gy)

    # error bars scale with the sums of absolute products (what a float32 summation in any order is bounded by)
    ymag, gxmag, gwmag, gbmag = _reference(x.detach().abs(), w.detach().abs(), b.detach().abs(), gy.abs())
    assert float(((y.detach().cpu().double() - y64).abs() / ymag).max()) <= 1e-6
    assert float(((gx.cpu().double() - gx64).abs() / (gxmag + 1e-30)).max()) <= 1e-6
    assert gw.shape == w.shape and gb.shape == b.shape
    assert float(((gw.cpu().double() - gw64).abs() / (gwmag + 1e-30)).max()) <= 2e-6
    assert float(((gb.cpu().double() - gb64).abs() / (gbmag + 1e-30)).max()) <= 2e-6

    # bit-reproducible: fixed summation orders everywhere, no atomics
    y2 = ops.head_rows(x, w, b)
    gx2, gw2, gb2 = torch.autograd.grad(y2, (x, w, b), gy)
    assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_head_rows_without_bias_and_partial_gradients(ops):
    """bias=None; only the input needs a gradient (frozen head) / only the parameters do (first layer)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 40, 3, 8, 8), generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((14, 40, 1, 1, 1), generator=g) * 0.2).to(DEV)
    gy = torch.randn((2, 14, 3, 8, 8), generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    want = F.conv3d(x.double(), w.double())
    y = ops.head_rows(x, w, None)
    assert float((y.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    xr = x.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(ops.head_rows(xr, w, None), (xr,), gy)
    xd = x.double().requires_grad_(True)
    (gx64,) = torch.autograd.grad(F.conv3d(xd, w.double()), (xd,), gy.double())
    assert float((gx.double() - gx64).abs().max()) <= 1e-5 * float(gx64.abs().max())
    wr = w.clone().requires_grad_(True)
    (gw,) = torch.autograd.grad(ops.head_rows(x, wr, None), (wr,), gy)
    wd = w.double().requires_grad_(True)
    (gw64,) = torch.autograd.grad(F.conv3d(x.double(), wd), (wd,), gy.double())
    assert float((gw.double() - gw64).abs().max()) <= 2e-5 * float(gw64.abs().max())


@pytest.mark.parametrize("B,C,sp,L", [(2, 40, (4, 16, 12), 14), (1, 8, (3, 5, 7), 14), (1, 33, (2, 5, 9), 3), (64, 8, (128, 128), 14)])
def test_head_rows_never_touch_memory_outside_their_operands(ops, B, C, sp, L):
    """Every operand ends (and, second pass, starts) on an UNMAPPED page: a read or write one element outside any of them kills the
    process with `Memory access fault by GPU`.  The last shape is the N > 1 step's faulting convolution (profiles/r05_n_gt_1.md)."""
    from tools.guard_alloc import guarded_like
    g = torch.Generator().manual_seed(C + L)
    x0 = torch.randn((B, C) + sp, generator=g).to(DEV)
    x0 = x0.contiguous(memory_format=_mf(x0))
    w0 = (torch.randn((L, C), generator=g) * 0.2).to(DEV)
    b0 = torch.randn(L, generator=g).to(DEV)
    gy0 = torch.randn((B, L) + sp, generator=g).to(DEV)
    gy0 = gy0.contiguous(memory_format=_mf(gy0))
    want_y = ops._HIP.head_rows_fwd(x0, w0, b0)
    want = ops._HIP.head_rows_bwd(gy0, x0, w0, True, True)
    for flush, align in (("end", 16), ("start", 16), ("end", 4)):
        # align 16: the kernels the plain tensors took (LDS-staged where the shape allows) -> the same bits; align 4: operands that may start
        # off a 16-byte boundary take the direct kernels (another, equally fixed, summation order for the weight gradient)
        x, w, b, gy = (guarded_like(t, flush=flush, align=16 if t is x0 else align) for t in (x0, w0, b0, gy0))
        y = ops._HIP.head_rows_fwd(x, w, b)
        got = ops._HIP.head_rows_bwd(gy, x, w, True, True)
        torch.cuda.synchronize()
        assert torch.equal(y, want_y)
        for a, e in zip(got, want):
            if align == 16:
                assert torch.equal(a, e)
            else:
                assert float((a - e).abs().max()) <= 1e-5 * (float(e.abs().max()) + 1e-20)


def test_model_heads_run_on_k8(ops, monkeypatch):
    """Every deep-supervision head of the tiny 3-D NexToU goes through K8 in a train step, forward and backward; NEXTOU_HEAD_ROWS=0
    hands them back to the library convolution and the logits agree to convolution round-off."""
    from nextou_amd.harness import config_3d_fullres_nextou, downsample_targets, synthetic_batch
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU
    cfg = config_3d_fullres_nextou(patch_size=(32, 128, 128), base=6, max_features=48, batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainer_NexToU(cfg, 14, device=DEV, log=None).initialize()
    data, target = synthetic_batch(cfg, 1, 14, 2, DEV)
    calls = {"fwd": 0, "bwd": 0}
    real_f, real_b = ops._HIP.head_rows_fwd, ops._HIP.head_rows_bwd
    monkeypatch.setattr(ops._HIP, "head_rows_fwd", staticmethod(lambda *a: (calls.__setitem__("fwd", calls["fwd"] + 1), real_f(*a))[1]))
    monkeypatch.setattr(ops._HIP, "head_rows_bwd", staticmethod(lambda *a: (calls.__setitem__("bwd", calls["bwd"] + 1), real_b(*a))[1]))
    tr.network.train()
    state = {k: v.clone() for k, v in tr.network.state_dict().items()}
    outs = tr.network(data)
    loss = tr.loss(outs, downsample_targets(target, outs))
    loss.backward()
    heads = len(outs)
    # the lowest-resolution head has weight 0 in the deep-supervision loss: autograd never reaches its backward
    assert calls == {"fwd": heads, "bwd": heads - 1}, calls
    g_own = [None if p.grad is None else p.grad.clone() for p in tr.network.decoder.seg_layers.parameters()]
    # A/B against the library convolution on the same weights and running statistics
    monkeypatch.setenv("NEXTOU_HEAD_ROWS", "0")
    tr.network.load_state_dict(state)
    tr.network.zero_grad(set_to_none=True)
    outs_lib = tr.network(data)
    tr.loss(outs_lib, downsample_targets(target, outs_lib)).backward()
    assert calls == {"fwd": heads, "bwd": heads - 1}          # unchanged: the second pass ran on the library convolution
    for a, b in zip(outs, outs_lib):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    for a, p in zip(g_own, tr.network.decoder.seg_layers.parameters()):
        assert (a is None) == (p.grad is None)
        if a is not None:
            assert float((a - p.grad).abs().max()) <= 1e-4 * (float(p.grad.abs().max()) + 1e-12)
