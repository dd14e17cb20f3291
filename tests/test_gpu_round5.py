"""Round-5 regression tests on the MI355X for the advisor's findings that need a device."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from nextou_amd import _lib, graph_ops
    _lib.lib()
    assert "libnextou_hip.so" in open("/proc/self/maps").read(), "HIP extension not loaded into this process"
    return graph_ops


def test_fused_dice_declines_fractional_masks_and_clip_tp(ops, monkeypatch):
    """K5d treats `loss_mask` as a 0 / 1 flag and has no `clip_tp`: a float mask with fractional weights, or a Dice module that carries a
    clip, must take the base class (nnU-Net's own arithmetic) — same loss as with the fused path switched off; a bool mask stays fused."""
    from nextou_amd.loss.nnunet_losses import MemoryEfficientSoftDiceLoss, softmax_helper_dim1
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn((2, 14, 6, 10, 9), generator=gen) * 2).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    y = torch.randint(0, 14, (2, 1, 6, 10, 9), generator=gen).float().to(DEV)
    frac = torch.rand((2, 1, 6, 10, 9), generator=gen).to(DEV)                     # weights in (0, 1)
    dice = MemoryEfficientSoftDiceLoss(apply_nonlin=softmax_helper_dim1, batch_dice=False, do_bg=False, smooth=1e-5, ddp=False)
    calls = []
    real = ops.dice_stats
    monkeypatch.setattr(ops, "dice_stats", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    got = float(dice(x, y, loss_mask=frac))
    assert not calls, "a fractional float mask must not reach the 0 / 1 flag kernel"
    monkeypatch.setenv("NEXTOU_FUSED_DICE", "0")
    want = float(dice(x, y, loss_mask=frac))
    monkeypatch.delenv("NEXTOU_FUSED_DICE")
    assert got == want
    binary = float(dice(x, y, loss_mask=(frac > 0.5)))                            # bool mask: fused
    assert calls and abs(binary - got) > 1e-6                                      # ... and a different loss than the fractional weights give
    n = len(calls)
    dice.clip_tp = 0.5
    dice(x, y)
    assert len(calls) == n, "a Dice module with clip_tp must take the base class"


def test_float_atomic_aggregation_backward_variant(ops):
    """NEXTOU_MR_BWD=float keeps round 1's LDS float-atomic scatter (mr_bwd_arg_kernel) reachable through nextou_mr_aggregate_bwd_arg for
    A/B runs; the switch is read once per process, so the variant runs in its own interpreter (every kernel reachable
    through the C-ABI has a test): dx / dy against the float64 scatter, pooled and self graphs."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from nextou_amd import _lib, graph_ops
_lib.lib()
dev = torch.device("cuda:0")
worst = 0.0
for (B, C, N, M) in ((2, 12, 300, 64), (2, 7, 168, None), (1, 5, 2000, 5000)):
    g = torch.Generator().manual_seed(N)
    m = M or N
    gout = torch.randn(B, 2 * C, N, generator=g)
    arg = torch.randint(0, m, (B, C, N), generator=g).to(torch.int16)
    dx, dy = graph_ops._HIP.mr_bwd_arg(gout.to(dev), arg.to(dev), m, M is not None)
    scat = torch.zeros(B, C, m, dtype=torch.float64).scatter_add_(2, arg.long() & 0xffff, gout[:, 1::2].double())
    ident = (gout[:, 0::2] - gout[:, 1::2]).double()
    want_dx, want_dy = (ident + scat, None) if M is None else (ident, scat)
    worst = max(worst, float((dx.cpu().double() - want_dx).abs().max()) / float(want_dx.abs().max()))
    if want_dy is not None:
        worst = max(worst, float((dy.cpu().double() - want_dy).abs().max()) / float(want_dy.abs().max()))
import ctypes, json
L = _lib.lib()
L.nextou_profile_enable(8)
graph_ops._HIP.mr_bwd_arg(gout.to(dev), arg.to(dev), m, M is not None)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
L.nextou_profile_report(buf, len(buf))
print("RESULT", worst, json.loads(buf.value.decode())[0]["kernel"])
''' % REPO
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NEXTOU_MR_BWD="float"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[2].startswith("mr_bwd_arg_kernel"), line          # the float-atomic kernel really ran
    assert float(line[1]) <= 1e-5


@pytest.mark.parametrize("shape,kernel", [((2, 40, 6, 20, 24), (3, 3, 3)), ((2, 40, 5, 16, 12), (1, 3, 3)), ((1, 72, 4, 10, 12), (3, 3, 3))])
def test_dgrad_as_forward_with_own_filter_flip_matches_aten(shape, kernel, monkeypatch):
    """The data gradient of the plain stages' stride-1 convolutions is a forward convolution with the transposed, flipped filter; that
    filter comes from nextou_filter_flip_t (one launch) instead of ATen's transpose -> flip -> contiguous.  Same data-gradient bits as the ATen
    construction (the library convolution that consumes it is the same call), and the gradients agree with the library's own
    convolution_backward to summation-order round-off — with the filter parameter stored contiguous AND channels-last."""
    from nextou_amd import graph_ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    B, C = shape[:2]
    x = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    pad = tuple(k // 2 for k in kernel)
    for weight_cl in (False, True):
        w = (torch.randn((C, C) + kernel, generator=g) * 0.1).to(dev)
        if weight_cl:
            w = w.contiguous(memory_format=torch.channels_last_3d)
        gy = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("NEXTOU_FILTER_FLIP", mode)
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = graph_ops.conv_dgrad_as_forward(xr, wr, pad)
            res[mode] = (y.detach(),) + torch.autograd.grad(y, (xr, wr), gy)
        assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])     # same filter bits -> same data gradient
        # (the weight gradient does not involve the flipped filter; the library's kernel for it may sum with atomics)
        assert float((res["1"][2] - res["0"][2]).abs().max()) <= 1e-5 * float(res["0"][2].abs().max())
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = torch.nn.functional.conv3d(xr, wr, None, 1, pad)
        gx, gw = torch.autograd.grad(y, (xr, wr), gy)
        assert float((res["1"][1] - gx).abs().max()) <= 2e-5 * float(gx.abs().max())
        assert float((res["1"][2] - gw).abs().max()) <= 2e-4 * float(gw.abs().max())
