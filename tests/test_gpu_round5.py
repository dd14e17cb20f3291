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
