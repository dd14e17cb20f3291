"""(An experiment kept reachable: OFF by default, NEXTOU_SKIP_FORK=1 — measured neutral on the cfg-2 step, graph_ops.skip_fork.)
The skip connection's gradient without autograd's add pass (graph_ops.skip_fork, nextou_norm_act_bwd_two; reference
NexToU_Encoder_Decoder.py:143-150 `skips.append(x)`, :311-337 `torch.cat((x, skip), 1)`): K6's channels-last backward reads the two incoming
gradients — the next stage's (dense) and the concatenation's (a channel range of wider rows) — and sums them on load.  Bar: identical
results to the plain path (autograd's aten::add, then nextou_norm_act_bwd) up to the fp32 rounding of the float64 partial sums' inputs —
the gradient sum itself is the same fp32 add — stated as 1e-6 of each tensor's scale; bit-reproducible; guard pages."""
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


def _case(shape, c_cat, c_off, seed):
    g = torch.Generator().manual_seed(seed)
    mf = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
    C = shape[1]
    x = torch.randn(shape, generator=g).to(DEV).contiguous(memory_format=mf)
    g1 = torch.randn(shape, generator=g).to(DEV).contiguous(memory_format=mf)
    wide = torch.randn((shape[0], c_cat) + tuple(shape[2:]), generator=g).to(DEV).contiguous(memory_format=mf)
    w, b = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    return x, g1, wide.narrow(1, c_off, C), w, b


@pytest.mark.parametrize("shape,c_cat,c_off", [
    ((2, 40, 6, 20, 24), 80, 40),        # the stage-0 skip of cfg 2 (padded 33 -> 40 channels), concatenated behind 40 up-convolution channels
    ((2, 72, 4, 10, 12), 144, 72),       # stage 1
    ((1, 8, 3, 5, 7), 32, 12),           # ragged rows, a range in the middle of the wide rows
    ((3, 128, 9, 11), 132, 4),           # 2-D, the widest rows of the bn_cl kernels
])
@pytest.mark.parametrize("training", [True, False])
def test_two_gradient_backward_matches_add_then_backward(ops, shape, c_cat, c_off, training):
    x, g1, g2, w, b = _case(shape, c_cat, c_off, 5)
    assert ops.two_gradients_eligible(x, g1, g2)
    H = ops._HIP
    y, mean, invstd = H.norm_act_fwd(x, w, b, None, None, True, 0.1, 1e-5, 0.01, 0, None, channels_last=True)
    want = H.norm_act_bwd(x, (g1 + g2).contiguous(memory_format=torch.channels_last if len(shape) == 4 else torch.channels_last_3d), w, b, mean,
                          invstd, training, 0.01, 0, 1e-5, channels_last=True)
    got = H.norm_act_bwd_two(x, g1, g2, w, b, mean, invstd, training, 0.01)
    again = H.norm_act_bwd_two(x, g1, g2, w, b, mean, invstd, training, 0.01)
    for a, e, r in zip(got, want, again):
        assert a.shape == e.shape and a.stride() == e.stride()
        assert float((a - e).abs().max()) <= 1e-6 * float(e.abs().max()) + 1e-30
        assert torch.equal(a, r)


def test_skip_fork_in_autograd_and_fallbacks(ops, monkeypatch):
    """norm -> two consumers: gradients of everything equal with the fork on, off, and where the second gradient is not in the kernel's form."""
    from nextou_amd.network_architecture.norm_act import BatchNormAct3d
    torch.manual_seed(0)
    norm = BatchNormAct3d(40).to(DEV).train()
    norm.negative_slope = 0.01
    x0 = torch.randn(2, 40, 4, 12, 16, device=DEV).contiguous(memory_format=torch.channels_last_3d)
    up = torch.randn(2, 40, 4, 12, 16, device=DEV).contiguous(memory_format=torch.channels_last_3d)
    wa = torch.randn_like(x0)
    wc = torch.randn(2, 80, 4, 12, 16, device=DEV).contiguous(memory_format=torch.channels_last_3d)

    def run(fork, cat_form=True):
        monkeypatch.setenv("NEXTOU_SKIP_FORK", "1" if fork else "0")
        x = x0.clone().requires_grad_(True)
        norm.zero_grad(set_to_none=True)
        y = norm(x)
        a, skip = ops.skip_fork(y)
        assert (a is not y) is fork
        if cat_form:
            loss = (a * wa).sum() + (torch.cat((up, skip), 1) * wc).sum()        # the skip's gradient: a narrow view of the cat's
        else:
            loss = (a * wa).sum() + (skip.sin() * wa).sum()                       # a dense gradient: the in-backward add
        loss.backward()
        return x.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()

    for cat_form in (True, False):
        ref = run(False, cat_form)
        got = run(True, cat_form)
        for a, e in zip(got, ref):
            assert float((a - e).abs().max()) <= 2e-6 * float(e.abs().max())


def test_model_step_with_and_without_the_fork(ops, monkeypatch):
    """One training forward + backward of the tiny 3-D network: bn_cl_bwd_*2 kernels run for the plain stages' skips, parameter gradients of
    the first stage agree with the un-forked run to what two runs of this network agree to anyway."""
    import ctypes
    import json
    from nextou_amd import _lib
    from nextou_amd.harness import config_3d_fullres_nextou, downsample_targets, synthetic_batch
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU
    cfg = config_3d_fullres_nextou(patch_size=(32, 128, 128), base=6, max_features=48, batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainer_NexToU(cfg, 5, device=DEV, log=None).initialize()
    data, target = synthetic_batch(cfg, 1, 5, 2, DEV, seed=5)
    losses = {}
    for fork in ("1", "0"):
        monkeypatch.setenv("NEXTOU_SKIP_FORK", fork)
        tr.network.zero_grad(set_to_none=True)
        _lib.lib().nextou_profile_enable(4096)
        outs = tr.network(data)
        loss = tr.loss(outs, downsample_targets(target, outs))
        loss.backward()
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 20)
        n = _lib.lib().nextou_profile_report(buf, len(buf))
        _lib.lib().nextou_profile_enable(0)
        labels = [r["kernel"] for r in json.loads(buf.value.decode())] if n else []
        assert any(l.startswith("bn_cl_bwd_apply2_kernel") for l in labels) is (fork == "1"), labels
        losses[fork] = float(loss.detach())
        assert all(torch.isfinite(p.grad).all() for p in tr.network.parameters() if p.grad is not None)
    assert abs(losses["1"] - losses["0"]) <= 1e-2 * abs(losses["0"])


def test_two_gradient_backward_on_guard_pages(ops):
    from tools.guard_alloc import GuardScope
    x, g1, g2, w, b = _case((2, 40, 3, 9, 13), 80, 40, 7)
    H = ops._HIP
    y, mean, invstd = H.norm_act_fwd(x, w, b, None, None, True, 0.1, 1e-5, 0.01, 0, None, channels_last=True)
    want = H.norm_act_bwd_two(x, g1, g2, w, b, mean, invstd, True, 0.01)
    torch.cuda.synchronize()
    wide = g2._base if g2._base is not None else g2
    for flush in ("end", "start"):
        scope = GuardScope(flush=flush, align=16)
        try:
            gx_, g1_, wide_, w_, b_, m_, i_ = (scope.like(t.clone(memory_format=torch.preserve_format)) for t in (x, g1, wide, w, b, mean, invstd))
            with scope.patched_outputs():
                got = H.norm_act_bwd_two(gx_, g1_, wide_.narrow(1, 40, 40), w_, b_, m_, i_, True, 0.01)
            torch.cuda.synchronize()
            for a, e in zip(got, want):
                assert torch.equal(a, e)
        finally:
            torch.cuda.synchronize()
            scope.close()
