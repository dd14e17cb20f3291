"""An overflowing step under fp16 autocast must cost a skipped optimizer step, not the process.

nnU-Net v2 trains under ``torch.autocast("cuda")`` (float16) with a ``GradScaler``: now and then an activation overflows, the loss and the
gradients come out inf / NaN, ``scaler.step`` skips the update and ``scaler.update`` lowers the scale.  The reference survives that because
``torch.topk`` returns ids inside the candidate set whatever the distances hold (torch_edge.py:58-110).  Here the graph kernels sit on
that path with hand-written selections: a NaN distance that never enters a list left sentinel (2 147 483 647) or unwritten (-1 …) neighbour
ids — measured with the clamp of csrc/knn_graph.hip `finite_or_last` compiled out, tests/test_gpu_knn_small.py::test_non_finite_features_… —
and the aggregation after it (NexToU_Encoder_Decoder.py:401-418) gathered out of bounds: silently on this GPU (8 GB past a tensor is still
mapped memory of the process), a memory fault wherever it is not.  These tests hold the whole step to the reference's behaviour."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _trainer():
    from nextou_amd import _lib
    from nextou_amd.harness import config_3d_fullres_nextou
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU
    _lib.lib()
    assert "libnextou_hip.so" in open("/proc/self/maps").read(), "HIP extension not loaded into this process"
    cfg = config_3d_fullres_nextou(patch_size=(32, 128, 128), base=6, max_features=48, batch_size=2)
    torch.manual_seed(0)
    return nnUNetTrainer_NexToU(cfg, 5, device=DEV, log=None).initialize(), cfg


def test_overflow_under_fp16_autocast_skips_the_step_and_training_goes_on():
    from nextou_amd.harness import downsample_targets, synthetic_batch
    tr, cfg = _trainer()
    net, opt = tr.network.train(), tr.optimizer
    data, target = synthetic_batch(cfg, 1, 5, 2, DEV, seed=5)
    with torch.no_grad():
        targets = downsample_targets(target, net(data))
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)

    def step(x):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = tr.loss(net(x), targets)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        scaler.step(opt)
        scaler.update()
        return float(loss)

    first = step(data)
    assert first == first and abs(first) < 1e4, first                       # a finite loss, the scale kept
    assert scaler.get_scale() == 1024.0
    params = [p for p in net.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in params]
    poisoned = data * 3e5                                                   # far beyond fp16's 65 504 after the first convolutions
    bad = step(poisoned)
    torch.cuda.synchronize()                                                # (a fault in any kernel of the step would surface here)
    assert not (bad == bad and abs(bad) < 1e30), "the poisoned step was meant to overflow (loss %r)" % bad
    assert scaler.get_scale() < 1024.0, "GradScaler saw no inf / NaN gradient"
    for p, b in zip(params, before):
        assert torch.equal(p.detach(), b), "an overflowing step must not move a parameter"
    # batch statistics normalise in training mode, so the (now NaN) running statistics of the reference's BatchNorm do not enter the next step
    again = step(data)
    assert again == again and abs(again) < 1e4, again
    assert any(not torch.equal(p.detach(), b) for p, b in zip(params, before)), "the step after the overflow did not update anything"


def test_overflow_in_fp32_gives_nan_not_a_fault():
    """The same poisoned batch without autocast: every graph stage sees inf / NaN features in fp32; the step must finish (NaN loss)."""
    from nextou_amd.harness import downsample_targets, synthetic_batch
    tr, cfg = _trainer()
    net = tr.network.train()
    data, target = synthetic_batch(cfg, 1, 5, 2, DEV, seed=6)
    with torch.no_grad():
        targets = downsample_targets(target, net(data))
    x = data.clone()
    x[0, 0, 3, 40:60, 40:60] = float("inf")
    x[1, 0, 7, 10, 10] = float("nan")
    loss = tr.loss(net(x), targets)
    loss.backward()
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(loss)), float(loss)
