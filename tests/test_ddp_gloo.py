"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel gradient exchange.

The N > 1 path of bench.py is: per-rank forward/backward -> BucketedGradientAverager hooks fire
asynchronous all-reduces per flat bucket -> finalize() -> optimizer.  With per-replica BatchNorm the
averaged gradient must equal the mean of the two ranks' single-process gradients.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(seed=0):
    import oracle
    from nextou_amd import graph_ops
    from nextou_amd.harness import config_3d_fullres_nextou, downsample_targets, synthetic_batch
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU
    graph_ops.install_cpu_checker(oracle.CanonicalBackend)
    cfg = config_3d_fullres_nextou(patch_size=(32, 128, 128), base=6, max_features=24, batch_size=1)
    torch.manual_seed(seed)
    trainer = nnUNetTrainer_NexToU(cfg, 4, log=None).initialize()
    return trainer, cfg, downsample_targets, synthetic_batch


def _grads_single(rank_seed):
    torch.set_num_threads(2)   # same MKLDNN partitioning as the workers: identical conv arithmetic
    trainer, cfg, downsample_targets, synthetic_batch = _build(seed=0)
    data, target = synthetic_batch(cfg, 1, 4, 1, torch.device("cpu"), seed=1234 + rank_seed)
    out = trainer.network(data)
    trainer.loss(out, downsample_targets(target, out)).backward()
    return {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
            for n, p in trainer.network.named_parameters() if p.requires_grad}


def _worker(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from nextou_amd.ddp import BucketedGradientAverager, init_process_group_from_env
    r, lr, w = init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    # rank 1 starts from different weights: broadcast_state must overwrite them with rank 0's
    trainer, cfg, downsample_targets, synthetic_batch = _build(seed=rank * 17)
    averager = BucketedGradientAverager(trainer.network, bucket_bytes=64 << 10)   # many small buckets
    assert len(averager.buckets) > 3
    n_trainable = sum(p.numel() for p in trainer.network.parameters() if p.requires_grad)
    n_tensors = sum(1 for p in trainer.network.parameters() if p.requires_grad)
    # frozen relative_pos tables are not reduced; one "was produced" flag per parameter rides in the same collectives
    assert averager.bytes_per_step == 4 * (n_trainable + n_tensors)
    data, target = synthetic_batch(cfg, 1, 4, 1, torch.device("cpu"), seed=1234 + rank)
    for step in range(3):    # later steps check the per-step bucket reset, with both ways of dropping the gradients
        if step == 1:
            trainer.optimizer.zero_grad(set_to_none=True)
        else:
            averager.zero_grad()
        out = trainer.network(data)
        trainer.loss(out, downsample_targets(target, out)).backward()
        averager.finalize()
        named = [(n, p) for n, p in trainer.network.named_parameters() if p.requires_grad]
        # a parameter NO rank produced a gradient for keeps grad = None, as in a single-process step
        assert [n for n, p in named if p.grad is None] == ["decoder.seg_layers.0.weight", "decoder.seg_layers.0.bias"]
        # every other gradient is a view of its flat bucket (nothing is copied back)
        assert all(p.grad.data_ptr() == averager.buckets[averager._slot[p][0]].views[averager._slot[p][1]].data_ptr()
                   for n, p in named if p.grad is not None)
        if step == 0:
            grads = {n: p.grad.clone() for n, p in named if p.grad is not None}
        else:
            assert all(torch.equal(grads[n], p.grad) for n, p in named if p.grad is not None)
    # round 6: the collectives deferred to after backward — the form harness.SplitGraphedTrainStep replays as graph / eager all-reduces / graph —
    # gives the same reduced gradients, and the hooks launch nothing
    averager.defer_collectives = True
    for step in range(2):
        averager.zero_grad()
        out = trainer.network(data)
        trainer.loss(out, downsample_targets(target, out)).backward()
        assert all(b.work is None for b in averager.buckets)
        averager.fill_missing()
        averager.reduce_all()
        assert all(b.work is not None for b in averager.buckets)
        averager.wait_all()
        averager.finish_local()
        assert [n for n, p in named if p.grad is None] == ["decoder.seg_layers.0.weight", "decoder.seg_layers.0.bias"]
        assert all(torch.equal(grads[n], p.grad) for n, p in named if p.grad is not None)
    averager.defer_collectives = False
    averager.check_consistency()
    state = {k: v.clone() for k, v in trainer.network.state_dict().items()}
    torch.save({"grads": grads}, os.path.join(tmp, "rank%d.pt" % rank))
    # all ranks hold the same parameters and the same reduced gradients
    flat = torch.cat([g.reshape(-1) for g in grads.values()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert torch.equal(other[0], other[1])
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bucketed_average_equals_mean_of_single_process_grads(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "rank0.pt"))["grads"]
    # reference: each rank's gradient computed alone with rank-0 weights (seed 0), then averaged
    g0, g1 = _grads_single(0), _grads_single(1)
    worst = 0.0
    for n in got:
        want = 0.5 * (g0[n] + g1[n])
        scale = max(1.0, float(want.abs().max()))
        worst = max(worst, float((got[n] - want).abs().max()) / scale)
    assert worst <= 1e-5, worst
    # the zero-weighted lowest deep-supervision head receives no gradient: finalize() must not hang, and leaves None
    assert "decoder.seg_layers.0.weight" not in got and len(got) == len(g0) - 2
