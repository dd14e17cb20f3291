#!/usr/bin/env python
"""Body of tests/test_gpu_parity2.py::test_graphed_train_step_matches_eager, run in its own interpreter: three tiny models from
one seed — two trained eagerly for 3 steps, one with 1 eager + 2 replayed steps (harness.GraphedTrainStep) — and one JSON line with
the distances between them."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from nextou_amd import _lib  # noqa: E402
from nextou_amd.harness import GraphedTrainStep, downsample_targets, synthetic_batch  # noqa: E402

DEV = torch.device("cuda:0")


def make(lr=None):
    trainer, cfg, batch, classes = bench.build_trainer("tiny", DEV, False, seed=7)
    bench.move_to(trainer, DEV)
    if lr is not None:
        for group in trainer.optimizer.param_groups:
            group["lr"] = lr
    data, target = synthetic_batch(cfg, 1, classes, batch, DEV, seed=11)
    targets = downsample_targets(target, bench._head_shapes(cfg))
    return trainer, bench.make_step(trainer, data, targets, None)


def weights(trainer):
    return torch.cat([p.detach().flatten() for p in trainer.network.parameters() if p.requires_grad])


def main():
    _lib.lib()
    torch.backends.cudnn.benchmark = True          # MIOpen's find mode, as bench.py and nnU-Net run it
    runs = []
    for _ in range(2):
        t, step = make()
        for _ in range(3):
            loss = step()
        torch.cuda.synchronize()
        runs.append((float(loss.detach()), weights(t)))
    tb, step_b = make()
    graphed = GraphedTrainStep(step_b, warmup=1)        # eager step 1, capture (executes nothing), then replays 2 and 3
    for _ in range(2):
        loss_b = graphed()
    torch.cuda.synchronize()
    wb = weights(tb)
    # Tight part (round 5): with the learning rate at ZERO the weights stay put, so the discrete outcomes trained runs fall into (a neighbour /
    # pooling tie decided the other way after a 1e-7 difference: the reason for the loose floors below) cannot occur: the replayed step's
    # gradients and the optimizer's momentum buffers must equal the eager step's to the library kernels' atomics noise.
    def frozen(graphed_mode):
        t, step = make(lr=0.0)
        run = GraphedTrainStep(step, warmup=1) if graphed_mode else step
        for _ in range(2 if graphed_mode else 3):
            run()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.detach().flatten() for p in t.network.parameters() if p.grad is not None])
        m = torch.cat([t.optimizer.state[p]["momentum_buffer"].flatten() for p in t.network.parameters()
                       if p in t.optimizer.state and t.optimizer.state[p].get("momentum_buffer") is not None])
        return g, m
    (ga, ma), (gb2, mb2), (gg, mg) = frozen(False), frozen(False), frozen(True)
    print(json.dumps({
        "frozen_grad_scale": float(ga.abs().max()), "frozen_momentum_scale": float(ma.abs().max()),
        "frozen_grad_eager_vs_eager": float((ga - gb2).abs().max()), "frozen_grad_replay_vs_eager": min(float((gg - ga).abs().max()), float((gg - gb2).abs().max())),
        "frozen_momentum_eager_vs_eager": float((ma - mb2).abs().max()),
        "frozen_momentum_replay_vs_eager": min(float((mg - ma).abs().max()), float((mg - mb2).abs().max())),
        "hip_library_loaded": "libnextou_hip.so" in open("/proc/self/maps").read(),
        "loss_eager": runs[0][0], "loss_replayed": float(loss_b.detach()),
        "loss_eager_vs_eager": abs(runs[0][0] - runs[1][0]),
        "weights_eager_vs_eager": float((runs[0][1] - runs[1][1]).abs().max()),
        "loss_replay_vs_eager": min(abs(runs[i][0] - float(loss_b.detach())) for i in range(2)),
        "weights_replay_vs_eager": min(float((runs[i][1] - wb).abs().max()) for i in range(2)),
    }))


if __name__ == "__main__":
    main()
