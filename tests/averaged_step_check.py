#!/usr/bin/env python
"""Body of tests/test_gpu_ddp.py::test_averaged_step_over_rccl_matches_plain_step, run in its own interpreter (a process group and
RCCL's communicator should not outlive a test): the N > 1 train step — nextou_amd.ddp.BucketedGradientAverager: hooks, flat
buckets, asynchronous all-reduce on RCCL's stream, finalize — on a world-size-1 `nccl` (= RCCL) group on this box's one GPU,
eager AND captured into a hipGraph (harness.GraphedTrainStep), against the plain step of the same model on the same batch.

With one rank the mean over ranks is the identity, so every gradient must come out as the plain step's: what the averager adds
(bucket copies, the collective launch, the 1/world scale, p.grad pointing into the buckets, the remembered grad-is-None pattern
of the zero-weighted head) is exactly what is under test.  One JSON line: distances, the eager-vs-eager floor beside them.

The learning rate is ZERO in every run: two PLAIN runs of this tiny random-label network agree to ~1e-7 of the gradient scale in step 1
(the library's atomics) and then fall into different discrete outcomes — a neighbour or pooling tie decided the other way — that differ by
1e-3 of the weights one step later (measured: loss 3.55645 vs 3.55720 at step 2, either outcome in any mode), so trained weights cannot be
compared run to run.  With lr = 0 the weights stay put, every step's gradient must equal step 1's, and the optimizer's MOMENTUM BUFFERS —
a deterministic function of the gradients it was handed through `p.grad`, i.e. through the bucket views under capture — show that the
update consumed the right numbers.

    python tests/averaged_step_check.py [--backend nccl|gloo] [--workload tiny]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from nextou_amd import _lib  # noqa: E402
from nextou_amd.ddp import BucketedGradientAverager, init_single_process_group  # noqa: E402
from nextou_amd.harness import GraphedTrainStep, SplitGraphedTrainStep, downsample_targets, synthetic_batch  # noqa: E402

DEV = torch.device("cuda:0")


def make(workload, averaged):
    trainer, cfg, batch, classes = bench.build_trainer(workload, DEV, averaged, seed=7)
    bench.move_to(trainer, DEV)
    for group in trainer.optimizer.param_groups:
        group["lr"] = 0.0            # see the module docstring
    averager = BucketedGradientAverager(trainer.network, bucket_bytes=1 << 20) if averaged else None     # 1 MiB: several buckets for the tiny net
    data, target = synthetic_batch(cfg, 1, classes, batch, DEV, seed=11)
    targets = downsample_targets(target, bench._head_shapes(cfg))
    return trainer, bench.make_step(trainer, data, targets, averager), averager


def grads(trainer):
    """(flat gradient vector, names of the parameters whose grad is None) after a step"""
    named = [(n, p) for n, p in trainer.network.named_parameters() if p.requires_grad]
    return (torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().flatten().clone() for _, p in named]),
            [n for n, p in named if p.grad is None])


def weights(trainer):
    return torch.cat([p.detach().flatten() for p in trainer.network.parameters() if p.requires_grad])


def momentum(trainer):
    bufs = []
    for p in trainer.network.parameters():
        st = trainer.optimizer.state.get(p, {})
        if p.requires_grad and st.get("momentum_buffer") is not None:
            bufs.append(st["momentum_buffer"].detach().flatten())
    return torch.cat(bufs)


def dist_max(a, b):
    return float((a - b).abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--workload", default="tiny")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--captured-collectives", action="store_true",
                    help="also run the averaged step CAPTURED into a hipGraph (RCCL collectives inside the capture): the explicit opt-in of "
                         "bench.py --graph on, where PyTorch's RCCL watchdog thread can abort the process (DESIGN.md 6); without the flag the "
                         "modes are the defaults of every N (eager averaged step, captured plain step)")
    args = ap.parse_args()
    _lib.lib()
    torch.cuda.set_device(DEV)
    torch.backends.cudnn.benchmark = False        # immediate mode: the same library kernels in every model of this process
    init_single_process_group(args.backend)

    runs = {}
    modes = [("plain_a", False, False), ("plain_b", False, False), ("avg_eager", True, False), ("avg_split", True, "split"),
             ("plain_graph", False, True)]
    if args.captured_collectives:
        modes.insert(3, ("avg_graph", True, True))
    for name, averaged, graphed in modes:
        t, step, averager = make(args.workload, averaged)
        if graphed == "split":
            # the default of every N > 1 bench run: two graphs around eager collectives (no collective captured)
            averager.defer_collectives = True
            run = SplitGraphedTrainStep(step.part1, step.between, step.part2, warmup=1, network=t.network, loss=t.loss)
            done = 1
        elif graphed:
            # step 1 eager, then capture (executes nothing); the replays are steps 2..  — with the default 2 steps the comparison is
            # "second step replayed" against "second step eager": one update after identical first steps, before the tiny random-label
            # network's chaos (discrete neighbour choices, train-mode BN on 2 patches) has amplified the run-to-run noise
            run = GraphedTrainStep(step, warmup=1, network=t.network, loss=t.loss)
            done = 1
        else:
            run, done = step, 0
        first = None
        if not graphed:
            run()
            done = 1
            torch.cuda.synchronize()
            first = grads(t)
        while done < args.steps:
            loss = run()
            done += 1
        torch.cuda.synchronize()
        g_last, none_last = grads(t)
        if averager is not None:
            averager.check_consistency()
            averager.remove_hooks()
        runs[name] = {"first": first, "last": g_last, "none": none_last, "w": weights(t), "m": momentum(t), "loss": float(loss.detach()),
                      "buckets": None if averager is None else len(averager.buckets)}
    runs.setdefault("avg_graph", None)
    scale_g = float(runs["plain_a"]["last"].abs().max())

    def vs_plain(mode, key):
        return None if runs[mode] is None else min(dist_max(runs[mode][key], runs[k][key]) for k in ("plain_a", "plain_b"))

    out = {
        "hip_library_loaded": "libnextou_hip.so" in open("/proc/self/maps").read(),
        "backend": dist.get_backend(), "world_size": dist.get_world_size(), "steps": args.steps,
        "buckets": runs["avg_eager"]["buckets"],
        "grad_scale": scale_g,
        "grad_is_none_plain": runs["plain_a"]["none"], "grad_is_none_avg_eager": runs["avg_eager"]["none"],
        "grad_is_none_avg_graph": None if runs["avg_graph"] is None else runs["avg_graph"]["none"],
        "captured_collectives": bool(args.captured_collectives),
        # step 1: same weights, same batch -> the gradients themselves
        "grad1_plain_vs_plain": dist_max(runs["plain_a"]["first"][0], runs["plain_b"]["first"][0]),
        "grad1_avg_eager_vs_plain": dist_max(runs["avg_eager"]["first"][0], runs["plain_a"]["first"][0]),
        # last step (lr = 0: still the same weights): gradients and the optimizer's momentum buffers
        "grad_plain_vs_plain": dist_max(runs["plain_a"]["last"], runs["plain_b"]["last"]),
        "grad_avg_eager_vs_plain": min(dist_max(runs["avg_eager"]["last"], runs[k]["last"]) for k in ("plain_a", "plain_b")),
        "grad_avg_graph_vs_plain": vs_plain("avg_graph", "last"),
        "grad_avg_split_vs_plain": vs_plain("avg_split", "last"),
        "momentum_avg_split_vs_plain": vs_plain("avg_split", "m"),
        "grad_is_none_avg_split": runs["avg_split"]["none"],
        "grad_plain_graph_vs_plain": min(dist_max(runs["plain_graph"]["last"], runs[k]["last"]) for k in ("plain_a", "plain_b")),
        "momentum_scale": float(runs["plain_a"]["m"].abs().max()),
        "momentum_plain_vs_plain": dist_max(runs["plain_a"]["m"], runs["plain_b"]["m"]),
        "momentum_avg_eager_vs_plain": min(dist_max(runs["avg_eager"]["m"], runs[k]["m"]) for k in ("plain_a", "plain_b")),
        "momentum_avg_graph_vs_plain": vs_plain("avg_graph", "m"),
        "momentum_plain_graph_vs_plain": min(dist_max(runs["plain_graph"]["m"], runs[k]["m"]) for k in ("plain_a", "plain_b")),
        "weights_moved": max(dist_max(v["w"], runs["plain_a"]["w"]) for v in runs.values() if v is not None),
        "loss": {k: v["loss"] for k, v in runs.items() if v is not None},
    }
    dist.destroy_process_group()
    sys.stdout.flush()
    print(json.dumps(out), flush=True)       # (the last line: RCCL prints its banner when the communicator comes up, not after)


if __name__ == "__main__":
    main()
