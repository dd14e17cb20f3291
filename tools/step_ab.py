#!/usr/bin/env python
"""What do the round-5 step-glue changes buy inside the replayed cfg-2 step?  One process, one box: the bench's step and variants of it
are captured as hipGraphs side by side and replayed in alternation, so box-to-box spread (+-1.5 ms) and clock drift cancel.

    python tools/step_ab.py [--workload cfg2] [--rounds 3] [--steps 10] [--variants base,sgd,sgd+ncs,all,no_opt] [--eager]

variants:  base     = the step as of round 4: clip_grad_norm_(12) + torch.optim.SGD(fused), the decoder concatenation's backward as
                      narrow().contiguous() + channel_sum, the up-convolutions on the library
           sgd      = base with clip + SGD on nextou_amd.optim.ClipSGD (own norm / update kernels)
           sgd+ncs  = sgd with the concatenation's backward as one pass (nextou_narrow_copy_sum)
           all      = sgd+ncs with the up-convolutions as K7 GEMMs + shuffle-concatenation (graph_ops._UpConvCat): bench.py's default step
           no_opt   = all without clip and optimizer   (probe, not a valid bench step: the floor any optimizer work sits on)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from nextou_amd.harness import GraphedTrainStep, downsample_targets, synthetic_batch  # noqa: E402


VARIANTS = {     # name -> (optimizer, environment while the step is built, warmed up, captured and run eagerly)
    "base": ("torch", {"NEXTOU_NARROW_COPY_SUM": "0", "NEXTOU_UPCONV_GEMM": "0"}),
    "sgd": ("own", {"NEXTOU_NARROW_COPY_SUM": "0", "NEXTOU_UPCONV_GEMM": "0"}),
    "sgd+ncs": ("own", {"NEXTOU_NARROW_COPY_SUM": "1", "NEXTOU_UPCONV_GEMM": "0"}),
    "all": ("own", {"NEXTOU_NARROW_COPY_SUM": "1", "NEXTOU_UPCONV_GEMM": "1"}),
    "no_opt": ("none", {"NEXTOU_NARROW_COPY_SUM": "1", "NEXTOU_UPCONV_GEMM": "1"}),
}


def set_env(name):
    os.environ.update(VARIANTS[name][1])


def make_variant(name, workload, device):
    kind = VARIANTS[name][0]
    trainer, cfg, batch, classes = bench.build_trainer(workload, device, False)
    bench.move_to(trainer, device, fused_sgd=True)
    if kind == "own":
        from nextou_amd.optim import ClipSGD
        trainer.optimizer = ClipSGD(trainer.network.parameters(), trainer.initial_lr, weight_decay=trainer.weight_decay,
                                    momentum=trainer.momentum, nesterov=True)
    else:       # round 4's bench step: torch's fused multi-tensor SGD
        trainer.optimizer = torch.optim.SGD(trainer.network.parameters(), trainer.initial_lr, weight_decay=trainer.weight_decay,
                                            momentum=trainer.momentum, nesterov=True, fused=True)
    data, target = synthetic_batch(cfg, 1, classes, batch, device, seed=1234, blob_labels=(workload == "cfg4"))
    targets = downsample_targets(target, bench._head_shapes(cfg))
    params = [p for p in trainer.network.parameters() if p.requires_grad]

    def step():
        trainer.optimizer.zero_grad(set_to_none=True)
        loss = trainer.loss(trainer.network(data), targets)
        loss.backward()
        if kind == "own":
            trainer.optimizer.clip_and_step(12)
        elif kind == "torch":
            torch.nn.utils.clip_grad_norm_(params, 12)
            trainer.optimizer.step()
        return loss
    return step, trainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--variants", default="base,sgd,sgd+ncs,all,no_opt")
    ap.add_argument("--eager", action="store_true", help="time the eager steps as well")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = True
    kinds = args.variants.split(",")
    graphs, eager = {}, {}
    for k in kinds:
        set_env(k)
        step, trainer = make_variant(k, args.workload, device)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        eager[k] = step
        graphs[k] = (GraphedTrainStep(step, warmup=1, network=trainer.network, loss=trainer.loss), trainer)
        for _ in range(2):
            graphs[k][0]()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        print("captured %s; %.1f GB reserved" % (k, torch.cuda.memory_reserved() / 2**30), flush=True)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    rows = {k: [] for k in kinds}
    erows = {k: [] for k in kinds}
    for _ in range(args.rounds):
        for k in kinds:
            rows[k].append(timed(graphs[k][0]))
        if args.eager:
            for k in kinds:
                set_env(k)
                erows[k].append(timed(eager[k]))
    print("| variant | replayed hipGraph, ms / step (%d rounds x %d steps, alternating) | mean |%s" % (
        args.rounds, args.steps, " eager ms / step | mean |" if args.eager else ""))
    print("|---|---|---:|%s" % ("---|---:|" if args.eager else ""))
    for k in kinds:
        line = "| %s | %s | %.3f |" % (k, " / ".join("%.3f" % v for v in rows[k]), sum(rows[k]) / len(rows[k]))
        if args.eager:
            line += " %s | %.3f |" % (" / ".join("%.3f" % v for v in erows[k]), sum(erows[k]) / len(erows[k]))
        print(line)


if __name__ == "__main__":
    main()
