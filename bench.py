#!/usr/bin/env python
"""bench.py — voxels/sec of the NexToU train step (fwd + loss + bwd [+ grad all-reduce] + SGD) on MI355X.

    python bench.py [--gpus N --steps K --warmup W]            # N = 1: plain process; N > 1 typed like this: bench.py starts its own
                                                               # N ranks (nextou_amd/launch.py) — same line as below, free port
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W  # one rank per GPU over RCCL (what the round driver types)

Workload (BASELINE.json configs[1], SURVEY.md §8d cfg 2): 3-D NexToU, patch 64x224x192, base 33 /
max 324 features, 6 stages, 14 classes, batch 2 per GPU, fp32, BatchNorm in train mode,
deep-supervision-weighted cross-entropy; synthetic N(0,1) volumes and random-init (He) weights.
One step = zero_grad -> forward (5 heads) -> loss -> backward -> (N > 1: bucketed RCCL gradient
average; default: two hipGraphs around the eager all-reduces, `--graph off | on` overlap them with backward)
-> clip_grad_norm_(12) -> SGD(nesterov).  `value` = all ranks'
voxels / max-over-ranks time of exactly K steps between barrier + synchronize pairs.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     — the dominant own HIP kernel of the timed region: algorithmic flops or bytes per launch
                 / its mean launch duration (HIP events on the launch stream, recorded inside
                 libnextou_hip.so), against the MI355X peak it is bound by;
  cpu_baseline — the oracle's PyTorch-CPU port of the reference op sequence (oracle/ref_ops.py) timed on
                 this box's host cores on a bounded sample (N = 1 only).
"""
from __future__ import annotations

import argparse
import copy
import ctypes
import json
import os
import sys
import time

# MIOpen's find step (torch.backends.cudnn.benchmark) also times its naive reference solvers, which
# need ~0.1-0.4 s PER CALL on the 64x224x192 stages (measured: 317 s of a 350 s warm-up, profiles/).
# They can never win, so they are excluded from the search; nothing else about MIOpen is changed.
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from nextou_amd import _lib, graph_ops  # noqa: E402
from nextou_amd.ddp import BucketedGradientAverager, init_process_group_from_env, init_single_process_group  # noqa: E402
from nextou_amd.launch import check_world, needs_self_launch, self_launch  # noqa: E402
from nextou_amd.harness import (GraphedTrainStep, SplitGraphedTrainStep, config_3d_fullres_nextou, deep_supervision_weights, downsample_targets,  # noqa: E402
                                synthetic_batch)
from nextou_amd.loss.nnunet_losses import DeepSupervisionWrapper, RobustCrossEntropyLoss  # noqa: E402
from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU  # noqa: E402
from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_BTI_Synapse import nnUNetTrainer_NexToU_BTI_Synapse  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA = f32 vector peak
MODEL_MFLOP_PER_VOXEL_STEP = 1.697  # 565.6 kFLOP / voxel forward (FlopCounterMode on the reference, cfg 2) x 3 for fwd + bwd

WORKLOADS = {
    # name: (patch, base, max_features, batch/GPU, classes)
    "cfg2": ((64, 224, 192), 33, 324, 2, 14),
    "cfg4": ((64, 224, 192), 33, 324, 2, 14),      # cfg2 + Dice + CE + BTI loss, blob labels
    "cfg5": ((96, 256, 256), 33, 324, 2, 14),      # BASELINE.json configs[4] shape (use with --autocast-bf16)
    "tiny": ((32, 128, 128), 6, 48, 2, 14),        # plumbing check only — never a reported number
}


class _CETrainer(nnUNetTrainer_NexToU):
    """cfg 2: deep-supervision-weighted cross-entropy."""

    def _build_loss(self):
        return DeepSupervisionWrapper(RobustCrossEntropyLoss(),
                                      deep_supervision_weights(len(self._get_deep_supervision_scales())))


def build_trainer(workload, device, is_ddp, seed=0):
    patch, base, max_f, batch, classes = WORKLOADS[workload]
    cfg = config_3d_fullres_nextou(patch_size=patch, base=base, max_features=max_f, batch_size=batch)
    cls = nnUNetTrainer_NexToU_BTI_Synapse if workload == "cfg4" else _CETrainer
    torch.manual_seed(seed)
    trainer = cls(cfg, classes, num_input_channels=1, device=torch.device("cpu"), is_ddp=is_ddp, log=None)
    trainer.initialize()              # built on the host (position tables), moved below
    trainer.device = device
    return trainer, cfg, batch, classes


def move_to(trainer, device, fused_sgd=False):
    trainer.network.to(device)
    trainer.device = device
    # fused_sgd: torch's single-launch multi-tensor SGD (same update rule; -0.5 ms per cfg-2 step).  Only without the gradient averager:
    # the averaged step's gradients are views into the flat buckets and keeps the foreach implementation.  (The GPU memory fault round 4
    # first blamed on this combination is MIOpen's backward-data kernel of the 1x1 head convolution reading past its operand —
    # tools/conv_bwd_fault_repro.py, profiles/r05_n_gt_1.md; the heads run on K8 since.)
    fused = fused_sgd and device.type == "cuda" and os.environ.get("NEXTOU_SGD_FUSED", "1") != "0"
    if device.type == "cuda" and os.environ.get("NEXTOU_CLIP_SGD", "1") != "0":
        # round 5: the clip and the update on the library's step-glue kernels (nextou_amd/optim.py: a torch.optim.SGD whose step names
        # its tensors through a table in device memory — 3 launches, 144 us replayed against torch's 575).  Under the gradient averager the gradients are
        # views of the flat buckets; a view whose element order differs from its channels-last filter is copied into the filter's order first
        # (optim.ClipSGD._plan), the step itself stays on the own kernels
        from nextou_amd.optim import ClipSGD
        trainer.optimizer = ClipSGD(trainer.network.parameters(), trainer.initial_lr, weight_decay=trainer.weight_decay,
                                    momentum=trainer.momentum, nesterov=True)
    else:
        trainer.optimizer = torch.optim.SGD(trainer.network.parameters(), trainer.initial_lr, weight_decay=trainer.weight_decay,
                                            momentum=trainer.momentum, nesterov=True, **({"fused": True} if fused else {}))
    if hasattr(trainer.loss, "loss") and hasattr(trainer.loss.loss, "ti"):
        trainer.loss = trainer._build_loss()      # interaction tensors follow the device


class TrainStep:
    """One training step in three parts — ``part1`` (zero_grad -> forward -> loss -> backward [+ the buckets' stragglers filled]), ``between``
    (the bucket all-reduces still to launch, and the join) and ``part2`` (1 / world scale -> clip_grad_norm_(12) -> SGD) — so that it can run
    eagerly (``step()``), as one hipGraph (harness.GraphedTrainStep(step)) or as two graphs around eager collectives
    (harness.SplitGraphedTrainStep(step.part1, step.between, step.part2))."""

    def __init__(self, trainer, data, targets, averager, bf16=False):
        self.trainer, self.data, self.targets, self.averager, self.bf16 = trainer, data, targets, averager, bf16
        self.params = [p for p in trainer.network.parameters() if p.requires_grad]

    def part1(self):
        tr = self.trainer
        if self.averager is not None:
            self.averager.zero_grad()                 # same effect, keeps the flat buckets
        tr.optimizer.zero_grad(set_to_none=True)
        if self.bf16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                outs = tr.network(self.data)
            loss = tr.loss([o.float() for o in outs], self.targets)
        else:
            loss = tr.loss(tr.network(self.data), self.targets)
        loss.backward()
        if self.averager is not None:
            self.averager.fill_missing()
        return loss

    def between(self):
        if self.averager is not None:
            self.averager.reduce_all()
            self.averager.wait_all()

    def part2(self):
        tr = self.trainer
        if self.averager is not None:
            self.averager.finish_local()
        if hasattr(tr.optimizer, "clip_and_step"):
            tr.optimizer.clip_and_step(12)            # clip_grad_norm_(params, 12) + SGD step on the step-glue kernels
        else:
            torch.nn.utils.clip_grad_norm_(self.params, 12)
            tr.optimizer.step()

    def __call__(self):
        loss = self.part1()
        self.between()
        self.part2()
        return loss


def make_step(trainer, data, targets, averager, bf16=False):
    return TrainStep(trainer, data, targets, averager, bf16)


def profile_report():
    buf = ctypes.create_string_buffer(1 << 20)
    n = _lib.lib().nextou_profile_report(buf, len(buf))
    return json.loads(buf.value.decode()) if n else []


_PMC_TRAFFIC = None


def _pmc_traffic(label):
    """HBM bytes per launch of this kernel label from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), or None."""
    global _PMC_TRAFFIC
    if _PMC_TRAFFIC is None:
        tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
        _PMC_TRAFFIC = json.load(open(tpath)) if os.path.exists(tpath) else {}
    return _PMC_TRAFFIC.get(label)


def _roofline_entry(top):
    per_launch_work = top["work"] / top["launches"]
    per_launch_s = top["ms"] / top["launches"] / 1e3
    if top["bound"] == "mfma":
        achieved, peak, unit = per_launch_work / per_launch_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
    else:
        achieved, peak, unit = per_launch_work / per_launch_s / 1e9, HBM_PEAK_GBS, "GB/s"
    return {"kernel": top["kernel"], "bound": top["bound"], "achieved": round(achieved, 3), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "launches": top["launches"], "avg_us": round(per_launch_s * 1e6, 2),
            # the committed PMC FETCH / WRITE bytes per launch of this label beside the algorithmic work (null: no counter run)
            "traffic": _pmc_traffic(top["kernel"]), "algorithmic_work_per_launch": round(per_launch_work, 1)}


def roofline_graph_from(report):
    """The graph kernels north_star names, beside the dominant-kernel object: the K1 (kNN, fp32 MFMA bound) and K2
    (max-relative aggregation, HBM bound) launch shapes with the largest summed time in the timed region."""
    def pick(prefixes):
        rows = [r for r in report if r["kernel"].startswith(prefixes)]
        return _roofline_entry(max(rows, key=lambda r: r["ms"])) if rows else None
    def worst(prefixes):
        """the launch shape of these kernels that sits furthest below its roofline (the small stage-4 / 5 calls)"""
        rows = [_roofline_entry(r) for r in report if r["kernel"].startswith(prefixes)]
        return min(rows, key=lambda e: e["frac"]) if rows else None
    return {"K1_knn": pick(("knn_fused_kernel", "knn_window_kernel", "knn_small_kernel")), "K2_mr_forward": pick(("mr_fwd",)), "K2_mr_backward": pick(("mr_bwd",)),
            "K1_knn_worst_shape": worst(("knn_fused_kernel", "knn_window_kernel", "knn_small_kernel")), "K2_mr_forward_worst_shape": worst(("mr_fwd",)),
            "K2K7_mr_grouped_forward": pick(("mr_grp_rows_kernel",)), "K2K7_mr_grouped_backward": pick(("mr_grp_rows_bwd_kernel",)),
            "K5_argmax_labels": pick(("argmax_labels_kernel",)), "K5_bti_critical": pick(("bti_critical_kernel",)),
            "K5_bti_ce_forward": pick(("bti_ce_fwd_kernel",)), "K5_bti_ce_backward": pick(("bti_ce_bwd_kernel",)),
            "K5_ce_mean_forward": pick(("ce_mean_fwd_kernel",)), "K5_ce_mean_backward": pick(("ce_mean_bwd_kernel",)),
            "K5_dice_stats_forward": pick(("dice_stats_fwd_kernel",)), "K5_dice_stats_backward": pick(("dice_stats_bwd_kernel",)),
            "K7_pointwise_rows": pick(("pw_rows_kernel", "pw_rows_sw_kernel")),
            "K7_pointwise_wgrad": pick(("pw_wgrad_kernel", "pw_wgrad_so_kernel")),
            "K7_pointwise_rows_worst_shape": worst(("pw_rows_kernel", "pw_rows_sw_kernel")),
            "K8_head_forward": pick(("head_fwd_kernel", "head_fwd_lds_kernel")),
            "K8_head_backward": pick(("head_bwd_lds_kernel", "head_dgrad_kernel", "head_wgrad_kernel")),
            "K2K7_pool_grouped_forward": pick(("mr_grp_cm_kernel",)),
            # round 6 (ABI v14): the stem block — first conv -> norm -> act of the network without the convolution's output in memory
            "K9_stem_forward": pick(("stem_apply_kernel",)), "K9_stem_backward": pick(("stem_bwd_kernel",)),
            "K9_stem_moments": pick(("stem_moments_kernel",)),
            # round 5's step glue (ABI v13): the gradient norm + clip / SGD update over the whole parameter list, the decoder concatenation with the
            # up-convolution's pixel shuffle (forward) and its one-pass backward (channel range copied / un-shuffled + bias sums)
            "glue_grad_norm": pick(("multi_sumsq_kernel",)), "glue_clip_sgd": pick(("clip_sgd_kernel",)),
            "glue_upconv_cat_forward": pick(("upconv_cat_rows_kernel", "cat_bias_rows_kernel", "cat_skip_half_kernel")),
            "K7_upconv_store_in_place": pick(("pw_rows_kernel<2,11|up>", "pw_rows_kernel<2,9|up>", "pw_rows_kernel<2,7|up>", "pw_rows_kernel<2,6|up>")),
            "glue_cat_backward": pick(("narrow_copy_stats_kernel",)),
            "graph_kernels_ms_per_step": None}


def roofline_from(report):
    """Dominant own kernel (largest summed time in the timed region)."""
    if not report:
        return None
    top = max(report, key=lambda r: r["ms"])
    per_launch_work = top["work"] / top["launches"]
    per_launch_s = top["ms"] / top["launches"] / 1e3
    if top["bound"] == "mfma":
        achieved, peak, unit = per_launch_work / per_launch_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
    else:
        achieved, peak, unit = per_launch_work / per_launch_s / 1e9, HBM_PEAK_GBS, "GB/s"
    traffic = None
    if os.path.exists(os.path.join(REPO, "profiles", "pmc_traffic.json")):   # HBM bytes per launch from a committed rocprofv3 --pmc run
        traffic = _pmc_traffic(top["kernel"])
        if traffic is None:     # never silent: the committed counter table has no row for today's dominant label
            print("bench.py: profiles/pmc_traffic.json has no entry for the dominant kernel label %r: roofline.traffic = null "
                  "(regenerate with tools/pmc_traffic.sh)" % top["kernel"], file=sys.stderr)
    real = _real_channel_fraction(top["kernel"])
    return {"bound": top["bound"], "achieved": round(achieved, 3), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "frac_on_unpadded_bytes": None if real is None else round(real * achieved / peak, 4),
            "frac_on_unpadded_bytes_note": "the plain stages carry 33 -> 40 / 66 -> 72 channels inside the network (channel_pad.py); "
                                           "`achieved` counts the bytes the kernel really moves, this field only the real channels' share",
            "traffic_source": "not measured in this run: looked up in profiles/pmc_traffic.json, the committed rocprofv3 "
                              "--pmc FETCH_SIZE / WRITE_SIZE passes (separate, gfx950-corrected) of this kernel label; "
                              "null when the label has no committed counter run",
            "kernel": top["kernel"], "launches": top["launches"], "avg_us": round(per_launch_s * 1e6, 2),
            "own_kernels_ms_per_step": None}


def roofline_step(workload, voxels_per_step, seconds_per_step, world, bf16):
    """End to end: the MODEL's algorithmic FLOPs per step (1.697 MFLOP per input voxel for forward + backward of the
    cfg-2 topology — the loss, the clip and the optimiser add nothing measurable) over the timed step, as a fraction of the fp32 MFMA
    peak of the GPUs used.  The dense convolution stages hold 98 % of these FLOPs and run on MIOpen / CK (north_star), so this is
    their efficiency more than the own kernels'; null for the informational bf16 runs (another peak applies)."""
    if bf16 or workload not in ("cfg2", "cfg4", "cfg5"):
        return None
    flop = MODEL_MFLOP_PER_VOXEL_STEP * 1e6 * voxels_per_step
    achieved = flop / seconds_per_step / 1e12
    return {"bound": "mfma", "flop_per_step": flop, "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS * world, "unit": "TFLOP/s",
            "frac": round(achieved / (MFMA_F32_PEAK_TFLOPS * world), 4)}


def _real_channel_fraction(label):
    """C_real / C_padded when the launch label is a K6 call on an internally padded plain-stage tensor (C = 40 or 72 at cfg 2)."""
    import re
    m = re.search(r"\[B\d+ C(\d+) S\d+\]", label)
    if not m or not label.startswith(("bn_", "channel_sum")):
        return None
    return {40: 33.0 / 40.0, 72: 66.0 / 72.0}.get(int(m.group(1)))


def parity_record(workload):
    """The metric's second half ("max logit abs-diff vs ref", BASELINE.json): NOT measured in this run — the margins a GPU run of the parity
    tests printed (tools/measure.sh margins -> profiles/rNN_parity_margins.txt -> tools/parity_json.py -> profiles/parity_margins.json), for
    the headline configuration; null for the other workloads."""
    path = os.path.join(REPO, "profiles", "parity_margins.json")
    if workload not in ("cfg2", "cfg4") or not os.path.exists(path):
        return None
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


CPU_BASELINE_THREADS = 32       # best of the committed sweep on the GPU boxes' EPYC 9575F hosts (profiles/r03_cpu_baseline_thread_sweep.md)


def cpu_baseline(workload, timed_steps=3, threads=None, sweep=False, batch=1):
    """oracle/ref_ops.py (the reference's op sequence, PyTorch-CPU fp32) on this host's cores: train steps of the same
    network at batch 1 — 1 warm-up + ``timed_steps`` timed, median reported.

    Departure from SURVEY.md §8(d) ("batch 2, 1 warm-up + >= 3 timed", all physical cores), on purpose: that protocol is ~10 min of
    host time per bench run, against this file's contract of a default run that finishes within minutes.  Here: 1 warm-up + 3
    timed steps (round 6: three, as §8(d) asks) at batch 1 on the thread count the committed sweep found fastest — ~25 s per
    step.  Batch 1 is a per-voxel-equivalent sample (samples are independent on the CPU path
    apart from batch-norm statistics); the warm-up removes oneDNN primitive creation and first-touch allocation, which
    was the un-warmed round-1 number's noise."""
    from oracle.ref_ops import TorchRefBackend   # checker / baseline only — never the product path
    import oracle  # noqa: F401
    patch, base, max_f, _, classes = WORKLOADS[workload]
    default_threads = torch.get_num_threads()
    graph_ops.install_cpu_checker(TorchRefBackend)
    try:
        trainer, cfg, _, _ = build_trainer(workload, torch.device("cpu"), False)
        data, target = synthetic_batch(cfg, 1, classes, batch, torch.device("cpu"), blob_labels=(workload == "cfg4"))
        with torch.no_grad():
            shapes = [tuple(o.shape[2:]) for o in _head_shapes(cfg)]
        targets = [target if s == tuple(target.shape[2:]) else
                   torch.nn.functional.interpolate(target, size=s, mode="nearest") for s in shapes]
        step = make_step(trainer, data, targets, None)

        def timed():
            t0 = time.perf_counter()
            step()
            return time.perf_counter() - t0

        # 128 threads on a 256-logical-core host lost to the reference on 8 cores.  `--cpu-thread-sweep`
        # times ONE step per thread count after the warm-up (~6 extra steps, ~5 min: not the default run); the default run
        # uses the thread count that sweep found best on this host type (CPU_BASELINE_THREADS) and says so in `cores`.
        warm = timed()
        physical = max(1, (os.cpu_count() or 2) // 2)
        chosen = min(threads or CPU_BASELINE_THREADS or default_threads, max(physical, 1))
        sweep_times = {}
        if sweep:
            for n in sorted({min(n, physical) for n in (8, 16, 32, 64, 128)} | {chosen}):
                torch.set_num_threads(n)
                sweep_times[n] = timed()
            chosen = min(sweep_times, key=sweep_times.get)
        torch.set_num_threads(chosen)
        times = ([sweep_times[chosen]] if sweep else []) + [timed() for _ in range(max(timed_steps - (1 if sweep else 0), 1))]
        best = chosen
    finally:
        graph_ops.install_cpu_checker(None)
        torch.set_num_threads(default_threads)
    voxels = batch * int(np.prod(patch))
    timed_sorted = sorted(times)
    n = len(timed_sorted)
    dt = timed_sorted[n // 2] if n % 2 else 0.5 * (timed_sorted[n // 2 - 1] + timed_sorted[n // 2])
    return {"value": round(voxels / dt, 1), "unit": "voxels/s", "cores": best, "kind": "port",
            "sample": "train steps (fwd+loss+bwd+SGD) at batch %d of the %s patch, fp32: 1 warm-up (%.1f s) + %d timed on %d threads "
                      "(%s s), median %.1f s%s"
                      % (batch, "x".join(map(str, patch)), warm, len(times), best, ", ".join("%.1f" % t for t in times), dt,
                         "; thread sweep, one step each: %s" % {k: round(v, 1) for k, v in sweep_times.items()} if sweep else ""),
            "thread_sweep_s_per_step": {str(k): round(v, 2) for k, v in sweep_times.items()} or None,
            "cpu": _cpu_model()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " (%d logical)" % os.cpu_count()
    except OSError:
        pass
    return "unknown"


class _Shape:
    def __init__(self, shape):
        self.shape = shape


def _head_shapes(cfg):
    """(B, L, *spatial) of the deep-supervision heads, highest resolution first (all but the
    bottleneck resolution)."""
    shape = list(cfg.patch_size)
    out = []
    for pool in cfg.pool_op_kernel_sizes[:-1]:
        shape = [s // p for s, p in zip(shape, pool)]
        out.append(_Shape((1, 1, *shape)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps after one warm-up (batch 1)")
    ap.add_argument("--cpu-threads", type=int, default=None, help="threads of the CPU-baseline leg (default: the committed sweep's best)")
    ap.add_argument("--cpu-thread-sweep", action="store_true",
                    help="CPU-baseline leg: after the warm-up, time one step at 8/16/32/64/128 threads and report the best (~5 min extra)")
    ap.add_argument("--cpu-protocol", choices=("bounded", "survey"), default="bounded",
                    help="CPU-baseline leg: 'bounded' (default) = batch 1, 1 warm-up + --cpu-steps timed on the sweep's best thread count "
                         "(~1.5 min); 'survey' = SURVEY.md §8(d) to the letter: batch 2, 1 warm-up + 3 timed, all physical cores (~10 min)")
    ap.add_argument("--no-miopen-find", action="store_true", help="disable MIOpen's find/benchmark mode")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--force-averager", action="store_true",
                    help="N = 1 only: run the bucketed gradient averager on a world-size-1 group, to time the hooks + "
                         "bucket copies + collective launches of the N > 1 path on one GPU")
    ap.add_argument("--autocast-bf16", action="store_true",
                    help="informational (cfg-5 regime): conv stages under bf16 autocast, graph ops stay fp32; "
                         "never the headline number")
    ap.add_argument("--graph", choices=("auto", "on", "off", "split"), default="auto",
                    help="how the step is replayed.  auto: N = 1 -> ONE captured hipGraph (harness.GraphedTrainStep; cfg 4: the BTI target "
                         "validation runs on the device, checked after timing); with a process group up (N > 1, --force-averager) -> `split`: two "
                         "hipGraphs around EAGER bucket all-reduces (harness.SplitGraphedTrainStep) — no collective is captured, because capturing "
                         "RCCL collectives can end in an uncatchable abort inside PyTorch's watchdog thread (DESIGN.md 6).  auto falls back to the "
                         "eager step if a capture fails (error reported in the JSON line).  on: one graph, collectives included (raises if the "
                         "capture fails); off: the eager step")
    ap.add_argument("--channels-last", action="store_true",
                    help="experiment: run the dense stages in channels_last_3d (NDHWC) memory format")
    args = ap.parse_args()

    # `python bench.py --gpus N` typed without a launcher (the round driver's command shape, VERDICT r5 missing #2): start the N ranks here
    if needs_self_launch(args.gpus):
        raise SystemExit(self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    check_world(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    # "rank 0 prints ONE JSON line": native libraries write to file descriptor 1 as well (RCCL prints a version banner when its
    # communicator comes up — it followed the JSON line in the world-size-1 runs of profiles/r03_averaged_step_hipgraph.md), so
    # descriptor 1 is pointed at stderr for the whole run and the JSON line goes to a private duplicate of the real stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    _lib.lib()  # fail loudly if the HIP extension is missing
    # RCCL ("nccl") is the product backend; NEXTOU_DIST_BACKEND=gloo lets the N > 1 code path be exercised
    # by two ranks sharing the single GPU of a test box (tests/test_gpu_parity.py)
    backend = os.environ.get("NEXTOU_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl" and torch.cuda.device_count() > local:
        torch.cuda.set_device(local)
    rank, local_rank, world = init_process_group_from_env(backend)
    assert world == args.gpus
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = not args.no_miopen_find

    trainer, cfg, batch, classes = build_trainer(args.workload, device, world > 1)
    cpu_copy_ok = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    move_to(trainer, device, fused_sgd=(world == 1 and not args.force_averager) or os.environ.get("NEXTOU_SGD_FUSED") == "force")
    if args.channels_last:
        trainer.network.to(memory_format=torch.channels_last_3d)
    if args.force_averager and world == 1:
        init_single_process_group(backend)
    averager = BucketedGradientAverager(trainer.network, bucket_bytes=args.bucket_mb << 20) \
        if (world > 1 or args.force_averager) else None
    data, target = synthetic_batch(cfg, 1, classes, batch, device, seed=1234 + rank,
                                   blob_labels=(args.workload == "cfg4"))
    targets = downsample_targets(target, _head_shapes(cfg))
    if args.channels_last:
        data = data.contiguous(memory_format=torch.channels_last_3d)
    step = make_step(trainer, data, targets, averager, bf16=args.autocast_bf16)

    for _ in range(args.warmup):
        step()
    # auto: the whole step as one hipGraph, N = 1 and N > 1 alike (the averaged step is capturable: RCCL collectives on their
    # own stream, no host synchronisation in the hooks or in finalize() once the warm-up steps have seen the gradient pattern)
    # (gloo — the two-ranks-on-one-GPU test backend — synchronises with the host inside its collectives and cannot be captured)
    # round 6: with a process group up the default keeps the collectives OUT of any capture — two graphs
    # around eager all-reduces; the single graph with captured collectives stays available under `--graph on`
    split = args.graph == "split" or (args.graph == "auto" and averager is not None)
    want_graph = args.graph in ("on", "split") or args.graph == "auto"
    graph_mode = ("one graph, collectives captured (--graph on)" if args.graph == "on" else "eager (--graph off)" if args.graph == "off" else
                  "one graph (auto: no process group)" if not split else
                  "two graphs around eager collectives (%s)" % ("--graph split" if args.graph == "split" else
                                                                "auto: a process group is up; RCCL collectives are captured only on --graph on"))
    graphed, capture_error = None, None
    if want_graph:
        try:
            if split and averager is not None:
                averager.defer_collectives = True
                graphed = SplitGraphedTrainStep(step.part1, step.between, step.part2, warmup=1, network=trainer.network, loss=trainer.loss)
            else:
                graphed = GraphedTrainStep(step, warmup=1, network=trainer.network, loss=trainer.loss)
            for _ in range(2):
                graphed()
        except Exception as exc:
            # never silent: `--graph on` fails; `auto` falls back to the eager step — the same computation —
            # and the JSON line says so (config.step_replayed_as_hipgraph = false, config.graph_capture_error)
            if args.graph in ("on", "split"):
                raise
            if averager is not None:
                averager.defer_collectives = False        # the eager step overlaps the collectives with backward again
            capture_error = "%s: %s" % (type(exc).__name__, exc)
            print("bench.py: hipGraph capture failed (%s); timing the eager step" % capture_error, file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()
    # the launch profiler's record pool is sized from a COUNTED step (a fixed 1 536 records dropped the third
    # step's backward tail): one eager step with a generous pool tells how many launches of this library a step makes
    _lib.lib().nextou_profile_enable(1 << 15)
    step()
    torch.cuda.synchronize()
    launches_per_step = sum(r["launches"] for r in profile_report())
    _lib.lib().nextou_profile_enable(0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    run = graphed if graphed is not None else step
    if graphed is None:
        _lib.lib().nextou_profile_enable(launches_per_step * max(args.steps, 1) + 64)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if graphed is not None:
        # per-kernel HIP-event timing (the roofline objects) cannot live inside a captured graph: three eager steps of the
        # same computation after the timed region provide it
        _lib.lib().nextou_profile_enable(launches_per_step * 3 + 64)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    report = profile_report()
    dropped = _lib.lib().nextou_profile_dropped()
    _lib.lib().nextou_profile_enable(0)
    profiled_steps = 3 if graphed is not None else args.steps
    uneven = [r["kernel"] for r in report if r["launches"] % profiled_steps]
    profile_check = {"launches_per_step": launches_per_step, "profiled_steps": profiled_steps, "dropped_records": dropped,
                     "labels_not_a_multiple_of_the_steps": uneven}
    if dropped or uneven:       # never silent, never fatal to the timed number: the per-step sums below would be off
        print("bench.py: launch profile inconsistent: %s" % profile_check, file=sys.stderr)
    if graphed is not None:
        graphed.check()                     # deferred (B)TI target validation of the captured step
    if averager is not None:
        averager.check_consistency()        # every rank produced gradients for the same parameters on every step
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # what the process group really was: a SCALE record shows that RCCL saw N ranks on N distinct devices
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "device": str(device), "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
          "uuid": str(getattr(props, "uuid", "")) or None}
    if dist.is_initialized():
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)
        dist_info = {"initialized": True, "backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks,
                     "distinct_devices": len({(r["pci_bus_id"], r["uuid"], r["device"]) for r in ranks})}
    else:
        dist_info = {"initialized": False, "backend": None, "world_size": 1, "ranks": [me], "distinct_devices": 1}

    if rank == 0:
        voxels_per_step = world * batch * int(np.prod(cfg.patch_size))
        ms = elapsed / args.steps * 1e3
        roof = roofline_from(report)
        graph = roofline_graph_from(report) if report else None
        if roof is not None:
            roof["own_kernels_ms_per_step"] = round(sum(r["ms"] for r in report) / profiled_steps, 3)
            graph["graph_kernels_ms_per_step"] = round(sum(r["ms"] for r in report if r["kernel"].startswith(
                ("knn_", "mr_", "window_", "pool_rows", "cell_"))) / profiled_steps, 3)
        line = {
            "metric": "voxels/sec fwd+bwd, 3D %s patch batch=%d" % ("x".join(map(str, cfg.patch_size)), batch),
            "value": round(voxels_per_step / (elapsed / args.steps), 1),
            "unit": "voxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16-autocast(conv)/f32(graph)" if args.autocast_bf16 else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: 3D NexToU %s, base %d / max %d features, batch %d per GPU, "
                                   "%d classes, fp32, train-mode BN, %s; step = fwd+loss+bwd%s+clip+SGD"
                                   % ({"cfg4": 3, "cfg5": 4}.get(args.workload, 1), "x".join(map(str, cfg.patch_size)),
                                      cfg.UNet_base_num_features,
                                      cfg.unet_max_num_features, batch, classes,
                                      "Dice+CE+BTI(Synapse) loss" if args.workload == "cfg4" else "deep-supervision CE loss",
                                      "+RCCL grad all-reduce" if world > 1 else ""),
                       "name": args.workload, "global_batch": world * batch,
                       "layout": "channels-last stages %s" % sorted(trainer.network.encoder.channels_last_stages),
                       "internal_channel_padding_modules": getattr(trainer.network, "padded_modules", 0),
                       "gradient_averager": averager is not None,
                       "optimizer": "SGD(nesterov, momentum %g, weight_decay %g, %s)" % (
                           trainer.momentum, trainer.weight_decay,
                           {"own": "own clip + update kernels (nextou_amd.optim.ClipSGD)",
                            "torch": "ClipSGD -> torch foreach: %s" % getattr(trainer.optimizer, "last_reason", None)}.get(
                               trainer.optimizer.last_path, "not stepped") if hasattr(trainer.optimizer, "last_path")
                           else ("fused" if trainer.optimizer.defaults.get("fused") else "foreach")),
                       "step_replayed_as_hipgraph": graphed is not None, "graph_capture_error": capture_error,
                       "graph_mode": graph_mode if graphed is not None or not want_graph else "eager (the capture failed)",
                       "parallelism": "dp%d" % world, "final_loss": float(loss.detach())},
            "dist": dist_info,
            "parity": parity_record(args.workload),
            "roofline": roof,
            "roofline_graph": graph,
            "launch_profile_check": profile_check,
            "roofline_step": roofline_step(args.workload, voxels_per_step, elapsed / args.steps, world, args.autocast_bf16),
        }
        if cpu_copy_ok:
            try:
                if args.cpu_protocol == "survey":
                    line["cpu_baseline"] = cpu_baseline(args.workload, 3, max(1, (os.cpu_count() or 2) // 2), False, batch=batch)
                    line["cpu_baseline"]["protocol"] = "SURVEY.md 8(d): batch 2, 1 warm-up + 3 timed, all physical cores"
                else:
                    line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_steps, args.cpu_threads, args.cpu_thread_sweep)
            except Exception as e:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "voxels/s", "cores": torch.get_num_threads(),
                                        "kind": "port", "sample": "failed: %r" % (e,)}
        elif world == 1:
            line["cpu_baseline"] = None
        if world == 1 and args.workload == "cfg2":
            # both CPU numbers in the line: `cpu_baseline` is measured in THIS run on a bounded sample (batch 1, the
            # sweep's best thread count — the faster of the two protocols, i.e. the fairer yardstick); `cpu_baseline_survey` is SURVEY.md
            # 8(d) to the letter (batch 2, 1 warm-up + 3 timed, all physical cores), ~10 min of host time and therefore a COMMITTED run
            # (`bench.py --cpu-protocol survey` reproduces it), not re-measured here
            spath = next((p for p in (os.path.join(REPO, "profiles", n) for n in ("r06_bench_cfg2_cpu_survey.json", "r04_bench_cfg2_cpu_survey.json"))
                          if os.path.exists(p)), None)
            if spath is not None:
                try:
                    sv = json.load(open(spath))["cpu_baseline"]
                    line["cpu_baseline_survey"] = dict(sv, source="committed run profiles/%s (not measured in this run)" % os.path.basename(spath),
                                                       batch=2, timed_steps=3)
                except (OSError, ValueError, KeyError):
                    line["cpu_baseline_survey"] = None
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
