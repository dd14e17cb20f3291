#!/usr/bin/env python
"""Per-kernel roofline of libnextou_hip.so at the cfg-2 (or cfg-5) call shapes of SURVEY.md §A.2.

    python tools/kernel_bench.py [--cfg 2|5] [--iters 10] [--json out.json]

Every launch is timed with HIP events on its own stream by the library's launch profiler
(nextou_profile_enable / nextou_profile_report); `achieved` = algorithmic flops or bytes per launch
/ mean launch time, `frac` = achieved / MI355X peak (fp32 MFMA 157.3 TFLOP/s, HBM 8000 GB/s).
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextou_amd import _lib, graph_ops  # noqa: E402

# (label, B', C, N, M (None = self), k)   — cfg 2, batch 2
CFG2 = [("s2 Pool", 2, 132, 10752, 168, 14), ("s2 Swin", 1024, 132, 168, None, 7),
        ("s3 Pool", 2, 264, 10752, 1344, 28), ("s3 Swin", 128, 264, 168, None, 14),
        ("s4 Pool", 2, 324, 1344, None, 32), ("s4 Swin", 16, 324, 168, None, 14),
        ("s5 Pool", 2, 324, 168, None, 32), ("s5 Swin", 2, 324, 168, None, 28)]
# cfg 5: patch 96x256x256, min shape (6,8,8) = 384 points per window
CFG5 = [("s2 Pool", 2, 132, 24576, 384, 32), ("s2 Swin", 1024, 132, 384, None, 16),
        ("s3 Pool", 2, 264, 24576, 3072, 32), ("s3 Swin", 128, 264, 384, None, 32),
        ("s4 Pool", 2, 324, 3072, None, 32), ("s4 Swin", 16, 324, 384, None, 32)]
# (label, B, C, S, instance norm?)  — the (norm -> LeakyReLU) calls of one cfg-2 step, largest first
NORM2 = [("s0 padded conv BN+act", 2, 40, 64 * 224 * 192, False), ("s1 padded conv BN+act", 2, 72, 64 * 112 * 96, False),
         ("s0 conv BN+act", 2, 33, 64 * 224 * 192, False), ("s1 conv BN+act", 2, 66, 64 * 112 * 96, False),
         ("s2 conv BN+act", 2, 132, 32 * 56 * 48, False), ("s2 FFN hidden", 2, 528, 32 * 56 * 48, False),
         ("s2 Swin fc BN", 1024, 132, 168, False), ("s2 Swin graph BN", 1024, 264, 168, False),
         ("s2 Pool graph IN", 2, 264, 10752, True), ("s3 conv BN+act", 2, 264, 16 * 28 * 24, False),
         ("s4 conv BN+act", 2, 324, 8 * 14 * 12, False)]
PEAK = {"mfma": 157.3e12, "hbm": 8000e9}


def bench_norm(args, dev, L):
    """K6 at the cfg-2 shapes, next to PyTorch-ROCm's batch_norm + leaky_relu (MIOpen) on the same tensors."""
    F = torch.nn.functional
    rows = []
    print("%-18s %-52s %10s %12s %7s" % ("call", "kernel", "us/launch", "achieved", "frac"))
    for label, B, C, S, inst in NORM2:
        if args.only and args.only not in label:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn((B, C, S), generator=g, device=dev)
        gy = torch.randn((B, C, S), generator=g, device=dev)
        if args.cl:
            if inst:
                continue
            x = x.view(B, C, S, 1).contiguous(memory_format=torch.channels_last)
            gy = gy.view(B, C, S, 1).contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        w = torch.rand((C,), generator=g, device=dev) + 0.5
        b = torch.randn((C,), generator=g, device=dev) * 0.1
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)

        def ours():
            y = graph_ops.norm_act(x, w, b, None if inst else rm, None if inst else rv, True, 0.1, 1e-5, 0.01, instance=inst)
            torch.autograd.grad(y, x, gy)

        def stock():
            z = F.instance_norm(x, None, None, w, b, True, 0.1, 1e-5) if inst else F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
            torch.autograd.grad(F.leaky_relu(z, 0.01), x, gy)

        for it in range(2 + args.iters):
            if it == 2:
                torch.cuda.synchronize()
                L.nextou_profile_enable(8 * args.iters)
            ours()
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 20)
        L.nextou_profile_report(buf, len(buf))
        L.nextou_profile_enable(0)
        own_us = 0.0
        for r in json.loads(buf.value.decode()):
            per_s = r["ms"] / r["launches"] / 1e3
            ach = r["work"] / r["launches"] / per_s
            own_us += per_s * 1e6
            rows.append({"call": label, "kernel": r["kernel"], "bound": r["bound"], "us": per_s * 1e6,
                         "achieved": ach, "frac": ach / PEAK[r["bound"]]})
            print("%-18s %-52s %10.1f %8.0f GB/s %6.1f%%" % (label, r["kernel"][:52], per_s * 1e6, ach / 1e9,
                                                           100 * ach / PEAK["hbm"]))
        def wall(fn):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / args.iters

        # own_us sums the profiled (ProfScope) kernels only; own_wall also contains the per-channel finalize launches,
        # the workspace allocations and the autograd glue — the number to compare against the stock path's wall time
        own_wall, stock_us = wall(ours), wall(stock)
        print("%-18s own fwd+bwd %.1f us (kernels) / %.1f us (wall)   PyTorch-ROCm batch_norm+leaky_relu fwd+bwd %.1f us "
              "(wall)   (%.2fx)" % (label, own_us, own_wall, stock_us, stock_us / own_wall))
        rows.append({"call": label, "own_us": own_us, "own_wall_us": own_wall, "stock_us": stock_us})
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


def bench_bti(args, dev, L):
    """K5 at the cfg-4 shapes (BASELINE.json configs[3]): the four deep-supervision scales the BTI loss runs on (batch 2,
    14 classes, Synapse exclusion list, connectivity 26), blob labels and blob-ish logits (the arg-max of the logits is a
    noisy copy of the labels, so ~5-10 % of the voxels are critical, as in the cfg-4 train step).  Per kernel: HIP-event
    time inside the library, algorithmic bytes (logits 4L per voxel for the arg-max and again for the
    CE, labels / critical map 1 byte each) over it, fraction of the 8 TB/s HBM peak."""
    from nextou_amd.harness import config_3d_fullres_nextou, synthetic_batch
    from nextou_amd.loss.bti_loss import BTI_Loss
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_BTI_Synapse import nnUNetTrainer_NexToU_BTI_Synapse as Synapse
    loss = BTI_Loss(3, 26, [], Synapse.exclusion_list, 1)
    loss.validate_targets = False
    cfg = config_3d_fullres_nextou(patch_size=(64, 224, 192), base=33, max_features=324, batch_size=2)
    _, target = synthetic_batch(cfg, 1, 14, 2, dev, blob_labels=True)
    rows = []
    print("%-16s %-44s %10s %12s %7s" % ("scale", "kernel", "us/launch", "achieved", "frac"))
    shape = [64, 224, 192]
    for scale, pool in enumerate([(1, 1, 1), (1, 2, 2), (2, 2, 2), (2, 2, 2)]):
        shape = [s // p for s, p in zip(shape, pool)]
        label = "x".join(map(str, shape))
        if args.only and args.only not in label:
            continue
        t = torch.nn.functional.interpolate(target, size=shape, mode="nearest")
        g = torch.Generator(device=dev).manual_seed(7)
        logits = torch.randn((2, 14, *shape), generator=g, device=dev) * 1.5
        logits.scatter_add_(1, t.long(), torch.full_like(t, args.bti_margin))
        logits.requires_grad_(True)
        for it in range(2 + args.iters):
            if it == 2:
                torch.cuda.synchronize()
                L.nextou_profile_enable(16 * args.iters)
            v = loss(logits, t)
            torch.autograd.grad(v, logits)
        torch.cuda.synchronize()
        with torch.no_grad():
            crit = float(loss.critical_voxels_from_labels(graph_ops.argmax_labels(logits)).float().mean())
        buf = ctypes.create_string_buffer(1 << 20)
        L.nextou_profile_report(buf, len(buf))
        L.nextou_profile_enable(0)
        for r in json.loads(buf.value.decode()):
            per_s = r["ms"] / r["launches"] / 1e3
            ach = r["work"] / r["launches"] / per_s
            rows.append({"call": label, "kernel": r["kernel"], "bound": r["bound"], "us": per_s * 1e6, "achieved": ach,
                         "frac": ach / PEAK[r["bound"]], "critical_fraction": crit})
            print("%-16s %-44s %10.1f %8.0f GB/s %6.1f%%" % (label, r["kernel"][:44], per_s * 1e6, ach / 1e9, 100 * ach / PEAK["hbm"]))
        print("%-16s critical voxels: %.1f %%" % (label, 100 * crit))
    print("total K5 kernel time for one loss fwd+bwd over the four scales: %.3f ms" % (sum(r["us"] for r in rows) / 1e3))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


def bench_mrg(args, dev, L):
    """K2 + K7 (nextou_mr_grouped_rows) at the cfg-2 stage-2 Swin shape, train and eval variants, against the three launches it replaces;
    the channel-major K2 + K7 of the pooled graphs (nextou_mr_grouped_cm) at Pool s2 / s3.  --only swin | pool | "pool s2" | "pool s3"."""
    from torch import nn
    from nextou_amd.network_architecture import NexToU_Encoder_Decoder as encdec
    patch, strides = (64, 224, 192), [[1, 1, 1], [1, 2, 2]] + [[2, 2, 2]] * 4
    shapes, _ = encdec._stage_shapes(nn.Conv3d, patch, strides)
    _, _, window = encdec.gnn_stage_hyperparameters(nn.Conv3d, shapes[-1], 6)
    spatial, C, k, groups, B = tuple(shapes[2]), 132, 7, 6, 2
    shift = tuple(w // 2 for w in window)
    be = graph_ops._HIP
    g = torch.Generator(device=dev).manual_seed(1)
    vol = torch.randn((B, C) + spatial, generator=g, device=dev).contiguous(memory_format=torch.channels_last_3d)
    windows = graph_ops.window_gather(vol, window, shift)
    idx = graph_ops.knn_graph(windows, None, None, k)
    w = torch.randn((2 * C, 2 * C // groups), generator=g, device=dev) * 0.1
    n_, k_ = w.shape[0] // groups, w.shape[1]
    wt = w.reshape(groups, n_, k_).transpose(1, 2).reshape(groups * k_, n_).contiguous()
    print("windows", tuple(windows.shape), "volume", spatial, "window", tuple(window))
    only = (args.only or "").lower()
    pools = [p for p in (("pool s2", 132, 10752, 168, 14), ("pool s3", 264, 10752, 1344, 28)) if not only or only in p[0] or only == "pool"]
    pool_data = []
    pools = [p for p in pools if be.mr_grouped_cm_tiles(2, p[1], 6, 2 * p[1] // 6, p[2], p[3], p[4]) > 0]     # (Pool s3 streams its source in chunks: not taken by default)
    for (_, pc, pn, pm, pk) in pools:      # Pool s2 / s3 of cfg 2: the channel-major K2 + K7 launch (mr_grp_cm_kernel)
        gp = torch.Generator(device=dev).manual_seed(pc)
        pool_data.append((torch.randn((2, pc, pn), generator=gp, device=dev), torch.randn((2, pc, pm), generator=gp, device=dev),
                          torch.randint(0, pm, (2, pn, pk), generator=gp, device=dev, dtype=torch.int32),
                          torch.randn((2 * pc, 2 * pc // 6), generator=gp, device=dev) * 0.1, pk))
    for it in range(2 + args.iters):
        if it == 2:
            torch.cuda.synchronize()
            L.nextou_profile_enable(64 * args.iters)
        if not only or "swin" in only:
            _, arg, h0, _ = be.mr_grouped_rows(windows, idx, k, 1, w, groups, B, spatial, window, shift, True, True, True)
            be.mr_grouped_rows_bwd(h0, w, arg, groups, spatial, window, shift)
            if not only:        # (a counter run wants ONE launch shape per kernel name: --only swin stops here)
                be.mr_grouped_rows(windows, idx, k, 1, w, groups, B, spatial, window, shift, False, False, True)
                be.mr_grouped_rows(windows, idx, k, 1, w, groups, B, spatial, window, shift, False, False, False)
                agg, arg = be.mr_fwd(windows, None, idx, None, k, 1, want_arg=True)
                a0 = be.window_scatter(agg, None, spatial, window, shift)
                h0, _ = be.pw_rows_fused(a0, w, groups, want_stats=True)
                ga = be.pw_rows(h0, wt, None, groups)
                be.mr_bwd_arg(be.window_gather(ga, window, shift), arg, windows.shape[2], False)
        for (px, py, pidx, pw, pk) in pool_data:
            be.mr_grouped_cm(px, py, pidx, pk, 1, pw, 6, True, True, True)        # training: aggregate + arg tape written too
            if not only:
                be.mr_grouped_cm(px, py, pidx, pk, 1, pw, 6, False, False, True)  # eval
                be.mr_fwd(px, py, pidx, None, pk, 1, want_arg=True)               # what it replaces, first launch of three
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 20)
    L.nextou_profile_report(buf, len(buf))
    L.nextou_profile_enable(0)
    rows = []
    for r in json.loads(buf.value.decode()):
        per_s = r["ms"] / r["launches"] / 1e3
        ach = r["work"] / r["launches"] / per_s
        print("%-70s %8.1f us %8.1f GB/s %5.1f%%  x%d" % (r["kernel"][:70], per_s * 1e6, ach / 1e9, 100 * ach / PEAK[r["bound"]], r["launches"]))
        rows.append({"call": "mrg", "kernel": r["kernel"], "bound": r["bound"], "us": per_s * 1e6, "achieved": ach, "frac": ach / PEAK[r["bound"]]})
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="substring filter on the call label")
    ap.add_argument("--norm", action="store_true", help="bench K6 (norm + LeakyReLU) instead of K1/K2")
    ap.add_argument("--cl", action="store_true", help="with --norm: channels-last tensors")
    ap.add_argument("--mrg", action="store_true", help="bench the K2 + K7 kernel (aggregate + grouped conv) at the cfg-2 stage-2 Swin shape")
    ap.add_argument("--bti-margin", type=float, default=3.0, help="--bti: logit bonus of the labelled class over N(0, 1.5) noise; 3.0 -> the "
                    "arg-max disagrees with the labels at many voxels (~80 %% critical), 12.0 -> arg-max == labels (critical voxels only at "
                    "real label interfaces)")
    ap.add_argument("--bti", action="store_true", help="bench K5 (arg-max labels, critical map, critical-voxel CE) at the cfg-4 scales")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    if args.norm:
        return bench_norm(args, dev, L)
    if args.bti:
        return bench_bti(args, dev, L)
    if args.mrg:
        return bench_mrg(args, dev, L)
    calls = CFG2 if args.cfg == 2 else CFG5
    rows = []
    for label, B, C, N, M, k in calls:
        if args.only and args.only not in label:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn((B, C, N), generator=g, device=dev)
        y = None if M is None else torch.randn((B, C, M), generator=g, device=dev)
        rp = torch.randn((N, M or N), generator=g, device=dev) * 0.05
        x.requires_grad_(True)
        if y is not None:
            y.requires_grad_(True)
        gout = torch.randn((B, 2 * C, N), generator=g, device=dev)
        for it in range(2 + args.iters):
            if it == 2:
                torch.cuda.synchronize()
                L.nextou_profile_enable(64 * args.iters)
            idx = graph_ops.knn_graph(x, y, rp, k)
            out = graph_ops.mr_aggregate(x, idx, y)
            torch.autograd.grad(out, [x] if y is None else [x, y], gout)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 20)
        L.nextou_profile_report(buf, len(buf))
        L.nextou_profile_enable(0)
        for r in json.loads(buf.value.decode()):
            per_s = r["ms"] / r["launches"] / 1e3
            ach = r["work"] / r["launches"] / per_s
            rows.append({"call": label, "kernel": r["kernel"], "bound": r["bound"], "us": per_s * 1e6,
                         "achieved": ach, "frac": ach / PEAK[r["bound"]]})
    print("%-8s %-66s %5s %10s %14s %7s" % ("call", "kernel", "bound", "us/launch", "achieved", "frac"))
    for r in rows:
        unit = "TFLOP/s" if r["bound"] == "mfma" else "GB/s"
        val = r["achieved"] / (1e12 if r["bound"] == "mfma" else 1e9)
        print("%-8s %-66s %5s %10.1f %9.2f %-7s %6.1f%%" % (r["call"], r["kernel"][:66], r["bound"], r["us"], val, unit,
                                                          100 * r["frac"]))
    print("total own-kernel time for one fwd+bwd over these calls: %.3f ms" % (sum(r["us"] for r in rows) / 1e3))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
