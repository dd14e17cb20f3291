#!/usr/bin/env python
"""K8 (segmentation heads, csrc/head_rows.hip) at the cfg-2 head shapes, next to the library route it replaces
(graph_ops.conv_own_bias_grad: MIOpen convolution forward / backward + K6's channel sum for the bias gradient).

    python tools/head_bench.py [--iters 20]

Own kernels: per-launch HIP-event times from the library's launch profiler, achieved GB/s on the algorithmic bytes
4 P (C + L) against 8 TB/s.  Both routes: wall time of forward + backward (torch.cuda events around `iters` repetitions).
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextou_amd import _lib, graph_ops  # noqa: E402

# (label, B, C, spatial)  — decoder outputs of cfg 2 (batch 2), channel counts as padded by channel_pad.py
HEADS = [("full res 40->14", 2, 40, (64, 224, 192)), ("1/2 res 72->14", 2, 72, (64, 112, 96)), ("s2 132->14", 2, 132, (32, 56, 48)),
         ("s3 264->14", 2, 264, (16, 28, 24)), ("s4 324->14", 2, 324, (8, 14, 12))]


def wall(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--classes", type=int, default=14)
    ap.add_argument("--only", default=None, help="substring filter on the head label")
    ap.add_argument("--json", default=None, help="write the own kernels' rows (launch label, us, GB/s) here")
    ap.add_argument("--own-only", action="store_true", help="skip the library route (counter runs)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L_ = _lib.lib()
    torch.backends.cudnn.benchmark = True
    nl = args.classes
    print("%-18s %-44s %10s %10s %7s" % ("head", "kernel", "us/launch", "GB/s", "frac"))
    rows = []
    jrows = []
    for label, B, C, sp in HEADS:
        if args.only and args.only not in label:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn((B, C) + sp, generator=g, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        w = (torch.randn((nl, C, 1, 1, 1), generator=g, device=dev) * 0.1).requires_grad_(True)
        b = torch.zeros(nl, device=dev, requires_grad=True)
        gy = torch.randn((B, nl) + sp, generator=g, device=dev).contiguous(memory_format=torch.channels_last_3d)

        def own():
            torch.autograd.grad(graph_ops.head_rows(x, w, b), (x, w, b), gy)

        def lib():
            y = graph_ops.conv_own_bias_grad(x, w, b, (1, 1, 1), (0, 0, 0), (1, 1, 1), False, (0, 0, 0), 1)
            torch.autograd.grad(y, (x, w, b), gy)

        t_lib = 0.0 if args.own_only else wall(lib, args.iters)
        t_own = wall(own, args.iters)
        L_.nextou_profile_enable(8 * args.iters)
        for _ in range(args.iters):
            own()
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 20)
        L_.nextou_profile_report(buf, len(buf))
        L_.nextou_profile_enable(0)
        for r in json.loads(buf.value.decode()):
            us = r["ms"] / r["launches"] * 1e3
            gbs = r["work"] / r["launches"] / (us * 1e-6) / 1e9
            print("%-18s %-44s %10.1f %10.0f %7.3f" % (label, r["kernel"], us, gbs, gbs / 8000.0))
            jrows.append({"call": label, "kernel": r["kernel"], "us": us, "achieved": gbs * 1e9, "frac": gbs / 8000.0})
        print("%-18s forward + backward wall: own %.1f us, library route %.1f us" % (label, t_own, t_lib))
        rows.append((label, t_own, t_lib))
    print("\nsum over the five heads: own %.1f us, library route %.1f us" % (sum(r[1] for r in rows), sum(r[2] for r in rows)))
    if args.json:
        json.dump(jrows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
