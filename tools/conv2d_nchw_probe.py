#!/usr/bin/env python
"""Stage-0 convolutions ([1,3,3] kernels) as 2-D problems in BOTH memory layouts: the channels-last (B*D, H, W, C) view the model
uses (MIOpen: igemm_*_gtcx35_nhwc_fp32 assembly kernels) against plain NCHW (B*D, C, H, W) tensors, where MIOpen's fp32 Winograd
solvers could apply.  Prints forward / data-gradient / weight-gradient times and the kernels that ran.

    python tools/conv2d_nchw_probe.py [--iters 5]

Informational: decides whether a depth-major NCHW layout for stage 0 is worth building.
"""
import argparse
import os

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

LAYERS = [("s0 conv1 40->40", 40, 40), ("s0 dec conv0 80->40", 80, 40), ("s0 conv0 4->40", 4, 40),
          ("unpadded 33->33", 33, 33), ("64->64", 64, 64)]


def timeit(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernels_of(fn):
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    rows = sorted(((e.self_device_time_total, e.key) for e in prof.key_averages() if e.self_device_time_total > 0), reverse=True)
    return "; ".join("%s %.0fus" % (k[:48], t) for t, k in rows[:3])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    N, H, W = 128, 224, 192
    for label, ci, co in LAYERS:
        for layout in ("nhwc", "nchw"):
            x = torch.randn((N, ci, H, W), device=dev)
            w = torch.randn((co, ci, 3, 3), device=dev) * 0.05
            if layout == "nhwc":
                x = x.contiguous(memory_format=torch.channels_last)
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            fwd = lambda: F.conv2d(x, w, None, padding=1)                                      # noqa: E731
            y = F.conv2d(xg, w, None, padding=1)
            gy = torch.randn_like(y)
            dgr = lambda: torch.autograd.grad(y, xg, gy, retain_graph=True)                    # noqa: E731
            yw = F.conv2d(x, wg, None, padding=1)
            wgr = lambda: torch.autograd.grad(yw, wg, gy, retain_graph=True)                   # noqa: E731
            tf, td, tw = timeit(fwd, args.iters), (timeit(dgr, args.iters) if ci > 4 else float("nan")), timeit(wgr, args.iters)
            fl = 2.0 * N * H * W * ci * co * 9
            print("%-22s %s  fwd %.3f ms (%.0f TF/s)  dgrad %.3f  wgrad %.3f" % (label, layout, tf, fl / tf / 1e9, td, tw), flush=True)
            print("      fwd:   " + kernels_of(fwd))
            if ci > 4:
                print("      dgrad: " + kernels_of(dgr))
            print("      wgrad: " + kernels_of(wgr), flush=True)
            del x, w, xg, wg, y, yw, gy
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
