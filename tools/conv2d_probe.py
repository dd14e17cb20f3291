#!/usr/bin/env python
"""Stage 0 of the 3-D NexToU has convolution kernels [1,3,3]: no extent along the depth axis.  On a channels-last volume
(B, D, H, W, C) such a convolution is exactly a 2-D convolution of the (B*D, H, W, C) view — zero-copy.  Does MIOpen run the
2-D problem faster than the 3-D one?

    python tools/conv2d_probe.py [--iters 5]

Informational.
"""
import argparse
import os

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# (label, Cin, Cout, transposed)
LAYERS = [("s0 conv0 4->40", 4, 40, False), ("s0 conv1 40->40", 40, 40, False), ("s0 dec conv0 80->40", 80, 40, False),
          ("up s1->s0 72->40 T(1,2,2)", 72, 40, True)]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    B, D, H, W = 2, 64, 224, 192
    print("| layer | 3-D fwd / dgrad / wgrad ms | 2-D view fwd / dgrad / wgrad ms | max abs diff |")
    print("|---|---|---|---:|")
    for label, ci, co, transposed in LAYERS:
        hi, wi = (H // 2, W // 2) if transposed else (H, W)
        x3 = torch.randn((B, ci, D, hi, wi), device=dev).contiguous(memory_format=torch.channels_last_3d)
        x2 = x3.permute(0, 2, 1, 3, 4).reshape(B * D, ci, hi, wi)          # view of the same NDHWC memory, NHWC strides
        assert x2.data_ptr() == x3.data_ptr() and x2.is_contiguous(memory_format=torch.channels_last)
        if transposed:
            w3 = (torch.randn((ci, co, 1, 2, 2), device=dev) * 0.05)
            f3 = lambda a, w: F.conv_transpose3d(a, w, None, stride=(1, 2, 2))       # noqa: E731
            f2 = lambda a, w: F.conv_transpose2d(a, w, None, stride=(2, 2))          # noqa: E731
        else:
            w3 = (torch.randn((co, ci, 1, 3, 3), device=dev) * 0.05)
            f3 = lambda a, w: F.conv3d(a, w, None, padding=(0, 1, 1))                # noqa: E731
            f2 = lambda a, w: F.conv2d(a, w, None, padding=(1, 1))                   # noqa: E731
        w2 = w3.squeeze(2)
        y3, y2 = f3(x3, w3), f2(x2, w2)
        diff = float((y3.permute(0, 2, 1, 3, 4).reshape(y2.shape) - y2).abs().max())
        res = []
        for x, w, f in ((x3, w3, f3), (x2, w2, f2)):
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            tf = timeit(lambda: f(x, w), args.iters)
            y = f(xg, w)
            gy = torch.randn_like(y)
            td = timeit(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True), args.iters)
            yw = f(x, wg)
            tw = timeit(lambda: torch.autograd.grad(yw, wg, gy, retain_graph=True), args.iters)
            res.append((tf, td, tw))
            del xg, wg, y, yw, gy
        print("| %s | %.3f / %.3f / %.3f | %.3f / %.3f / %.3f | %.1e |" % ((label,) + res[0] + res[1] + (diff,)), flush=True)
        del x3, x2, y3, y2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
