#!/usr/bin/env python
"""A [3,3,3] stride-1 convolution of a channels-last volume as ONE 2-D convolution: the three depth taps become input channels
(x3[b, d] = cat(x[b, d-1], x[b, d], x[b, d+1]) over channels, zero slices at the ends; weight (Cout, 3*Cin, 3, 3)), which sends the
problem to MIOpen's 2-D igemm assembly kernels instead of CK's 3-D ones.  Same FLOPs, one extra pass to build x3.

    python tools/conv_depth_unroll_probe.py [--iters 5]

Informational: decides whether the stage-1 convolutions take that route.
"""
import argparse
import os

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# (label, Cin, Cout, (D, H, W))
LAYERS = [("s1 conv 72->72", 72, 72, (64, 112, 96)), ("s1 dec conv0 144->72", 144, 72, (64, 112, 96)),
          ("s1 dgrad-as-fwd 72->144", 72, 144, (64, 112, 96)), ("s2 dec conv 264->132", 264, 132, (32, 56, 48))]


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


def unroll_depth(x):
    """(B, C, D, H, W) channels_last_3d -> (B*D, 3C, H, W) channels_last: depth taps -1, 0, +1 stacked over channels."""
    b, c, d, h, w = x.shape
    rows = x.permute(0, 2, 3, 4, 1)                                   # (B, D, H, W, C) contiguous view
    out = torch.empty((b, d, h, w, 3 * c), device=x.device, dtype=x.dtype)
    out[:, 1:, :, :, 0:c] = rows[:, :-1]
    out[:, 0, :, :, 0:c] = 0
    out[:, :, :, :, c:2 * c] = rows
    out[:, :-1, :, :, 2 * c:] = rows[:, 1:]
    out[:, -1, :, :, 2 * c:] = 0
    return out.reshape(b * d, h, w, 3 * c).permute(0, 3, 1, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    B = 2
    print("| layer | GFLOP | conv3d fwd / wgrad ms (TF/s) | unroll ms | conv2d(3C) fwd / wgrad ms (TF/s) | max abs diff |")
    print("|---|---:|---|---:|---|---:|")
    for label, ci, co, sp in LAYERS:
        x = torch.randn((B, ci) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
        w = (torch.randn((co, ci, 3, 3, 3), device=dev) * 0.05)
        fl = 2.0 * B * sp[0] * sp[1] * sp[2] * ci * co * 27
        wg = w.clone().requires_grad_(True)
        t3f = timeit(lambda: F.conv3d(x, w, None, padding=1), args.iters)
        y3 = F.conv3d(x, wg, None, padding=1)
        gy = torch.randn_like(y3)
        t3w = timeit(lambda: torch.autograd.grad(y3, wg, gy, retain_graph=True), args.iters)
        tu = timeit(lambda: unroll_depth(x), args.iters)
        x3 = unroll_depth(x)
        w2 = w.permute(0, 2, 1, 3, 4).reshape(co, 3 * ci, 3, 3).contiguous(memory_format=torch.channels_last)
        w2g = w2.clone().requires_grad_(True)
        t2f = timeit(lambda: F.conv2d(x3, w2, None, padding=1), args.iters)
        y2 = F.conv2d(x3, w2g, None, padding=1)
        g2 = gy.permute(0, 2, 1, 3, 4).reshape(y2.shape)
        t2w = timeit(lambda: torch.autograd.grad(y2, w2g, g2, retain_graph=True), args.iters)
        diff = float((y3.permute(0, 2, 1, 3, 4).reshape(y2.shape) - y2).abs().max())
        print("| %s | %.0f | %.3f (%.0f) / %.3f (%.0f) | %.3f | %.3f (%.0f) / %.3f (%.0f) | %.1e |" % (
            label, fl / 1e9, t3f, fl / t3f / 1e9, t3w, fl / t3w / 1e9, tu, t2f, fl / t2f / 1e9, t2w, fl / t2w / 1e9, diff), flush=True)
        del x, w, wg, y3, gy, x3, w2, w2g, y2, g2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
