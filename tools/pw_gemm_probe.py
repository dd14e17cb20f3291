#!/usr/bin/env python
"""1x1 convolutions of the GNN stages on channels-last activations: MIOpen's convolution path (what nn.Conv3d takes)
against the same three products as plain library GEMMs on the (points, channels) matrix — forward x @ W^T, data gradient
gy @ W, weight gradient gy^T @ x (reduction over 172 032 points: a split-K shape).

    python tools/pw_gemm_probe.py [--iters 10]

Informational; decides whether the pointwise convolutions go through torch.mm.
"""
import argparse
import os
import sys

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# (label, B, Cin, Cout, spatial)
SHAPES = [("FFN s2 132->528", 2, 132, 528, (32, 56, 48)), ("FFN s2 528->132", 2, 528, 132, (32, 56, 48)),
          ("fc s2 132->132", 2, 132, 132, (32, 56, 48)), ("fc2 s2 264->132", 2, 264, 132, (32, 56, 48)),
          ("FFN s3 264->1056", 2, 264, 1056, (16, 28, 24)), ("FFN s3 1056->264", 2, 1056, 264, (16, 28, 24)),
          ("fc2 s3 528->264", 2, 528, 264, (16, 28, 24)), ("FFN s4 324->1296", 2, 324, 1296, (8, 14, 12)),
          ("head s0 40->14", 2, 40, 14, (64, 224, 192)), ("head s1 72->14", 2, 72, 14, (64, 112, 96))]


def timeit(fn, iters):
    for _ in range(3):
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
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    print("| layer | GFLOP | conv fwd / dgrad / wgrad ms | mm fwd / dgrad / wgrad ms | mm TF/s fwd / dgrad / wgrad |")
    print("|---|---:|---|---|---|")
    for label, B, ci, co, sp in SHAPES:
        P = B * sp[0] * sp[1] * sp[2]
        x = torch.randn((B, ci) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn((co, ci, 1, 1, 1), device=dev) * 0.05
        gy = torch.randn((B, co) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        cf = timeit(lambda: F.conv3d(x, w), args.iters)
        y1 = F.conv3d(xg, w)
        cd = timeit(lambda: torch.autograd.grad(y1, xg, gy, retain_graph=True), args.iters)
        y2 = F.conv3d(x, wg)
        cw = timeit(lambda: torch.autograd.grad(y2, wg, gy, retain_graph=True), args.iters)
        x2 = x.permute(0, 2, 3, 4, 1).reshape(P, ci)        # views: channels-last memory IS the (P, C) matrix
        g2 = gy.permute(0, 2, 3, 4, 1).reshape(P, co)
        w2 = w.reshape(co, ci)
        assert x2.data_ptr() == x.data_ptr() and g2.data_ptr() == gy.data_ptr()
        mf = timeit(lambda: torch.mm(x2, w2.t()), args.iters)
        md = timeit(lambda: torch.mm(g2, w2), args.iters)
        mw = timeit(lambda: torch.mm(g2.t(), x2), args.iters)
        err = float((torch.mm(g2.t(), x2) - torch.autograd.grad(y2, wg, gy, retain_graph=True)[0].reshape(co, ci)).abs().max())
        fl = 2.0 * P * ci * co
        print("| %s | %.1f | %.3f / %.3f / %.3f | %.3f / %.3f / %.3f | %.0f / %.0f / %.0f | (wgrad max diff %.1e)" % (
            label, fl / 1e9, cf, cd, cw, mf, md, mw, fl / mf / 1e9, fl / md / 1e9, fl / mw / 1e9, err), flush=True)
        del x, w, gy, xg, wg, y1, y2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
