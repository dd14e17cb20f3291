#!/usr/bin/env python
"""Where the conv stages' time goes on PyTorch-ROCm (MIOpen find mode), per layer shape of cfg 2.

For each shape: forward, backward-data (grad wrt input only), backward-weight (grad wrt weight only), and the
alternative formulation of backward-data as a forward convolution with the flipped / transposed weight; each in
NCDHW and channels_last_3d.  Informational tool for DESIGN.md — nothing in the product path depends on it.

    python tools/conv_probe.py [--iters 5]
"""
import argparse
import os
import sys

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (label, B, Cin, Cout, spatial, kernel)
SHAPES = [("s0 33->33 [1,3,3]", 2, 33, 33, (64, 224, 192), (1, 3, 3)),
          ("s0 66->33 [1,3,3] (decoder, after concat)", 2, 66, 33, (64, 224, 192), (1, 3, 3)),
          ("s1 66->66 [3,3,3]", 2, 66, 66, (64, 112, 96), (3, 3, 3)),
          ("s1 132->66 [3,3,3] (decoder)", 2, 132, 66, (64, 112, 96), (3, 3, 3)),
          ("s2 132->132 [3,3,3]", 2, 132, 132, (32, 56, 48), (3, 3, 3))]


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
    print("%-44s %-8s %9s %9s %9s %14s" % ("layer", "layout", "fwd ms", "dgrad ms", "wgrad ms", "dgrad-as-fwd ms"))
    for label, B, ci, co, sp, k in SHAPES:
        pad = tuple(i // 2 for i in k)
        for layout in ("ncdhw", "ndhwc"):
            mf = torch.channels_last_3d if layout == "ndhwc" else torch.contiguous_format
            x = torch.randn((B, ci) + sp, device=dev).contiguous(memory_format=mf)
            w = torch.randn((co, ci) + k, device=dev).contiguous(memory_format=mf) * 0.05
            b = torch.zeros(co, device=dev)
            gy = torch.randn((B, co) + sp, device=dev).contiguous(memory_format=mf)
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            t_f = timeit(lambda: F.conv3d(x, w, b, padding=pad), args.iters)
            y1 = F.conv3d(xg, w, b, padding=pad)
            t_d = timeit(lambda: torch.autograd.grad(y1, xg, gy, retain_graph=True), args.iters)
            y2 = F.conv3d(x, wg, b, padding=pad)
            t_w = timeit(lambda: torch.autograd.grad(y2, wg, gy, retain_graph=True), args.iters)
            wt = w.transpose(0, 1).flip(2, 3, 4).contiguous(memory_format=mf)
            t_a = timeit(lambda: F.conv3d(gy, wt, None, padding=pad), args.iters)
            (ref,) = torch.autograd.grad(y1, xg, gy, retain_graph=True)
            alt = F.conv3d(gy, wt, None, padding=pad)
            err = float((ref - alt).abs().max() / ref.abs().max())
            print("%-44s %-8s %9.2f %9.2f %9.2f %10.2f (rel err %.1e)" % (label, layout, t_f, t_d, t_w, t_a, err))
            del x, w, gy, xg, wg, y1, y2, wt, ref, alt
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
