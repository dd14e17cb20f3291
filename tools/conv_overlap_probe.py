#!/usr/bin/env python
"""Do the data gradient and the weight gradient of one library convolution run faster side by side (two HIP streams) than one after the
other?  The convolutions of cfg 2 sit at 35-42 % of the fp32 MFMA peak; if what limits them is not the chip's throughput, two of them at once
may finish sooner than in sequence.  Times, per layer shape: dgrad alone, wgrad alone, both on one stream, both on two streams.

    python tools/conv_overlap_probe.py [--iters 10]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402

from nextou_amd import graph_ops  # noqa: E402

SHAPES = [   # name, B, Cin, Cout, D, H, W, kernel
    ("s1 72->72 3x3x3", 2, 72, 72, 64, 112, 96, (3, 3, 3)),
    ("s1 144->72 3x3x3 (decoder)", 2, 144, 72, 64, 112, 96, (3, 3, 3)),
    ("s0 40->40 1x3x3", 2, 40, 40, 64, 224, 192, (1, 3, 3)),
    ("s0 80->40 1x3x3 (decoder)", 2, 80, 40, 64, 224, 192, (1, 3, 3)),
    ("s2 264->132 3x3x3 (decoder)", 2, 264, 132, 32, 56, 48, (3, 3, 3)),
]


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    side = torch.cuda.Stream()
    mf = torch.channels_last_3d
    print("%-30s %10s %10s %12s %12s %8s" % ("layer", "dgrad us", "wgrad us", "one stream", "two streams", "gain"))
    for name, B, ci, co, D, H, W, k in SHAPES:
        x = torch.randn(B, ci, D, H, W, device=dev).contiguous(memory_format=mf)
        gy = torch.randn(B, co, D, H, W, device=dev).contiguous(memory_format=mf)
        w = torch.randn((co, ci) + k, device=dev).contiguous(memory_format=mf)
        pad = tuple(v // 2 for v in k)
        ones, zeros = (1, 1, 1), (0, 0, 0)
        wt = graph_ops._HIP.filter_flip_t(w)
        flat = graph_ops.flat_depth_eligible(gy, wt, ones, pad, ones)

        def dgrad():
            if flat:
                return torch.nn.functional.conv2d(graph_ops.flat_depth(gy), wt.squeeze(2), None, 1, pad[1:])
            return torch.ops.aten.convolution(gy, wt, None, ones, pad, ones, False, zeros, 1)

        def wgrad():
            if graph_ops.wgrad_depth_unroll_eligible(x, w, pad):
                x3 = graph_ops._HIP.depth_unroll(x)
                w2 = w.permute(0, 2, 1, 3, 4).reshape(co, 3 * ci, 3, 3)
                return torch.ops.aten.convolution_backward(graph_ops.flat_depth(gy), x3, w2, None, (1, 1), pad[1:], (1, 1), False, (0, 0), 1,
                                                           [False, True, False])[1]
            if flat:
                return torch.ops.aten.convolution_backward(graph_ops.flat_depth(gy), graph_ops.flat_depth(x), w.squeeze(2), None, (1, 1), pad[1:], (1, 1),
                                                           False, (0, 0), 1, [False, True, False])[1]
            return torch.ops.aten.convolution_backward(gy, x, w, None, ones, pad, ones, False, zeros, 1, [False, True, False])[1]

        keep = []

        def both_one():
            keep[:] = [dgrad(), wgrad()]

        def both_two():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                r2 = wgrad()
            r1 = dgrad()
            torch.cuda.current_stream().wait_stream(side)
            keep[:] = [r1, r2]

        td, tw = timed(dgrad, a.iters), timed(wgrad, a.iters)
        t1, t2 = timed(both_one, a.iters), timed(both_two, a.iters)
        print("%-30s %10.0f %10.0f %12.0f %12.0f %7.1f%%" % (name, td, tw, t1, t2, 100.0 * (t1 - t2) / t1), flush=True)


if __name__ == "__main__":
    main()
