#!/usr/bin/env python
"""Can a large library weight gradient hide behind the small-stage GNN blocks?

The stage-4 / 5 blocks of cfg 2 are chains of 5-40-us kernels on a few dozen workgroups each (3.4 ms of a step as replayed graphs) — most of the chip
idles while they run.  A convolution's weight gradient is off the backward's critical path (only the optimizer reads it).  This probe times, on one
MI355X: the chain (stage-4 + stage-5 Grapher / FFN blocks, forward + backward, one replayed hipGraph) alone, a weight gradient alone (the model's route:
depth-flat / depth-unrolled 2-D problem on MIOpen), both one after the other on one stream, and both at once on two streams.

    python tools/small_stage_overlap_probe.py [--iters 10]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
from torch import nn  # noqa: E402

from nextou_amd import graph_ops  # noqa: E402
from nextou_amd.network_architecture import NexToU_Encoder_Decoder as encdec  # noqa: E402
from nextou_amd.network_architecture.norm_act import fuse_norm_act  # noqa: E402
from tools.gnn_stage_profile import CFGS, STRIDES  # noqa: E402

WGRADS = [   # name, B, Cin, Cout, D, H, W, kernel
    ("s0 40->40 1x3x3", 2, 40, 40, 64, 224, 192, (1, 3, 3)),
    ("s0 80->40 1x3x3 (decoder)", 2, 80, 40, 64, 224, 192, (1, 3, 3)),
    ("s1 72->72 3x3x3", 2, 72, 72, 64, 112, 96, (3, 3, 3)),
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


def small_stage_chain(dev, batch=2):
    """forward + backward of the stage-4 blocks (twice: encoder + decoder) and the stage-5 blocks of cfg 2 as one callable"""
    cfg = CFGS[2]
    shapes, _ = encdec._stage_shapes(nn.Conv3d, cfg["patch"], STRIDES)
    opt = encdec.OptInit(pool_op_kernel_sizes_len=6)
    opt.img_min_shape = shapes[-1]
    opt.n_size_list = [int(torch.tensor(s).prod()) for s in shapes]
    kw = dict(opt=opt, conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None)
    work = []
    for s, times in ((4, 2), (5, 1)):
        C = cfg["feats"][s]
        for blk in (encdec.PoolGNNBlocks(C, shapes[s], s - 2, 2, **kw), encdec.SwinGNNBlocks(C, shapes[s], s - 2, **kw)):
            fuse_norm_act(blk)
            blk = blk.to(dev).train()
            x = torch.randn((batch, C) + tuple(shapes[s]), device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
            work.append((blk, x, torch.randn_like(x), times))

    def run():
        for blk, x, gy, times in work:
            for _ in range(times):
                y = blk(x)
                torch.autograd.grad(y, [x] + [p for p in blk.parameters() if p.requires_grad], gy)
    return run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    side = torch.cuda.Stream()
    chain = small_stage_chain(dev)
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for _ in range(3):
            chain()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    t_chain = timed(g.replay, a.iters)
    print("small-stage chain (s4 x2 + s5, forward + backward, replayed hipGraph): %.0f us" % t_chain)
    print("%-30s %10s %12s %12s %8s %22s" % ("weight gradient", "alone us", "one stream", "two streams", "gain", "hidden share of chain"))
    mf = torch.channels_last_3d
    for name, B, ci, co, D, H, W, k in WGRADS:
        x = torch.randn(B, ci, D, H, W, device=dev).contiguous(memory_format=mf)
        gy = torch.randn(B, co, D, H, W, device=dev).contiguous(memory_format=mf)
        w = torch.randn((co, ci) + k, device=dev).contiguous(memory_format=mf)
        pad = tuple(v // 2 for v in k)

        def wgrad():
            if graph_ops.wgrad_depth_unroll_eligible(x, w, pad):
                x3 = graph_ops._HIP.depth_unroll(x)
                w2 = w.permute(0, 2, 1, 3, 4).reshape(co, 3 * ci, 3, 3)
                return torch.ops.aten.convolution_backward(graph_ops.flat_depth(gy), x3, w2, None, (1, 1), pad[1:], (1, 1), False, (0, 0), 1,
                                                           [False, True, False])[1]
            return torch.ops.aten.convolution_backward(graph_ops.flat_depth(gy), graph_ops.flat_depth(x), w.squeeze(2), None, (1, 1), pad[1:], (1, 1),
                                                       False, (0, 0), 1, [False, True, False])[1]

        keep = []

        def one_stream():
            g.replay()
            keep[:] = [wgrad()]

        def two_streams():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                keep[:] = [wgrad()]
            g.replay()
            torch.cuda.current_stream().wait_stream(side)

        tw = timed(wgrad, a.iters)
        t1, t2 = timed(one_stream, a.iters), timed(two_streams, a.iters)
        print("%-30s %10.0f %12.0f %12.0f %7.1f%% %21.0f%%" % (name, tw, t1, t2, 100.0 * (t1 - t2) / t1, 100.0 * (t1 - t2) / t_chain), flush=True)


if __name__ == "__main__":
    main()
