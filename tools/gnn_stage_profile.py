#!/usr/bin/env python
"""Where the time of the GNN stages goes: every Pool-GNN / Swin-GNN block and FFN of the cfg-2 (or cfg-5) network,
built stand-alone at its real shape, timed forward and forward+backward with HIP events, and — with --kernels — the
per-kernel breakdown of one forward+backward of the block from torch.profiler.

    python tools/gnn_stage_profile.py [--cfg 2] [--iters 10] [--kernels] [--only "s2 Swin"]

Informational (DESIGN.md §5 'inside the GNN stages'); nothing in the product path depends on it.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NEXTOU_FAST_RELPOS", "1")      # timing only: the separable position tables build in seconds
import torch  # noqa: E402
from torch import nn  # noqa: E402

from nextou_amd.network_architecture import NexToU_Encoder_Decoder as encdec  # noqa: E402
from nextou_amd.network_architecture.norm_act import fuse_norm_act  # noqa: E402

CFGS = {
    2: dict(patch=(64, 224, 192), feats=[33, 66, 132, 264, 324, 324]),
    5: dict(patch=(96, 256, 256), feats=[33, 66, 132, 264, 324, 324]),
}
STRIDES = [[1, 1, 1], [1, 2, 2]] + [[2, 2, 2]] * 4


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
    return e0.elapsed_time(e1) / iters * 1e3      # us


def timeit_graph(fn, iters):
    """``fn`` captured once as a hipGraph and replayed: what the bench's captured step pays for the block (no host launch gaps)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return timeit(g.replay, iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--graph", action="store_true", help="also time each block as a replayed hipGraph (the bench's mode)")
    ap.add_argument("--stages", default="2,3,4,5")
    ap.add_argument("--cl", action="store_true", help="channels_last_3d inputs: the NDHWC graph-stage path (fused window / pool kernels)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    cfg = CFGS[args.cfg]
    shapes, _ = encdec._stage_shapes(nn.Conv3d, cfg["patch"], STRIDES)
    opt = encdec.OptInit(pool_op_kernel_sizes_len=6)
    opt.img_min_shape = shapes[-1]
    opt.n_size_list = [int(torch.tensor(s).prod()) for s in shapes]
    kw = dict(opt=opt, conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None)
    print("%-12s %-14s %10s %12s %10s %12s %14s   (x2 per step: encoder + decoder; s5 x1)" % (
        "stage", "module", "fwd us", "fwd+bwd us", "bwd us", "graph fwd us", "graph f+b us"))
    total_f = total_fb = total_gf = total_gfb = 0.0
    for s in [int(v) for v in args.stages.split(",")]:
        C = cfg["feats"][s]
        blocks = [("Pool", encdec.PoolGNNBlocks(C, shapes[s], s - 2, 2, **kw)), ("Swin", encdec.SwinGNNBlocks(C, shapes[s], s - 2, **kw))]
        for kind, blk in blocks:
            fuse_norm_act(blk)
            blk = blk.to(dev).train()
            grapher, ffn = blk.blocks[0][0], blk.blocks[0][1]
            x = torch.randn((args.batch, C) + tuple(shapes[s]), device=dev)
            if args.cl:
                x = x.contiguous(memory_format=torch.channels_last_3d)
            x.requires_grad_(True)
            gy = torch.randn_like(x)
            for name, m in (("%sGrapher" % kind, grapher), ("FFN", ffn)):
                label = "s%d %s" % (s, kind)
                if args.only and args.only not in label + " " + name:
                    continue

                def fwd():
                    with torch.no_grad():
                        return m(x)

                def fwdbwd():
                    y = m(x)
                    torch.autograd.grad(y, [x] + [p for p in m.parameters() if p.requires_grad], gy)
                tf, tfb = timeit(fwd, args.iters), timeit(fwdbwd, args.iters)
                gf = gfb = float("nan")
                if args.graph:
                    gf, gfb = timeit_graph(fwd, args.iters), timeit_graph(fwdbwd, args.iters)
                mult = 1 if s == 5 else 2
                total_f += mult * tf
                total_fb += mult * tfb
                total_gf += mult * gf
                total_gfb += mult * gfb
                print("%-12s %-14s %10.1f %12.1f %10.1f %12.1f %14.1f" % (label, name, tf, tfb, tfb - tf, gf, gfb), flush=True)
                if args.kernels:
                    from torch.profiler import ProfilerActivity, profile
                    with profile(activities=[ProfilerActivity.CUDA]) as prof:
                        for _ in range(3):
                            fwdbwd()
                        torch.cuda.synchronize()
                    rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
                    print("      kernel time %.1f us over %d launches" % (sum(e.self_device_time_total for e in rows) / 3.0,
                                                                      sum(e.count for e in rows) // 3))
                    for e in rows[:40]:
                        print("      %9.1f us x%-3d %s" % (e.self_device_time_total / 3.0, e.count // 3, e.key[:110]))
            del blk, x, gy
            torch.cuda.empty_cache()
    print("sum over one step's GNN blocks: forward %.2f ms, forward+backward %.2f ms" % (total_f / 1e3, total_fb / 1e3))
    if args.graph:
        print("as replayed hipGraphs: forward %.2f ms, forward+backward %.2f ms" % (total_gf / 1e3, total_gfb / 1e3))


if __name__ == "__main__":
    main()
