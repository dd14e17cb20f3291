#!/usr/bin/env python
"""Per-layer table of the part of the cfg-2 step that north_star leaves on PyTorch-ROCm (MIOpen / CK / rocBLAS), AT THE ROUTING THE
MODEL REALLY USES (the round-2 table predates dgrad-as-forward, the depth-unrolled weight gradient, the 33 -> 40 /
66 -> 72 channel padding and the channels-last layout).

One forward of the real network records, for every convolution module (plain stages' 3x3x3 / 1x3x3 convolutions, the strided stage
entries, the transposed up-convolutions, the 1x1 convolutions that still go to a library), the input it saw; each module is then
replayed ALONE through its own forward — i.e. through the same route selection in network_architecture/norm_act.py (channel padding,
depth-flat 2-D view, dgrad as a forward convolution, depth-unrolled 2-D weight gradient, rows GEMM ...) — on a tensor of that shape,
layout and requires_grad state, forward and forward + backward, HIP-event timed.

    python tools/conv_layer_table.py [--iters 5] [--md out.md]

FLOPs: 2 x output voxels x Cout x (Cin / groups) x prod(kernel) with the REAL (un-padded) channel counts; backward = 2 x forward.
Informational tool; nothing in the product path depends on it.
"""
import argparse
import os
import sys

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
os.environ.setdefault("NEXTOU_FAST_RELPOS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

PEAK_TF = 157.3


def timeit(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--md", default=None)
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    trainer, cfg, batch, classes = bench.build_trainer("cfg2", dev, False)
    bench.move_to(trainer, dev)
    net = trainer.network.train()
    from nextou_amd.harness import synthetic_batch
    data, _ = synthetic_batch(cfg, 1, classes, batch, dev)

    convs = [(n, m) for n, m in net.named_modules()
             if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)) and not n.startswith("decoder.encoder")]
    seen = {}
    handles = []
    for name, m in convs:
        def hook(mod, inp, out, name=name):
            x = inp[0]
            seen.setdefault(name, []).append((tuple(x.shape), tuple(x.stride()), x.requires_grad, tuple(out.shape)))
        handles.append(m.register_forward_hook(hook))
    # a second up-convolution path: norm_act.up_conv_cat calls conv_own_bias_grad directly, not the module — record through the wrapper
    from nextou_amd import graph_ops
    direct = []
    real = graph_ops.conv_own_bias_grad

    def spy(x, weight, bias, stride, padding, dilation, transposed, output_padding, groups):
        y = real(x, weight, bias, stride, padding, dilation, transposed, output_padding, groups)
        direct.append((tuple(x.shape), tuple(x.stride()), weight.detach(), tuple(stride), tuple(padding), tuple(dilation), bool(transposed),
                       tuple(output_padding), int(groups), tuple(y.shape)))
        return y
    graph_ops.conv_own_bias_grad = spy
    x = data.clone().requires_grad_(False)
    outs = net(x)
    sum(o.float().sum() for o in outs).backward()
    graph_ops.conv_own_bias_grad = real
    for h in handles:
        h.remove()
    net.zero_grad(set_to_none=True)
    torch.cuda.synchronize()

    rows = []
    mods = dict(convs)
    for name, calls in seen.items():
        m = mods[name]
        shape, strides, rg, oshape = calls[0]
        g = torch.Generator(device=dev).manual_seed(1)
        base = torch.empty_strided(shape, strides, device=dev, dtype=torch.float32)
        base.copy_(torch.randn(shape, generator=g, device=dev))
        gy = torch.randn(oshape, generator=g, device=dev)

        def fwd():
            with torch.no_grad():
                m(base)

        def fwdbwd():
            xi = base.detach().requires_grad_(True)      # (the first layer of the real net gets no input gradient: slightly pessimistic there)
            y = m(xi)
            torch.autograd.grad(y, [xi] + [p for p in m.parameters() if p.requires_grad], gy.to(y.dtype) if gy.shape == y.shape else torch.ones_like(y),
                                allow_unused=True)

        tf = timeit(fwd, args.iters)
        tfb = timeit(fwdbwd, args.iters)
        w = m.weight
        transposed = isinstance(m, (nn.ConvTranspose2d, nn.ConvTranspose3d))
        cin, cout = (w.shape[0], w.shape[1] * m.groups) if transposed else (w.shape[1] * m.groups, w.shape[0])
        k = int(np.prod(w.shape[2:]))
        vox = int(np.prod((shape if transposed else oshape)[2:])) * shape[0]
        flop = 2.0 * vox * cout * (cin / m.groups) * k
        layout = "NDHWC" if strides[1] == 1 and len(shape) == 5 and shape[1] > 1 else "NCDHW"
        rows.append((name, "%d->%d%s k%s s%s g%d" % (cin, cout, " T" if transposed else "", "x".join(map(str, w.shape[2:])), "x".join(map(str, m.stride)), m.groups),
                     "x".join(map(str, shape)), layout, len(calls), flop, tf, tfb - tf))
    for i, (shape, strides, wt, stride, padding, dilation, transposed, out_pad, groups, oshape) in enumerate(direct):
        g = torch.Generator(device=dev).manual_seed(2)
        base = torch.empty_strided(shape, strides, device=dev, dtype=torch.float32)
        base.copy_(torch.randn(shape, generator=g, device=dev))
        gy = torch.randn(oshape, generator=g, device=dev).contiguous(memory_format=torch.channels_last_3d if len(oshape) == 5 else torch.channels_last)
        wp = wt.clone().requires_grad_(True)

        def fwd():
            with torch.no_grad():
                real(base, wp, None, stride, padding, dilation, transposed, out_pad, groups)

        def fwdbwd():
            xi = base.detach().requires_grad_(True)
            torch.autograd.grad(real(xi, wp, None, stride, padding, dilation, transposed, out_pad, groups), [xi, wp], gy)

        tf = timeit(fwd, args.iters)
        tfb = timeit(fwdbwd, args.iters)
        cin, cout = (wt.shape[0], wt.shape[1] * groups) if transposed else (wt.shape[1] * groups, wt.shape[0])
        k = int(np.prod(wt.shape[2:]))
        vox = int(np.prod((shape if transposed else oshape)[2:])) * shape[0]
        flop = 2.0 * vox * cout * (cin / groups) * k
        rows.append(("up_conv_cat #%d" % i, "%d->%d%s k%s s%s g%d (padded counts)" % (cin, cout, " T" if transposed else "", "x".join(map(str, wt.shape[2:])),
                                                                                   "x".join(map(str, stride)), groups),
                     "x".join(map(str, shape)), "NDHWC" if strides[1] == 1 else "NCDHW", 1, flop, tf, tfb - tf))
    rows.sort(key=lambda r: -(r[6] + r[7]) * r[4])
    lines = ["# cfg 2, batch 2: the convolution modules of the network, each replayed alone through its own forward (the route the model takes), MI355X fp32",
             "", "`python tools/conv_layer_table.py` — forward and backward (data + weight gradient) HIP-event times, FLOPs with the real channel counts, of the fp32 MFMA peak 157.3 TFLOP/s.",
             "", "| module | conv | input (as seen in the model: padded channels) | layout | calls / step | GFLOP fwd | fwd us | fwd TF/s | % peak | bwd us | bwd TF/s | % peak |",
             "|---|---|---|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    tot_f = tot_b = tot_flop = 0.0
    for name, desc, shp, layout, n, flop, tf, tb in rows:
        ftf, btf = flop / tf / 1e6, 2 * flop / max(tb, 1e-3) / 1e6
        lines.append("| %s | %s | %s | %s | %d | %.2f | %.0f | %.1f | %.0f%% | %.0f | %.1f | %.0f%% |" % (
            name, desc, shp, layout, n, flop / 1e9, tf, ftf, 100 * ftf / PEAK_TF, tb, btf, 100 * btf / PEAK_TF))
        tot_f += tf * n
        tot_b += tb * n
        tot_flop += flop * n
    lines += ["", "sum over the recorded module calls: forward %.1f ms, backward %.1f ms, %.2f TFLOP forward => %.1f / %.1f TF/s (%.0f%% / %.0f%% of peak)" % (
        tot_f / 1e3, tot_b / 1e3, tot_flop / 1e12, tot_flop / tot_f / 1e6, 2 * tot_flop / tot_b / 1e6, 100 * tot_flop / tot_f / 1e6 / PEAK_TF,
        100 * 2 * tot_flop / tot_b / 1e6 / PEAK_TF),
        "", "(`up_conv_cat #i`: the decoder's up-convolutions, called through `norm_act.up_conv_cat` -> graph_ops.conv_own_bias_grad with the bias folded into the "
            "concatenation kernel; their FLOPs use the padded channel counts the call carries.)"]
    text = "\n".join(lines)
    print(text)
    if args.md:
        open(args.md, "w").write(text + "\n")


if __name__ == "__main__":
    main()
