#!/usr/bin/env python
"""Evidence for the part of the cfg-2 step that north_star leaves on PyTorch-ROCm (MIOpen / CK / rocBLAS):
per convolution layer of the network — shape, FLOPs, forward / backward-data / backward-weight time, TFLOP/s and
the fraction of the fp32 MFMA peak (157.3 TFLOP/s) — in the memory layout the layer runs in inside the model
(NDHWC for the plain stages 0 / 1, NCDHW for the graph stages), and the same layers with the channel count
zero-padded to multiples of 8 / 16 / 32 (does MIOpen's solver choice punish C = 33 / 66?).

    python tools/conv_evidence.py [--iters 5] [--pad] [--md out.md]

Run under ``rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA`` with ``--once``
to get the matrix-pipe occupancy of the kernels MIOpen picks (one launch per direction per layer).
Informational tool; nothing in the product path depends on it.
"""
import argparse
import os
import sys

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK_TF = 157.3

# (label, layout, transposed, B, Cin, Cout, input spatial, kernel, stride, calls per step)
LAYERS = [
    ("enc s0 conv0 1->33", "ndhwc", False, 2, 1, 33, (64, 224, 192), (1, 3, 3), (1, 1, 1), 1),
    ("enc s0 conv1 33->33", "ndhwc", False, 2, 33, 33, (64, 224, 192), (1, 3, 3), (1, 1, 1), 1),
    ("dec s0 conv0 66->33", "ndhwc", False, 2, 66, 33, (64, 224, 192), (1, 3, 3), (1, 1, 1), 1),
    ("dec s0 conv1 33->33", "ndhwc", False, 2, 33, 33, (64, 224, 192), (1, 3, 3), (1, 1, 1), 1),
    ("enc s1 conv0 33->66 /(1,2,2)", "ndhwc", False, 2, 33, 66, (64, 224, 192), (3, 3, 3), (1, 2, 2), 1),
    ("enc s1 conv1 66->66", "ndhwc", False, 2, 66, 66, (64, 112, 96), (3, 3, 3), (1, 1, 1), 1),
    ("dec s1 conv0 132->66", "ndhwc", False, 2, 132, 66, (64, 112, 96), (3, 3, 3), (1, 1, 1), 1),
    ("dec s1 conv1 66->66", "ndhwc", False, 2, 66, 66, (64, 112, 96), (3, 3, 3), (1, 1, 1), 1),
    ("enc s2 conv 66->132 /2", "ndhwc", False, 2, 66, 132, (64, 112, 96), (3, 3, 3), (2, 2, 2), 1),
    ("dec s2 conv 264->132", "ncdhw", False, 2, 264, 132, (32, 56, 48), (3, 3, 3), (1, 1, 1), 1),
    ("enc s3 conv 132->264 /2", "ncdhw", False, 2, 132, 264, (32, 56, 48), (3, 3, 3), (2, 2, 2), 1),
    ("dec s3 conv 528->264", "ncdhw", False, 2, 528, 264, (16, 28, 24), (3, 3, 3), (1, 1, 1), 1),
    ("enc s4 conv 264->324 /2", "ncdhw", False, 2, 264, 324, (16, 28, 24), (3, 3, 3), (2, 2, 2), 1),
    ("dec s4 conv 648->324", "ncdhw", False, 2, 648, 324, (8, 14, 12), (3, 3, 3), (1, 1, 1), 1),
    ("enc s5 conv 324->324 /2", "ncdhw", False, 2, 324, 324, (8, 14, 12), (3, 3, 3), (2, 2, 2), 1),
    ("up s1->s0 66->33 T(1,2,2)", "ndhwc", True, 2, 66, 33, (64, 112, 96), (1, 2, 2), (1, 2, 2), 1),
    ("up s2->s1 132->66 T2", "ndhwc", True, 2, 132, 66, (32, 56, 48), (2, 2, 2), (2, 2, 2), 1),
    ("up s3->s2 264->132 T2", "ncdhw", True, 2, 264, 132, (16, 28, 24), (2, 2, 2), (2, 2, 2), 1),
    ("head s0 1x1 33->14", "ndhwc", False, 2, 33, 14, (64, 224, 192), (1, 1, 1), (1, 1, 1), 1),
    ("FFN s2 1x1 132->528", "ncdhw", False, 2, 132, 528, (32, 56, 48), (1, 1, 1), (1, 1, 1), 4),
    ("FFN s2 1x1 528->132", "ncdhw", False, 2, 528, 132, (32, 56, 48), (1, 1, 1), (1, 1, 1), 4),
    ("Swin s2 fc1 1x1 132->132 (windows)", "ncdhw", False, 1024, 132, 132, (4, 7, 6), (1, 1, 1), (1, 1, 1), 2),
    ("Swin s2 grouped 1x1 264->264 g6", "ncdhw6", False, 1024, 264, 264, (168, 1, 1), (1, 1, 1), (1, 1, 1), 2),
    ("Swin s2 fc2 1x1 264->132 (windows)", "ncdhw", False, 1024, 264, 132, (4, 7, 6), (1, 1, 1), (1, 1, 1), 2),
]

# padded twins: (label of the layer, padded Cin, padded Cout)
PADS = {
    "enc s0 conv0 1->33": [(1, 40), (4, 40), (8, 40)],
    "enc s2 conv 66->132 /2": [(72, 132), (72, 136)],
    "up s1->s0 66->33 T(1,2,2)": [(72, 40), (80, 40)],
    "up s2->s1 132->66 T2": [(132, 72), (136, 72)],
    "head s0 1x1 33->14": [(40, 14), (40, 16)],
    "enc s0 conv1 33->33": [(40, 40), (48, 48), (64, 64)],
    "dec s0 conv0 66->33": [(80, 40), (96, 48), (128, 64)],
    "enc s1 conv1 66->66": [(72, 72), (80, 80), (96, 96), (128, 128)],
    "dec s1 conv0 132->66": [(144, 72), (160, 80), (192, 96)],
    "enc s1 conv0 33->66 /(1,2,2)": [(40, 80), (48, 96), (64, 128)],
}


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


def run_layer(label, layout, transposed, B, ci, co, sp, k, stride, iters, once):
    dev = torch.device("cuda:0")
    groups = 6 if layout == "ncdhw6" else 1
    mf = torch.channels_last_3d if layout == "ndhwc" else torch.contiguous_format
    pad = tuple(i // 2 for i in k) if not transposed else (0, 0, 0)
    x = torch.randn((B, ci) + sp, device=dev).contiguous(memory_format=mf)
    if transposed:
        w = (torch.randn((ci, co) + k, device=dev) * 0.05).contiguous(memory_format=mf)
        conv = lambda a, ww: F.conv_transpose3d(a, ww, None, stride=stride)   # noqa: E731
        out_sp = tuple(s * st for s, st in zip(sp, stride))
    else:
        w = (torch.randn((co, ci // groups) + k, device=dev) * 0.05).contiguous(memory_format=mf)
        conv = lambda a, ww: F.conv3d(a, ww, None, stride=stride, padding=pad, groups=groups)   # noqa: E731
        out_sp = tuple((s + 2 * p - kk) // st + 1 for s, p, kk, st in zip(sp, pad, k, stride))
    gy = torch.randn((B, co) + out_sp, device=dev).contiguous(memory_format=mf)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    n_out = B * out_sp[0] * out_sp[1] * out_sp[2]
    n_in = B * sp[0] * sp[1] * sp[2]
    taps = k[0] * k[1] * k[2]
    flops = 2.0 * (n_in if transposed else n_out) * (ci // groups) * co * taps
    it = 1 if once else iters
    if once:
        y1 = conv(xg, w)
        torch.autograd.grad(y1, xg, gy, retain_graph=False)
        y2 = conv(x, wg)
        torch.autograd.grad(y2, wg, gy)
        torch.cuda.synchronize()
        return flops, 0.0, 0.0, 0.0
    t_f = timeit(lambda: conv(x, w), it)
    y1 = conv(xg, w)
    t_d = timeit(lambda: torch.autograd.grad(y1, xg, gy, retain_graph=True), it) if ci > 1 else float("nan")
    y2 = conv(x, wg)
    t_w = timeit(lambda: torch.autograd.grad(y2, wg, gy, retain_graph=True), it)
    del x, w, gy, xg, wg, y1, y2
    torch.cuda.empty_cache()
    return flops, t_f, t_d, t_w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--pad", action="store_true", help="also time the zero-padded channel counts")
    ap.add_argument("--once", action="store_true", help="one launch per direction (for rocprofv3 --pmc runs)")
    ap.add_argument("--md", default=None)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    lines = ["| layer | layout | GFLOP (one direction) | fwd ms | TF/s | % of 157.3 | dgrad ms | TF/s | % | wgrad ms | TF/s | % | calls/step |",
             "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    tot = [0.0, 0.0, 0.0, 0.0]

    def row(label, layout, transposed, B, ci, co, sp, k, stride, calls, count=True):
        flops, tf, td, tw = run_layer(label, layout, transposed, B, ci, co, sp, k, stride, args.iters, args.once)
        if args.once:
            return

        def cell(t):
            if t != t or t <= 0:
                return "— | — | —"
            tfs = flops / (t * 1e-3) / 1e12
            return "%.3f | %.1f | %.1f %%" % (t, tfs, 100 * tfs / PEAK_TF)
        lines.append("| %s | %s | %.1f | %s | %s | %s | %d |" % (label, layout.upper()[:5], flops / 1e9, cell(tf), cell(td), cell(tw), calls))
        print(lines[-1], flush=True)
        if count:
            tot[0] += flops * calls * (3 if td == td else 2)
            tot[1] += calls * tf
            tot[2] += calls * (td if td == td else 0.0)
            tot[3] += calls * tw

    for (label, layout, transposed, B, ci, co, sp, k, stride, calls) in LAYERS:
        row(label, layout, transposed, B, ci, co, sp, k, stride, calls)
        if args.pad and label in PADS:
            for (pci, pco) in PADS[label]:
                row("  padded %d->%d" % (pci, pco), layout, transposed, B, pci, pco, sp, k, stride, calls, count=False)
    if not args.once:
        t_all = tot[1] + tot[2] + tot[3]
        lines.append("")
        lines.append("sum over the listed layers x calls/step: %.2f TFLOP, fwd %.1f ms + dgrad %.1f ms + wgrad %.1f ms = %.1f ms "
                     "-> %.1f TFLOP/s = %.1f %% of the fp32 MFMA peak"
                     % (tot[0] / 1e12, tot[1], tot[2], tot[3], t_all, tot[0] / (t_all * 1e-3) / 1e12,
                        100 * tot[0] / (t_all * 1e-3) / 1e12 / PEAK_TF))
        print(lines[-1])
        if args.md:
            with open(args.md, "w") as f:
                f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
