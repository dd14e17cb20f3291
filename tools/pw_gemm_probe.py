#!/usr/bin/env python
"""1x1 convolutions of the GNN stages on channels-last activations: MIOpen's convolution path (what nn.Conv3d takes; 3-D
and the depth-flat 2-D view) against the same three products as plain library GEMMs on the (points, channels) matrix —
forward x @ W^T, data gradient gy @ W, weight gradient gy^T @ x (reduction over 172 032 points: a split-K shape) — and
against K7, the own f32-MFMA kernels (csrc/pw_gemm.hip).

    python tools/pw_gemm_probe.py [--iters 10] [--own-only]

Informational; decided that the pointwise convolutions go through K7 and not torch.mm.
"""
import argparse
import os
import sys

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from nextou_amd import graph_ops  # noqa: E402

# (label, B, Cin, Cout, spatial, groups)
SHAPES = [("FFN s2 132->528", 2, 132, 528, (32, 56, 48), 1), ("FFN s2 528->132", 2, 528, 132, (32, 56, 48), 1),
          ("fc s2 132->132", 2, 132, 132, (32, 56, 48), 1), ("fc2 s2 264->132", 2, 264, 132, (32, 56, 48), 1),
          ("gconv s2 264->264 g6", 2, 264, 264, (32, 56, 48), 6),
          ("FFN s3 264->1056", 2, 264, 1056, (16, 28, 24), 1), ("FFN s3 1056->264", 2, 1056, 264, (16, 28, 24), 1),
          ("fc s3 264->264", 2, 264, 264, (16, 28, 24), 1), ("fc2 s3 528->264", 2, 528, 264, (16, 28, 24), 1),
          ("gconv s3 528->528 g6", 2, 528, 528, (16, 28, 24), 6),
          ("FFN s4 324->1296", 2, 324, 1296, (8, 14, 12), 1), ("FFN s4 1296->324", 2, 1296, 324, (8, 14, 12), 1),
          ("fc2 s4 648->324", 2, 648, 324, (8, 14, 12), 1), ("FFN s5 324->1296", 2, 324, 1296, (4, 7, 6), 1),
          ("head s0 40->14", 2, 40, 14, (64, 224, 192), 1), ("head s1 72->14", 2, 72, 14, (64, 112, 96), 1)]


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
    ap.add_argument("--own-only", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    hip = graph_ops._HIP
    print("| layer | GFLOP | conv3d fwd / dgrad / wgrad ms | conv2d (flat) fwd / dgrad / wgrad ms | mm fwd / dgrad / wgrad ms | "
          "K7 fwd / dgrad / wgrad ms | K7 TF/s fwd / dgrad / wgrad | K7 max rel err fwd / dgrad / wgrad |")
    print("|---|---:|---|---|---|---|---|---|")
    tot = {"conv2d": 0.0, "k7": 0.0}
    own = []
    for label, B, ci, co, sp, g in SHAPES:
        if args.only and args.only not in label:
            continue
        P = B * sp[0] * sp[1] * sp[2]
        x = torch.randn((B, ci) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn((co, ci // g, 1, 1, 1), device=dev) * 0.05
        gy = torch.randn((B, co) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
        fl = 2.0 * P * ci * co / g
        cols = []
        if not args.own_only:
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            cf = timeit(lambda: F.conv3d(x, w, groups=g), args.iters)
            y1 = F.conv3d(xg, w, groups=g)
            cd = timeit(lambda: torch.autograd.grad(y1, xg, gy, retain_graph=True), args.iters)
            y2 = F.conv3d(x, wg, groups=g)
            cw = timeit(lambda: torch.autograd.grad(y2, wg, gy, retain_graph=True), args.iters)
            cols.append("%.3f / %.3f / %.3f" % (cf, cd, cw))
            del y1, y2
            xf, gf, wf = graph_ops.flat_depth(x), graph_ops.flat_depth(gy), w.squeeze(2)
            xfg, wfg = xf.detach().clone(memory_format=torch.preserve_format).requires_grad_(True), wf.clone().requires_grad_(True)
            ff = timeit(lambda: F.conv2d(xf, wf, groups=g), args.iters)
            y1 = F.conv2d(xfg, wf, groups=g)
            fd = timeit(lambda: torch.autograd.grad(y1, xfg, gf, retain_graph=True), args.iters)
            y2 = F.conv2d(xf, wfg, groups=g)
            fw = timeit(lambda: torch.autograd.grad(y2, wfg, gf, retain_graph=True), args.iters)
            cols.append("%.3f / %.3f / %.3f" % (ff, fd, fw))
            if (co // g) % 4 == 0 and (ci // g) % 4 == 0:
                tot["conv2d"] += ff + fd + fw
            del y1, y2, xfg, wfg, xg, wg
            if g == 1:
                x2 = x.permute(0, 2, 3, 4, 1).reshape(P, ci)
                g2 = gy.permute(0, 2, 3, 4, 1).reshape(P, co)
                w2 = w.reshape(co, ci)
                mf = timeit(lambda: torch.mm(x2, w2.t()), args.iters)
                md = timeit(lambda: torch.mm(g2, w2), args.iters)
                mw = timeit(lambda: torch.mm(g2.t(), x2), args.iters)
                cols.append("%.3f / %.3f / %.3f" % (mf, md, mw))
            else:
                cols.append("—")
        else:
            cols += ["—", "—", "—"]
        if (co // g) % 4 == 0 and (ci // g) % 4 == 0:
            w2 = w.reshape(co, ci // g).contiguous()
            wt = w2.reshape(g, co // g, ci // g).transpose(1, 2).reshape(ci, co // g).contiguous()
            kf = timeit(lambda: hip.pw_rows(x, w2, None, g), args.iters)
            kd = timeit(lambda: hip.pw_rows(gy, wt, None, g), args.iters)
            kw = timeit(lambda: hip.pw_wgrad(gy, x, g), args.iters)
            # references in float64 on a row subset (forward / data gradient) and on everything (weight gradient)
            rows = torch.arange(0, P, max(1, P // 4096), device=dev)
            x2 = x.permute(0, 2, 3, 4, 1).reshape(P, g, ci // g)
            g2 = gy.permute(0, 2, 3, 4, 1).reshape(P, g, co // g)
            w3 = w.reshape(g, co // g, ci // g).double()
            yf = hip.pw_rows(x, w2, None, g).permute(0, 2, 3, 4, 1).reshape(P, g, co // g)[rows]
            ref = torch.einsum("pgk,gnk->pgn", x2[rows].double(), w3)
            ef = float((yf - ref).abs().max() / ref.abs().max())
            yd = hip.pw_rows(gy, wt, None, g).permute(0, 2, 3, 4, 1).reshape(P, g, ci // g)[rows]
            ref = torch.einsum("pgn,gnk->pgk", g2[rows].double(), w3)
            ed = float((yd - ref).abs().max() / ref.abs().max())
            dw = hip.pw_wgrad(gy, x, g).reshape(g, co // g, ci // g)
            ref = torch.zeros_like(dw, dtype=torch.float64)
            for lo in range(0, P, 16384):
                ref += torch.einsum("pgn,pgk->gnk", g2[lo:lo + 16384].double(), x2[lo:lo + 16384].double())
            ew = float((dw - ref).abs().max() / ref.abs().max())
            cols.append("%.3f / %.3f / %.3f" % (kf, kd, kw))
            own.append((label, own_report(lambda: (hip.pw_rows(x, w2, None, g), hip.pw_rows(gy, wt, None, g), hip.pw_wgrad(gy, x, g)))))
            cols.append("%.0f / %.0f / %.0f" % (fl / kf / 1e9, fl / kd / 1e9, fl / kw / 1e9))
            cols.append("%.1e / %.1e / %.1e" % (ef, ed, ew))
            tot["k7"] += kf + kd + kw
        else:
            cols += ["n/a (channels not multiples of 4)", "", ""]
        print("| %s | %.1f | %s |" % (label, fl / 1e9, " | ".join(cols)), flush=True)
        del x, w, gy
        torch.cuda.empty_cache()
    print("\nK7 launches (library profiler, HIP events on the launch stream):\n| layer | kernel | us | TFLOP/s (or TB/s) |\n|---|---|---:|---:|")
    for label, rows in own:
        for k, us, rate in rows:
            print("| %s | `%s` | %.1f | %.1f |" % (label, k, us, rate))
    print("\nsum over the K7-eligible layers above (one call each): conv2d-flat route %.3f ms, K7 %.3f ms"
          % (tot["conv2d"], tot["k7"]))


def own_report(fn, reps=5):
    """per-kernel mean us of the library's own launches inside fn (nextou_profile_enable / _report)."""
    import ctypes
    import json
    from nextou_amd import _lib
    L = _lib.lib()
    fn()
    torch.cuda.synchronize()
    L.nextou_profile_enable(64 * reps)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.nextou_profile_report(buf, len(buf))
    L.nextou_profile_enable(0)
    return [(r["kernel"], 1e3 * r["ms"] / r["launches"], r["work"] / r["launches"] / (r["ms"] / r["launches"] * 1e-3) / 1e12)
            for r in json.loads(buf.value[:n].decode())]


if __name__ == "__main__":
    main()
