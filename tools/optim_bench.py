#!/usr/bin/env python
"""clip_grad_norm_(12) + SGD(nesterov) over the parameter tensors of the cfg-2 network, alone: torch's multi-tensor kernels
(foreach norm / mul + fused SGD) against nextou_amd.optim.ClipSGD, eager launches and replayed hipGraphs, HIP-event times.

    python tools/optim_bench.py [--workload cfg2] [--iters 20]
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from nextou_amd import _lib  # noqa: E402
from nextou_amd.optim import ClipSGD  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def graphed(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    trainer, cfg, batch, classes = bench.build_trainer(args.workload, device, False)
    trainer.network.to(device)
    params = [p for p in trainer.network.parameters() if p.requires_grad]
    gen = torch.Generator(device=device).manual_seed(1)
    for p in params:
        p.grad = torch.empty_like(p).normal_(generator=gen)
    n = sum(p.numel() for p in params)
    print("%d tensors, %.2f M elements" % (len(params), n / 1e6))
    kw = dict(weight_decay=trainer.weight_decay, momentum=trainer.momentum, nesterov=True)
    fused = torch.optim.SGD(params, 1e-6, fused=True, **kw)
    foreach = torch.optim.SGD(params, 1e-6, foreach=True, **kw)
    own = ClipSGD(params, 1e-6, **kw)

    def torch_step(opt):
        def f():
            torch.nn.utils.clip_grad_norm_(params, 12)
            opt.step()
        return f
    rows = [("torch: clip_grad_norm_ + SGD(fused)", torch_step(fused)), ("torch: clip_grad_norm_ + SGD(foreach)", torch_step(foreach)),
            ("torch: SGD(fused) alone", fused.step), ("ClipSGD.clip_and_step", lambda: own.clip_and_step(12)), ("ClipSGD.step (no clip)", own.step)]
    floor = {True: (4 + 24) * n, False: 20 * n}
    print("| step | eager us | replayed graph us | bytes / floor at 5.3 TB/s |")
    print("|---|---:|---:|---|")
    for name, fn in rows:
        e = timed(fn, args.iters)
        g = timed(graphed(fn), args.iters)
        by = floor["clip" in name]
        print("| %s | %.1f | %.1f | %.0f MB / %.0f us |" % (name, e, g, by / 1e6, by / 5.3e12 * 1e6))
    assert own.last_path == "own"
    from nextou_amd import optim
    for chunk in (4096, 65536, 16384):                 # elements per workgroup (the default last: the report below is its)
        optim._CHUNK = chunk
        o = ClipSGD(params, 1e-6, **kw)
        fn = lambda: o.clip_and_step(12)              # noqa: E731
        print("| ClipSGD.clip_and_step, %d-element chunks | %.1f | %.1f | |" % (chunk, timed(fn, args.iters), timed(graphed(fn), args.iters)))
    own = ClipSGD(params, 1e-6, **kw)
    L = _lib.lib()
    L.nextou_profile_enable(4096)
    for _ in range(5):
        own.clip_and_step(12)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 20)
    L.nextou_profile_report(buf, len(buf))
    L.nextou_profile_enable(0)
    for r in json.loads(buf.value.decode()):
        print("%-60s %3d launches  %8.1f us each  %7.1f GB/s" % (r["kernel"], r["launches"], r["ms"] / r["launches"] * 1e3,
                                                                r["work"] / r["launches"] / (r["ms"] / r["launches"] / 1e3) / 1e9))


if __name__ == "__main__":
    main()
