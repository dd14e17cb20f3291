#!/usr/bin/env python
"""Which host-side ops launch the non-convolution, non-own kernels of the cfg-2 step?  rocprofv3 names the kernels
(`elementwise_kernel_manual_unroll<direct_copy>` ...) but not who asked for them; torch.profiler ties each device kernel
to the ATen op (and its input shapes) that launched it.

    python tools/aten_glue_profile.py [--workload cfg2] [--top 40]

Informational: finds strided copies / fills that a layout or caching change can remove.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from nextou_amd.harness import downsample_targets, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--stacks", action="store_true", help="group by Python call stack instead of input shapes")
    ap.add_argument("--parents", action="store_true", help="with --ops: group the matching ops by input shapes AND the chain of enclosing "
                    "ops (aten::contiguous <- _NormActBackward ...), which names the caller even on the autograd thread")
    ap.add_argument("--ops", default=None, help="comma-separated substrings: list only the ops whose name contains one of them")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
    trainer, cfg, batch, classes = bench.build_trainer(args.workload, device, False)
    bench.move_to(trainer, device)
    data, target = synthetic_batch(cfg, 1, classes, batch, device, seed=1234)
    targets = downsample_targets(target, bench._head_shapes(cfg))
    step = bench.make_step(trainer, data, targets, None)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=not args.stacks or args.parents,
                 with_stack=args.stacks) as prof:
        step()
        torch.cuda.synchronize()
    if args.parents:
        want = (args.ops or "copy_,fill_,add_").split(",")
        groups = {}
        for e in prof.events():
            dev = getattr(e, "self_device_time_total", 0) or 0
            if dev <= 0 or not any(o in e.name for o in want):
                continue
            chain, p = [], e.cpu_parent
            while p is not None and len(chain) < 4:
                chain.append(p.name[:48])
                p = p.cpu_parent
            key = (e.name, str(e.input_shapes)[:110], " <- ".join(chain))
            g = groups.setdefault(key, [0.0, 0])
            g[0] += dev
            g[1] += 1
        print("| self device us | calls | op | input shapes | enclosing ops |\n|---:|---:|---|---|---|")
        for (name, shapes, chain), (dev, n) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:args.top]:
            print("| %.0f | %d | `%s` | %s | %s |" % (dev, n, name, shapes, chain))
        print("total: %.0f us over %d calls" % (sum(v[0] for v in groups.values()), sum(v[1] for v in groups.values())))
        return
    if args.stacks:
        avg = prof.key_averages(group_by_stack_n=6)
    else:
        avg = prof.key_averages(group_by_input_shape=True)
    rows = []
    for e in avg:
        dev = getattr(e, "self_device_time_total", None)
        if dev is None:
            dev = e.self_cuda_time_total
        if dev <= 0:
            continue
        rows.append((dev, e.count, e.key, str(getattr(e, "input_shapes", ""))[:150],
                     " <- ".join(s.split("/")[-1] for s in (e.stack or [])[:6]) if args.stacks else ""))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print("one cfg-2 step under torch.profiler: %.1f ms of device time attributed to %d (op, shape) groups" % (
        total / 1e3, len(rows)))
    print("| self device us | calls | op | input shapes / stack |")
    print("|---:|---:|---|---|")
    skip = ("convolution", "nextou", "Conv")
    shown = 0
    for dev, n, key, shapes, stack in rows:
        if any(s in key for s in skip) or key.startswith(("void ", "_ZN", "igemm", "hipLaunch", "Memcpy", "Memset", "SubTensor")):
            continue
        if args.ops and not any(o in key for o in args.ops.split(",")):
            continue
        print("| %.0f | %d | `%s` | %s %s |" % (dev, n, key[:70], shapes, stack))
        shown += 1
        if shown >= args.top:
            break
    by_op = {}
    for dev, n, key, _, _ in rows:
        by_op.setdefault(key, [0.0, 0])
        by_op[key][0] += dev
        by_op[key][1] += n
    print("\n| op (all shapes) | self device ms | calls |\n|---|---:|---:|")
    for key, (dev, n) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:30]:
        print("| `%s` | %.3f | %d |" % (key[:80], dev / 1e3, n))


if __name__ == "__main__":
    main()
