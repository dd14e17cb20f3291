#!/usr/bin/env python
"""Tune MIOpen's solvers for the convolution problems of the cfg-2 step and keep the result as a user database that
travels with the repository (nextou_amd/miopen_db/): `MIOPEN_FIND_ENFORCE=SEARCH` makes every find call explore the
tunable solvers' configurations (CK instance choice) for the problem and write the best into the user perf-db /
find-db under MIOPEN_USER_DB_PATH.  bench.py points MIOPEN_USER_DB_PATH at the committed directory, which (a) replaces
the ~3 min find phase of a fresh box by database look-ups and (b) uses the tuned instances.

    MIOPEN_USER_DB_PATH=<dir> python tools/miopen_tune.py [--budget-s 1500] [--only s0]

The convolutions stay on PyTorch-ROCm / MIOpen (BASELINE.json north_star); this is library configuration, not a kernel.
"""
import argparse
import os
import sys
import time

os.environ.setdefault("MIOPEN_FIND_ENFORCE", "SEARCH")
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# (label, transposed, B, Cin, Cout, input spatial, kernel, stride) — channels-last, the padded channel counts of the model
PROBLEMS = [
    ("s0 conv1 40->40", False, 2, 40, 40, (64, 224, 192), (1, 3, 3), (1, 1, 1)),
    ("s0 dec conv0 80->40", False, 2, 80, 40, (64, 224, 192), (1, 3, 3), (1, 1, 1)),
    ("s1 conv1 72->72", False, 2, 72, 72, (64, 112, 96), (3, 3, 3), (1, 1, 1)),
    ("s1 dec conv0 144->72", False, 2, 144, 72, (64, 112, 96), (3, 3, 3), (1, 1, 1)),
    ("s1 conv0 40->72 /(1,2,2)", False, 2, 40, 72, (64, 224, 192), (3, 3, 3), (1, 2, 2)),
    ("s0 conv0 4->40", False, 2, 4, 40, (64, 224, 192), (1, 3, 3), (1, 1, 1)),
    ("s2 conv 72->132 /2", False, 2, 72, 132, (64, 112, 96), (3, 3, 3), (2, 2, 2)),
    ("s2 dec conv 264->132", False, 2, 264, 132, (32, 56, 48), (3, 3, 3), (1, 1, 1)),
    ("up s1->s0 72->40", True, 2, 72, 40, (64, 112, 96), (1, 2, 2), (1, 2, 2)),
    ("up s2->s1 132->72", True, 2, 132, 72, (32, 56, 48), (2, 2, 2), (2, 2, 2)),
    ("head s0 40->14", False, 2, 40, 14, (64, 224, 192), (1, 1, 1), (1, 1, 1)),
    ("FFN s2 132->528", False, 2, 132, 528, (32, 56, 48), (1, 1, 1), (1, 1, 1)),
    ("FFN s2 528->132", False, 2, 528, 132, (32, 56, 48), (1, 1, 1), (1, 1, 1)),
    ("s3 conv 132->264 /2", False, 2, 132, 264, (32, 56, 48), (3, 3, 3), (2, 2, 2)),
    ("s3 dec conv 528->264", False, 2, 528, 264, (16, 28, 24), (3, 3, 3), (1, 1, 1)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget-s", type=float, default=1500.0)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    print("MIOPEN_USER_DB_PATH=%s  MIOPEN_FIND_ENFORCE=%s" % (os.environ.get("MIOPEN_USER_DB_PATH"), os.environ["MIOPEN_FIND_ENFORCE"]))
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    t_start = time.time()
    for label, transposed, B, ci, co, sp, k, stride in PROBLEMS:
        if args.only and args.only not in label:
            continue
        if time.time() - t_start > args.budget_s:
            print("budget exhausted before", label)
            break
        mf = torch.channels_last_3d
        x = torch.randn((B, ci) + sp, device=dev).contiguous(memory_format=mf).requires_grad_(ci > 4)
        if transposed:
            w = (torch.randn((ci, co) + k, device=dev) * 0.05).requires_grad_(True)
            conv = lambda: F.conv_transpose3d(x, w, None, stride=stride)   # noqa: E731
        else:
            w = (torch.randn((co, ci) + k, device=dev) * 0.05).requires_grad_(True)
            pad = tuple(i // 2 for i in k)
            conv = lambda: F.conv3d(x, w, None, stride=stride, padding=pad)   # noqa: E731
        t0 = time.time()
        y = conv()
        gy = torch.randn_like(y)
        torch.autograd.grad(y, [t for t in (x, w) if t.requires_grad], gy)
        torch.cuda.synchronize()
        t1 = time.time()
        # steady-state timing after tuning
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            y = conv()
            torch.autograd.grad(y, [t for t in (x, w) if t.requires_grad], gy)
        e1.record()
        torch.cuda.synchronize()
        print("%-28s tuned in %6.1f s; fwd+bwd %.3f ms" % (label, t1 - t0, e0.elapsed_time(e1) / 3), flush=True)
        del x, w, y, gy
        torch.cuda.empty_cache()
    d = os.environ.get("MIOPEN_USER_DB_PATH")
    if d and os.path.isdir(d):
        for f in sorted(os.listdir(d)):
            print("  db file %s %d bytes" % (f, os.path.getsize(os.path.join(d, f))))


if __name__ == "__main__":
    main()
