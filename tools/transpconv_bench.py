#!/usr/bin/env python
"""Experiment: nnU-Net's up-sampling ConvTranspose3d (kernel == stride) as MIOpen runs it vs the same
operator written as one GEMM + depth-to-space permutation.  cfg-2 decoder shapes, batch 2, fp32."""
import os
import sys
import time

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextou_amd.network_architecture.upsample import transposed_conv_as_gemm  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
# (Cin, Cout, input spatial, stride) of decoder stages 0..4 for cfg 2
LAYERS = [(324, 324, (4, 7, 6), (2, 2, 2)), (324, 264, (8, 14, 12), (2, 2, 2)), (264, 132, (16, 28, 24), (2, 2, 2)),
          (132, 66, (32, 56, 48), (2, 2, 2)), (66, 33, (64, 112, 96), (1, 2, 2))]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


tot_a = tot_b = 0.0
for cin, cout, sp, st in LAYERS:
    x = torch.randn((2, cin) + sp, device=dev, requires_grad=True)
    w = (torch.randn((cin, cout) + st, device=dev) * 0.05).requires_grad_(True)
    b = torch.zeros(cout, device=dev, requires_grad=True)
    g = torch.randn((2, cout) + tuple(s * k for s, k in zip(sp, st)), device=dev)

    def ref():
        y = F.conv_transpose3d(x, w, b, stride=st)
        torch.autograd.grad(y, [x, w, b], g)

    def mine():
        y = transposed_conv_as_gemm(x, w, b, st)
        torch.autograd.grad(y, [x, w, b], g)

    ya, yb = F.conv_transpose3d(x, w, b, stride=st), transposed_conv_as_gemm(x, w, b, st)
    ga, gb = torch.autograd.grad(ya, [x, w, b], g), torch.autograd.grad(yb, [x, w, b], g)
    err = max(float((ya - yb).abs().max()), *[float((p - q).abs().max() / (p.abs().max() + 1e-12)) for p, q in zip(ga, gb)])
    ta, tb = timeit(ref), timeit(mine)
    tot_a += ta
    tot_b += tb
    print("Cin %3d Cout %3d in %-14s stride %s: MIOpen fwd+bwd %7.3f ms   GEMM form %7.3f ms   max err %.2e" %
          (cin, cout, sp, st, ta, tb, err), flush=True)
print("total: MIOpen %.2f ms, GEMM form %.2f ms" % (tot_a, tot_b))
