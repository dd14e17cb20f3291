#!/usr/bin/env python
"""The Pool MRConv's grouped 1x1 convolution on the channel-major (B, 2C, N, 1, 1) tensor (reference torch_nn.py:66-92 inside
NexToU_Encoder_Decoder.py:401-418): MIOpen's grouped convolution (what runs today) against the batched BLAS GEMM over the
(B, groups, C/g, N) view of the same memory, forward and forward + backward: eager wall clock and the sum of the GPU kernel times (torch.profiler).

    python tools/pool_basicconv_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tools.gnn_stage_profile import timeit, timeit_graph  # noqa: E402

SHAPES = [("s2 Pool", 2, 264, 10752), ("s3 Pool", 2, 528, 10752), ("s4 Pool", 2, 648, 1344), ("s5 Pool", 2, 648, 168)]
G = 6


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    print("%-8s %-22s %10s %12s %12s %14s" % ("call", "path", "fwd us", "fwd+bwd us", "kernels fwd", "kernels f+b"), flush=True)
    for label, B, C2, N in SHAPES:
        x = torch.randn(B, C2, N, 1, 1, device=dev, requires_grad=True)
        w = (torch.randn(C2, C2 // G, 1, 1, 1, device=dev) * 0.1).requires_grad_(True)
        gy = torch.randn(B, C2, N, 1, 1, device=dev)

        def conv():
            return F.conv3d(x, w, None, groups=G)

        def bmm():
            y = torch.matmul(w.view(G, C2 // G, C2 // G), x.view(B, G, C2 // G, N))
            return y.view(B, C2, N, 1, 1)

        a, b = conv(), bmm()
        err = float((a - b).detach().abs().max() / a.detach().abs().max())
        for name, fn in (("miopen grouped conv3d", conv), ("batched gemm (matmul)", bmm)):
            def fwd():
                with torch.no_grad():
                    return fn()

            def fb():
                torch.autograd.grad(fn(), [x, w], gy)
            from torch.profiler import ProfilerActivity, profile
            tf, tfb = timeit(fwd, 20), timeit(fb, 20)
            ksum = []
            for f in (fwd, fb):
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for _ in range(5):
                        f()
                    torch.cuda.synchronize()
                ksum.append(sum(e.self_device_time_total for e in prof.key_averages()) / 5.0)
            print("%-8s %-22s %10.1f %12.1f %12.1f %14.1f" % (label, name, tf, tfb, ksum[0], ksum[1]), flush=True)
        print("         relative difference of the two forwards: %.2e" % err)


if __name__ == "__main__":
    main()
