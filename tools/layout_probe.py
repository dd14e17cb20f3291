#!/usr/bin/env python
"""Would the full-resolution stage (stage 0 of cfg 2, 33 channels at 64x224x192) be faster in channels_last_3d?

Times forward + backward of the stage-0 slice of the U-Net (encoder convs, strided conv into stage 1, transposed conv
back, skip concat, decoder convs, segmentation head, CE loss) on PyTorch-ROCm with every tensor NCDHW, and with the
stage-0 tensors NDHWC (converted at the stage-1 boundary).  LeakyReLU stands in for the norm (K6 runs at the HBM roofline
in either layout).  Informational tool for DESIGN.md.
"""
import os
import sys

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402


class Slice(nn.Module):
    def __init__(self, cl, stage1_cl=False):
        super().__init__()
        self.cl, self.stage1_cl = cl, stage1_cl
        self.c1 = nn.Conv3d(1, 33, (1, 3, 3), padding=(0, 1, 1))
        self.c2 = nn.Conv3d(33, 33, (1, 3, 3), padding=(0, 1, 1))
        self.down = nn.Conv3d(33, 66, 3, stride=(1, 2, 2), padding=1)
        self.s1 = nn.Conv3d(66, 66, 3, padding=1)
        self.up = nn.ConvTranspose3d(66, 33, (1, 2, 2), stride=(1, 2, 2))
        self.d1 = nn.Conv3d(66, 33, (1, 3, 3), padding=(0, 1, 1))
        self.d2 = nn.Conv3d(33, 33, (1, 3, 3), padding=(0, 1, 1))
        self.seg = nn.Conv3d(33, 14, 1)

    def forward(self, x, target):
        CL = torch.channels_last_3d
        act = lambda t: F.leaky_relu(t, 0.01)
        if self.cl:
            x = x.contiguous(memory_format=CL)
        skip = act(self.c2(act(self.c1(x))))
        t = act(self.down(skip))
        if self.cl and not self.stage1_cl:
            t = t.contiguous()
        t = act(self.s1(t))
        if self.cl:
            t = t.contiguous(memory_format=CL)
        u = self.up(t)
        y = act(self.d2(act(self.d1(torch.cat((u, skip), 1)))))
        return F.cross_entropy(self.seg(y), target)


def main():
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    x = torch.randn(2, 1, 64, 224, 192, device=dev)
    target = torch.randint(0, 14, (2, 64, 224, 192), device=dev)
    for name, cl, s1 in (("NCDHW", False, False), ("stage 0 NDHWC", True, False), ("stages 0+1 NDHWC", True, True)):
        torch.manual_seed(0)
        m = Slice(cl, s1).to(dev)
        if cl:
            m = m.to(memory_format=torch.channels_last_3d) if s1 else m
        def step():
            m.zero_grad(set_to_none=True)
            m(x, target).backward()
        for _ in range(3):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        print("%-20s fwd+bwd %.2f ms" % (name, e0.elapsed_time(e1) / 5))
        if len(sys.argv) > 1:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                step()
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))


if __name__ == "__main__":
    main()
