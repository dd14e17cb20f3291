#!/usr/bin/env python
"""ATen-operator view of one cfg-2 train step (torch.profiler, GPU time per operator) — shows how the
step divides between the dense stages' operators and this package's own module code."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (sets the MIOpen env switches)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from nextou_amd.harness import downsample_targets, synthetic_batch  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    trainer, cfg, batch, classes = bench.build_trainer(workload, dev, False)
    bench.move_to(trainer, dev)
    data, target = synthetic_batch(cfg, 1, classes, batch, dev)
    targets = downsample_targets(target, bench._head_shapes(cfg))
    step = bench.make_step(trainer, data, targets, None)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))


if __name__ == "__main__":
    main()
