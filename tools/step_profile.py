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
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    shapes = "--shapes" in sys.argv      # group the data-movement operators by input shape (layout conversions)
    workload = args[0] if args else "cfg2"
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=shapes) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
    if shapes:
        movers = ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::add", "aten::add_",
                  "aten::_to_copy", "aten::fill_", "aten::zero_")
        rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in movers]
        rows.sort(key=lambda e: -e.self_device_time_total)
        print("\ndata-movement operators by input shape (2 steps), self GPU time:")
        for e in rows[:40]:
            print("%-18s %5d calls %9.3f ms  %s" % (e.key, e.count, e.self_device_time_total / 1e3, e.input_shapes))


if __name__ == "__main__":
    main()
