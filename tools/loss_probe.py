#!/usr/bin/env python
"""Time of the harness' Dice + CE deep-supervision loss (forward + backward) on cfg-2 logits, channels-last as the network emits them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nextou_amd.harness import downsample_targets, synthetic_batch
dev = torch.device("cuda:0")
trainer, cfg, batch, classes = bench.build_trainer("cfg2", dev, False)
bench.move_to(trainer, dev)
data, target = synthetic_batch(cfg, 1, classes, batch, dev, seed=1)
shapes = bench._head_shapes(cfg)
targets = downsample_targets(target, shapes)
logits = [torch.randn((batch, classes) + tuple(s.shape[2:]), device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True) for s in shapes]
def step():
    loss = trainer.loss(logits, targets)
    return torch.autograd.grad(loss, logits, allow_unused=True)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g = step()
e1.record(); torch.cuda.synchronize()
print("Dice + CE deep-supervision loss, forward + backward: %.3f ms per step" % (e0.elapsed_time(e1) / 10))
print("logit elements:", sum(l.numel() for l in logits), "grad layouts:", [("cl" if x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous() else "nc") for x in g if x is not None])
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
