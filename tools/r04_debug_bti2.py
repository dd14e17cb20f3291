import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch
import bench
from nextou_amd.harness import GraphedTrainStep, downsample_targets, synthetic_batch
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
trainer, cfg, batch, classes = bench.build_trainer("cfg4", dev, False)
bench.move_to(trainer, dev)
data, target = synthetic_batch(cfg, 1, classes, batch, dev, seed=1234, blob_labels=True)
targets = downsample_targets(target, bench._head_shapes(cfg))
print([(float(t.min()), float(t.max()), t.dtype, tuple(t.shape)) for t in targets], flush=True)
step = bench.make_step(trainer, data, targets, None)
ti = trainer.loss.loss.ti
for _ in range(2):
    step()
print("eager ok", flush=True)
g = GraphedTrainStep(step, warmup=1, network=trainer.network, loss=trainer.loss)
torch.cuda.synchronize(); print("after capture:", int(ti._bad_targets), flush=True)
for i in range(2):
    l = g(); torch.cuda.synchronize(); print("replay", i, float(l), int(ti._bad_targets), flush=True)
l = step(); torch.cuda.synchronize(); print("eager deferred:", float(l), int(ti._bad_targets), flush=True)
from nextou_amd import _lib
_lib.lib().nextou_profile_enable(4096)
l = step(); torch.cuda.synchronize(); print("eager deferred + profiler:", float(l), int(ti._bad_targets), flush=True)
