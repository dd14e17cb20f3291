import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nextou_amd.harness import GraphedTrainStep, config_3d_fullres_nextou, downsample_targets, synthetic_batch
from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_BTI_Synapse import nnUNetTrainer_NexToU_BTI_Synapse
dev = torch.device("cuda:0")
cfg = config_3d_fullres_nextou(patch_size=(32, 128, 128), base=6, max_features=48, batch_size=2)
torch.manual_seed(0)
tr = nnUNetTrainer_NexToU_BTI_Synapse(cfg, 14, device=dev, log=None).initialize()
data, target = synthetic_batch(cfg, 1, 14, 2, dev, blob_labels=True)
with torch.no_grad():
    outs = tr.network(data)
tg = downsample_targets(target, outs)
print([ (float(t.min()), float(t.max()), t.dtype) for t in tg])
def step():
    return tr.train_step(data, tg)
ti = tr.loss.loss.ti
step(); print("eager host-validated ok")
ti.validate_targets = "deferred"
step(); torch.cuda.synchronize(); print("after 1 eager deferred:", int(ti._bad_targets))
ti.validate_targets = True
g = GraphedTrainStep(step, warmup=1, network=tr.network, loss=tr.loss)
torch.cuda.synchronize(); print("after capture:", int(ti._bad_targets))
for i in range(3):
    l = g(); torch.cuda.synchronize(); print("replay", i, float(l), int(ti._bad_targets))
step(); torch.cuda.synchronize(); print("eager after:", int(ti._bad_targets))
