#!/usr/bin/env python
"""Sliding-window inference throughput of the cfg-2 network on one MI355X (informational; bench.py is the headline).

    python tools/infer_bench.py [--volume 128 448 384] [--mirror] [--batch 8] [--bf16]

voxels/s = volume voxels / wall time of predict_sliding_window (tile step 0.5, Gaussian weighting), synthetic
z-scored volume, random-init weights, eval-mode BatchNorm.
"""
import argparse
import json
import os
import sys
import time

for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    os.environ.setdefault(_k, "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextou_amd.harness import SimpleLabelManager, SimplePlansManager, config_3d_fullres_nextou  # noqa: E402
from nextou_amd.inference import compute_steps_for_sliding_window, predict_sliding_window  # noqa: E402
from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, nargs=3, default=[128, 448, 384])
    ap.add_argument("--mirror", action="store_true")
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--bf16", action="store_true")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    cfg = config_3d_fullres_nextou()
    torch.manual_seed(0)
    net = nnUNetTrainer_NexToU.build_network_architecture(SimplePlansManager(SimpleLabelManager(14)), {}, cfg, 1,
                                                          True).to(dev)
    image = torch.randn((1,) + tuple(args.volume), device=dev)
    axes = (0, 1, 2) if args.mirror else None
    tiles = 1
    for s in compute_steps_for_sliding_window(args.volume, cfg.patch_size, 0.5):
        tiles *= len(s)
    for bs in args.batch:
        kw = dict(tile_step_size=0.5, use_gaussian=True, mirror_axes=axes, batch_size=bs,
                  autocast_dtype=torch.bfloat16 if args.bf16 else None)
        predict_sliding_window(net, image, cfg.patch_size, **kw)   # warm-up: MIOpen find for every batch size used
        torch.cuda.synchronize()
        t0 = time.time()
        out = predict_sliding_window(net, image, cfg.patch_size, **kw)
        torch.cuda.synchronize()
        dt = time.time() - t0
        fw = tiles * (8 if args.mirror else 1)
        print(json.dumps({"volume": args.volume, "tiles": tiles, "forwards": fw, "batch": bs, "mirror": bool(args.mirror),
                          "dtype": "bf16-autocast" if args.bf16 else "f32", "seconds": round(dt, 3),
                          "ms_per_forward": round(1e3 * dt / fw, 2),
                          "voxels_per_s": round(image[0].numel() / dt, 1),
                          "patch_voxels_per_s": round(fw * float(torch.tensor(cfg.patch_size).prod()) / dt, 1),
                          "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
