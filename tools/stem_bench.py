#!/usr/bin/env python
"""K9 (csrc/stem_conv.hip) alone at the cfg-2 / cfg-5 image: launch times from the library's own HIP-event profiler.
    python tools/stem_bench.py [--cfg 2|5] [--iters 10]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nextou_amd import _lib, graph_ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None, help="write the launch-label rows here (tools/pmc_traffic_json.py --labels)")
    ap.add_argument("--two", action="store_true", help="also K6's two-gradient backward (the skip connection) at the stage-0 tensor")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shape = (2, 1, 64, 224, 192) if a.cfg == 2 else (2, 1, 96, 256, 256)
    C, cp = 33, 40
    x = torch.randn(shape, device=dev)
    w2 = torch.randn((C, 9), device=dev) * 0.3
    cb, gamma, beta = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gy = torch.randn((shape[0], cp) + shape[2:], device=dev).contiguous(memory_format=torch.channels_last_3d)
    H = graph_ops._HIP
    if a.two:
        xs = torch.randn((shape[0], cp) + shape[2:], device=dev).contiguous(memory_format=torch.channels_last_3d)
        g1 = torch.randn_like(xs)
        wide = torch.randn((shape[0], 2 * cp) + shape[2:], device=dev).contiguous(memory_format=torch.channels_last_3d)
        g2 = wide.narrow(1, cp, cp)
        gw, gb2 = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
        _, m2, i2 = H.norm_act_fwd(xs, gw, gb2, None, None, True, 0.1, 1e-5, 0.01, 0, None, channels_last=True)
    for it in range(a.iters + 2):
        if it == 2:
            torch.cuda.synchronize()
            _lib.lib().nextou_profile_enable(1024)
        y, mean, invstd, mom, act = H.stem_fwd(x, w2, cb, gamma, beta, rm, rv, True, 0.1, 1e-5, 0.01, cp)
        H.stem_bwd(x, gy, act, w2, gamma, mean, invstd, mom, 0.01, True, True, True)
        if a.two:
            H.norm_act_bwd_two(xs, g1, g2, gw, gb2, m2, i2, True, 0.01)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.lib().nextou_profile_report(buf, len(buf))
    rows = json.loads(buf.value.decode())
    if a.json:
        json.dump(rows, open(a.json, "w"))
    for r in rows:
        us = r["ms"] / r["launches"] * 1e3
        print("%-44s %8.1f us  %7.1f GB/s  %5.1f %% of 8 TB/s" % (r["kernel"], us, r["work"] / r["launches"] / us / 1e3, r["work"] / r["launches"] / us / 1e3 / 80))


if __name__ == "__main__":
    main()
