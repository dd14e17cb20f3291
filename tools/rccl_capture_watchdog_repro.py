#!/usr/bin/env python
"""Does an eager RCCL collective that is still on the process group's watchdog list break a hipGraph capture that follows it?  A
stand-alone probe (plain torch, nothing of this repository in the child processes) of one hypothesis for the rare
hipErrorCapturedEvent that terminates the averaged step.

PyTorch's NCCL (= RCCL) process group keeps every eager collective on a list that its watchdog thread polls every 100 ms with
hipEventQuery on the work's end event; finished work leaves the list at the next poll.  A collective issued inside a hipGraph capture
pulls the group's internal stream into the capture.  Hypothesis: if an eager work is still on the list at that moment, the watchdog's
next query of its end event — recorded BEFORE the capture, on a stream that is capturing NOW — fails with hipErrorCapturedEvent
("operation not permitted on an event last recorded in a capturing stream") and the watchdog terminates the process.

    python tools/rccl_capture_watchdog_repro.py [runs]     # parent: runs the child in both modes
    child modes:  immediate   eager all-reduce -> synchronize -> capture { all-reduce; host sleeps 0.3 s }
                  drained     eager all-reduce -> synchronize -> sleep 0.5 s -> the same capture

Result on MI355X / ROCm 7.2 / PyTorch 2.10 (profiles/r05_rccl_capture_watchdog_repro.txt): 3 / 3 clean in BOTH modes — this sequence
alone does not trigger the error; the hypothesis is not confirmed.  nextou_amd.harness.GraphedTrainStep keeps its 0.5-s pause between
the eager warm-up and the capture as a precaution that costs nothing.
"""
import os
import subprocess
import sys
import time


def child(mode: str) -> None:
    import socket

    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
    x = torch.ones(1 << 20, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            dist.all_reduce(x)                     # eager collectives: on the watchdog's list until its next poll after they finish
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if mode == "drained":
        time.sleep(0.5)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        dist.all_reduce(x)                         # RCCL's stream joins the capture here
        time.sleep(0.3)                            # ... and the capture stays open across several watchdog polls
    g.replay()
    torch.cuda.synchronize()
    time.sleep(0.3)
    print("CLEAN", mode, float(x[0]), flush=True)
    dist.destroy_process_group()


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] in ("immediate", "drained"):
        return child(sys.argv[1])
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for mode in ("immediate", "drained"):
        clean = 0
        last = ""
        for _ in range(runs):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, timeout=300)
            ok = p.returncode == 0 and "CLEAN" in p.stdout
            clean += ok
            if not ok:
                lines = [l for l in (p.stderr + p.stdout).splitlines() if "capturing" in l or "hipError" in l or "what()" in l]
                last = (lines[0] if lines else (p.stderr.strip().splitlines() or ["rc %d" % p.returncode])[-1])[:240]
        print("%-9s  %d / %d clean%s" % (mode, clean, runs, ("   last failure: " + last) if last else ""), flush=True)


if __name__ == "__main__":
    main()
