#!/usr/bin/env python
"""Condense a `rocprofv3 --kernel-trace --pmc ...` run of bench.py into a small per-kernel table (run ON the GPU box:
the raw counter CSV of a find-mode run is > 100 MB and must not travel back through gpurun_out/).

    python tools/pmc_step_summary.py <rocprof_dir> <out.md> [--steady <kernel substring> <launches per step>] [--top 40]

Per kernel name (of ONE steady-state step when --steady is given): dispatches, mean duration, mean counter values and,
when the counters are present, the matrix-pipe occupancy
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
(SQ_VALU_MFMA_BUSY_CYCLES counts per-SIMD busy cycles summed over the chip's 1024 SIMDs — 64 per v_mfma_f32_32x32x2_f32,
checked against SQ_INSTS_MFMA; GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: a 7.87 ms kernel shows 1.47e8 = 8 x
7.87 ms x 2.34 GHz).  Deletes nothing; the caller removes the raw directory.
"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)


def main():
    args = sys.argv[1:]
    steady, top = None, 40
    if "--steady" in args:
        i = args.index("--steady")
        steady = (args[i + 1], int(args[i + 2]))
        del args[i:i + 3]
    if "--top" in args:
        i = args.index("--top")
        top = int(args[i + 1])
        del args[i:i + 2]
    src, dst = args[0], args[1]
    # kernel trace: dispatch id -> (name, start, end)
    trace = {}
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            trace[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    lo, hi, note = -1, 1 << 62, "whole run"
    if steady:
        marks = sorted(v[1] for v in trace.values() if steady[0] in v[0])
        n = steady[1]
        if len(marks) >= 2 * n:
            lo, hi = marks[-2 * n], marks[-n]
            note = "ONE steady-state step (between consecutive '%s' launches): %.3f ms wall under the profiler" % (steady[0], (hi - lo) / 1e6)
    keep = {d for d, v in trace.items() if lo <= v[1] < hi}
    dur = collections.defaultdict(list)
    for d in keep:
        name, s, e = trace[d]
        dur[name].append((e - s) / 1e3)
    vals = collections.defaultdict(lambda: collections.defaultdict(dict))    # name -> counter -> dispatch -> value
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = r["Dispatch_Id"]
            if d not in keep:
                continue
            slot = vals[r["Kernel_Name"]][r["Counter_Name"]]
            slot[d] = slot.get(d, 0.0) + float(r["Counter_Value"])        # rows of one dispatch (per dimension) are summed
    counters = sorted({c for k in vals for c in vals[k]})
    ranked = sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:top]
    total = sum(sum(v) for v in dur.values())
    lines = ["rocprofv3 --kernel-trace --pmc %s; %s; %d kernel names, %d dispatches, %.3f ms of kernel time (serialised by the "
             "counter collection)" % (" ".join(counters), note, len(dur), sum(len(v) for v in dur.values()), total / 1e3), "",
             "| kernel | calls | total ms | mean us | " + " | ".join(counters) + " | mfma_busy |",
             "|---|---:|---:|---:|" + "---:|" * (len(counters) + 1)]
    for name, ds in ranked:
        cells, mean = [], {}
        for c in counters:
            v = vals.get(name, {}).get(c)
            if v:
                mean[c] = sum(v.values()) / len(v)
                cells.append("%.4g" % mean[c])
            else:
                cells.append("-")
        busy = "-"
        if mean.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
            busy = "%.1f %%" % (100.0 * mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * mean["GRBM_GUI_ACTIVE"] / 8.0))
        short = name.replace("void ", "")
        short = short if len(short) <= 110 else short[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.2f | %s | %s |" % (short, len(ds), sum(ds) / 1e3, sum(ds) / len(ds), " | ".join(cells), busy))
    with open(dst, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
