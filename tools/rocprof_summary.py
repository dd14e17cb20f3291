#!/usr/bin/env python
"""Condenses a rocprofv3 --kernel-trace [--stats] output directory into a small markdown table
(per kernel: calls, total ms, mean us, share) for profiles/.  Usage:
    python tools/rocprof_summary.py <rocprof_out_dir> <out.md> [title]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(src)
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + src)
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or row.get("kernel_name")
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3  # us
            agg[name][0] += 1
            agg[name][1] += dur
    total = sum(v[1] for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(dst, "w") as out:
        out.write("# %s\n\nrocprofv3 --kernel-trace; %d kernels, %d dispatches, %.3f ms of GPU kernel time in total\n\n"
                  % (title, len(rows), sum(v[0] for v in agg.values()), total / 1e3))
        out.write("| kernel | calls | total ms | mean us | share |\n|---|---:|---:|---:|---:|\n")
        for name, (calls, us) in rows[:45]:
            short = name if len(name) <= 110 else name[:107] + "..."
            out.write("| `%s` | %d | %.3f | %.2f | %.1f%% |\n" % (short.replace("|", "\\|"), calls, us / 1e3, us / calls, 100 * us / total))
        own = [(n, v) for n, v in rows if "nextou" in n]
        out.write("\n## own kernels (libnextou_hip.so)\n\n| kernel | calls | total ms | mean us |\n|---|---:|---:|---:|\n")
        for name, (calls, us) in own:
            out.write("| `%s` | %d | %.3f | %.2f |\n" % (name.replace("|", "\\|"), calls, us / 1e3, us / calls))
        out.write("\nown kernels: %.3f ms = %.1f%% of GPU kernel time\n" % (sum(v[1] for _, v in own) / 1e3,
                                                                         100 * sum(v[1] for _, v in own) / max(total, 1e-9)))
    print("wrote", dst)


if __name__ == "__main__":
    main()
