#!/usr/bin/env python
"""Condenses a rocprofv3 --kernel-trace run into a small markdown table for profiles/.

rocprofv3 of ROCm 7.2 writes a rocpd SQLite database (<dir>/<name>_results.db) by default, or CSV
with --output-format csv; both are handled.  Usage:

    python tools/rocprof_summary.py <rocprof_out_dir_or_db> <out.md> [title] [--steady <kernel substring> <per_step>]

--steady K n: restrict the table to ONE steady-state step, delimited by consecutive occurrences of the
kernel whose name contains K and which is launched n times per step (MIOpen's find phase in the
warm-up otherwise dominates a short run).
"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def load(src):
    """-> list of (name, start_ns, end_ns)"""
    dbs = [src] if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True)
    rows = []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows += list(cur.execute("select name, start, end from kernels order by start"))
    if rows:
        return rows
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return sorted(rows, key=lambda r: r[1])


def main():
    args = sys.argv[1:]
    steady = None
    if "--steady" in args:
        i = args.index("--steady")
        steady = (args[i + 1], int(args[i + 2]))
        del args[i:i + 3]
    src, dst = args[0], args[1]
    title = args[2] if len(args) > 2 else os.path.basename(src)
    rows = load(src)
    if not rows:
        raise SystemExit("no kernel records under " + src)
    note = "whole run"
    if steady:
        marks = [r[1] for r in rows if steady[0] in r[0]]
        n = steady[1]
        a, b = marks[-2 * n], marks[-n]          # the last complete step
        rows = [r for r in rows if a <= r[1] < b]
        note = "ONE steady-state step (between consecutive '%s' launches): %.3f ms wall" % (steady[0], (b - a) / 1e6)
    agg = defaultdict(lambda: [0, 0.0])
    for name, s, e in rows:
        agg[name][0] += 1
        agg[name][1] += (e - s) / 1e3
    total = sum(v[1] for v in agg.values())
    ranked = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(dst, "w") as out:
        out.write("# %s\n\nrocprofv3 --kernel-trace, %s; %d kernel names, %d dispatches, %.3f ms of GPU kernel time\n\n"
                  % (title, note, len(ranked), sum(v[0] for v in agg.values()), total / 1e3))
        out.write("| kernel | calls | total ms | mean us | share |\n|---|---:|---:|---:|---:|\n")
        for name, (calls, us) in ranked[:40]:
            short = name if len(name) <= 100 else name[:97] + "..."
            out.write("| `%s` | %d | %.3f | %.2f | %.1f%% |\n" % (short.replace("|", "\\|"), calls, us / 1e3, us / calls,
                                                               100 * us / total))
        own = [(n, v) for n, v in ranked if "nextou" in n]
        out.write("\n## own kernels (libnextou_hip.so)\n\n| kernel | calls | total ms | mean us |\n|---|---:|---:|---:|\n")
        for name, (calls, us) in own:
            short = name.replace("(anonymous namespace)::", "").split("(")[0]
            out.write("| `%s` | %d | %.3f | %.2f |\n" % (short.replace("|", "\\|"), calls, us / 1e3, us / calls))
        own_us = sum(v[1] for _, v in own)
        out.write("\nown kernels: %.3f ms = %.1f%% of GPU kernel time\n" % (own_us / 1e3, 100 * own_us / max(total, 1e-9)))
    print("wrote", dst)


if __name__ == "__main__":
    main()
