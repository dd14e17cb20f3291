#!/usr/bin/env python
"""Mean per-dispatch PMC counter values per kernel from rocprofv3 --pmc CSV output directories.

    python tools/pmc_table.py <dir> [<dir> ...] --match bn_   ->  markdown table, one column per kernel name
"""
import collections
import csv
import glob
import os
import sys


def main():
    args = sys.argv[1:]
    match = "nextou::"
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                if match in name:
                    vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = sorted(vals)
    counters = sorted({c for k in kernels for c in vals[k]})
    print("| counter | " + " | ".join("`%s`" % k.replace("nextou::", "") for k in kernels) + " |")
    print("|---|" + "---:|" * len(kernels))
    for c in counters:
        cells = []
        for k in kernels:
            v = vals[k].get(c)
            cells.append("%.3g" % (sum(v) / len(v)) if v else "-")
        print("| %s | %s |" % (c, " | ".join(cells)))


if __name__ == "__main__":
    main()
