#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection.csv files per kernel: tools/pmc_table.py gpurun_out/<tag>/<pass> [...]"""
import csv, glob, os, sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("ctl::", "").replace("void ", "")


def table(d):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int)); dur = defaultdict(float)
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
            key = (r["Dispatch_Id"], r["Counter_Name"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return acc, n, dur


if __name__ == "__main__":
    for d in sys.argv[1:]:
        acc, n, dur = table(d)
        print("==", d)
        for k in sorted(acc, key=lambda k: -dur[k]):
            print("%-44s launches %3d  %9.3f ms  " % (k[:44], max(n[k].values()), dur[k]) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(acc[k].items())))
