#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the MI355X box) into profiles/<tag>_*:

  <tag>_bench.json          the bench.py JSON line of that run
  <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats per-kernel summary (verbatim)
  <tag>_pmc_summary.csv     per kernel: launches, FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them), TCC hit rate
  roofline_traffic.json     HBM bytes per launch of the dominant kernel (read by bench.py -> roofline.traffic)

FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KB.  MI355X_MICROARCH.md ("HBM"): on gfx950 FETCH_SIZE counts
128-B requests as 64 B for wide coalesced streams (x2 correction); other access shapes are uncalibrated.  The traversal
kernel issues 16 B/lane loads of scattered 64-B groups, so both the raw figure and the x2 figure are kept; bench.py's
`traffic` uses the x2 (upper) figure for reads and the raw figure for writes.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("ctl::", "").replace("void ", "")


def read_counters(d):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return acc, n


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
    rows = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC"):
        acc, n = read_counters(os.path.join(src, "pmc_" + c))
        for k in acc:
            rows.setdefault(k, {})
            for name, v in acc[k].items():
                rows[k][name] = v; rows[k]["launches"] = n[k][name]
    out = os.path.join(dst, tag + "_pmc_summary.csv")
    with open(out, "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_total,WRITE_SIZE_KB_total,fetch_bytes_per_launch_raw,fetch_bytes_per_launch_x2,write_bytes_per_launch,TCC_hit_rate\n")
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
            L = max(1, r.get("launches", 1)); fe, wr = r.get("FETCH_SIZE", 0.0), r.get("WRITE_SIZE", 0.0)
            hit, miss = r.get("TCC_HIT_sum", 0.0), r.get("TCC_MISS_sum", 0.0)
            fh.write("%s,%d,%.1f,%.1f,%.0f,%.0f,%.0f,%s\n" % (k, L, fe, wr, fe * 1024 / L, 2 * fe * 1024 / L, wr * 1024 / L, ("%.3f" % (hit / (hit + miss))) if hit + miss > 0 else ""))
    # dominant kernel = closest-hit intersect without counters: k_intersect<false, false, FLAT>
    dom = [k for k in rows if k.startswith("k_intersect<false, false")]
    if dom:
        k = max(dom, key=lambda k: rows[k].get("FETCH_SIZE", 0)); r = rows[k]; L = max(1, r.get("launches", 1))
        # The profiled run's launches are not all alike (warm-up launches carry 2 passes, timed ones up to 32): per-launch figures are quoted for
        # a TIMED launch = bytes per ray of the whole run x the rays of one timed launch (bench.json of the same tag).
        b = json.load(open(os.path.join(src, "bench.json"))); rf = b["roofline"]
        rays_run = rf["rays_per_launch"] * rf["launches"] * (b["steps"] + b["warmup"]) / b["steps"]
        fe_ray, wr_ray = r.get("FETCH_SIZE", 0) * 1024 / rays_run, r.get("WRITE_SIZE", 0) * 1024 / rays_run
        R = rf["rays_per_launch"]
        json.dump({"tag": tag, "kernel": k, "launches_profiled": L, "rays_profiled": rays_run, "rays_per_timed_launch": R,
                   "fetch_bytes_per_ray_raw": fe_ray, "write_bytes_per_ray": wr_ray,
                   "fetch_bytes_per_launch_raw": fe_ray * R, "write_bytes_per_launch": wr_ray * R,
                   "k_intersect_closest_bytes_per_launch": (2 * fe_ray + wr_ray) * R,
                   "note": "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), per ray over every closest-hit launch of the profiled `python bench.py --no-cpu-baseline` run, times the rays of one timed launch"},
                  open(os.path.join(dst, "roofline_traffic.json"), "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
