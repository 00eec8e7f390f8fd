#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the MI355X box) into profiles/<tag>_*:

  <tag>_bench.json          the bench.py JSON line of that run
  <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats per-kernel summary (verbatim)
  <tag>_pmc_summary.csv     per kernel: launches, FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them), memory-side read requests, TCC hit rate, SQ counters
  roofline_traffic.json     per workload: HBM-side bytes per RAY of the dominant kernel (bench.py multiplies by the rays of its own launches)

FETCH_SIZE / WRITE_SIZE are reported in KB.  Calibration (profiles/README.md, tools/gather_probe.hip under --pmc): on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B
while every request of these kernels is a 128-B line (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ), for random 64-B gathers as for streams: memory-side bytes =
2 x FETCH_SIZE.  WRITE_SIZE is taken as reported.
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
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC", "SQ"):
        acc, n = read_counters(os.path.join(src, "pmc_" + c))
        for k in acc:
            rows.setdefault(k, {})
            for name, v in acc[k].items():
                rows[k][name] = v; rows[k]["launches"] = n[k][name]
    cols = ["FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU",
            "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_THREAD_CYCLES_VALU"]
    out = os.path.join(dst, tag + "_pmc_summary.csv")
    with open(out, "w") as fh:
        fh.write("kernel,launches," + ",".join(cols) + ",memory_side_read_bytes(2xFETCH_SIZE),TCC_hit_rate,valu_lane_utilisation\n")
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
            hit, miss = r.get("TCC_HIT_sum", 0.0), r.get("TCC_MISS_sum", 0.0)
            lane = r.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * r["SQ_INSTS_VALU"]) if r.get("SQ_INSTS_VALU") else None
            fh.write("\"%s\",%d,%s,%.0f,%s,%s\n" % (k, r.get("launches", 0), ",".join("%.6g" % r.get(c, 0.0) for c in cols), 2 * r.get("FETCH_SIZE", 0.0) * 1024,
                                               ("%.3f" % (hit / (hit + miss))) if hit + miss > 0 else "", ("%.3f" % lane) if lane else ""))
    # dominant kernel = the one bench.py's roofline names: the fused closest + any-hit launch (FuseTraversal, default) or the plain closest-hit intersect
    b = json.load(open(os.path.join(src, "bench.json"))); rf = b["roofline"]
    dom = [k for k in rows if k.startswith("k_intersect_pair")] if "pair" in rf.get("kernel", "") else [k for k in rows if k.startswith("k_intersect<false, false")]
    if dom and rf.get("workload_key"):
        k = max(dom, key=lambda k: rows[k].get("FETCH_SIZE", 0)); r = rows[k]
        # rays of the profiled run's launches of that kernel (path + shadow rays for the fused kernel): timed launches + warm-up launches (same rays per pass)
        rays_run = rf["rays_per_launch"] * rf["launches"] * (b["steps"] + b["warmup"]) / b["steps"]
        fe_ray, wr_ray = r.get("FETCH_SIZE", 0) * 1024 / rays_run, r.get("WRITE_SIZE", 0) * 1024 / rays_run
        hit, miss = r.get("TCC_HIT_sum", 0.0), r.get("TCC_MISS_sum", 0.0)
        tpath = os.path.join(dst, "roofline_traffic.json")
        try:
            t = json.load(open(tpath))
            if "workloads" not in t: t = {"workloads": {}}
        except Exception:
            t = {"workloads": {}}
        t["note"] = "HBM-side bytes per ray of the dominant traversal kernel (k_intersect_pair: per path-or-shadow ray) = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes) over every launch of that kernel in the profiled bench.py run / the rays of those launches; bench.py quotes an entry only for the same workload key and kernel build"
        t["workloads"][rf["workload_key"]] = {"tag": tag, "kernel": k, "kernel_build": rf.get("kernel_build"), "rays_profiled": rays_run,
                                               "fetch_bytes_per_ray_raw": fe_ray, "write_bytes_per_ray": wr_ray, "bytes_per_ray": 2 * fe_ray + wr_ray,
                                               "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss > 0 else None,
                                               "memory_side_read_requests_per_ray": r.get("TCC_EA0_RDREQ_sum", 0.0) / rays_run}
        json.dump(t, open(tpath, "w"), indent=1)
    # the contract's cross-check: rocprofv3's own durations of the dominant kernel's TIMED launches (the last `launches` dispatches of the kernel-trace run:
    # warm-up launches come first and carry fewer passes) against the average bench.py measured with HIP events in its own, unprofiled run
    kt = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)
    if kt and dom:
        want = "k_intersect_pair" if "pair" in rf.get("kernel", "") else "k_intersect<false, false"
        d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt[0])) if short(r["Kernel_Name"]).startswith(want)]
        d.sort()
        timed = d[-int(rf["launches"]):] if rf.get("launches") else d
        if timed:
            avg_ms = sum(e - s for s, e in timed) / len(timed) / 1e6
            chk = {"kernel": want, "profiled_dispatches": len(d), "timed_launches": len(timed), "rocprof_timed_avg_ms": round(avg_ms, 4),
                   "bench_avg_launch_ms": rf.get("avg_launch_ms"), "ratio": round(avg_ms / rf["avg_launch_ms"], 4) if rf.get("avg_launch_ms") else None,
                   "note": "rocprofv3 --kernel-trace durations of the last `timed_launches` dispatches of the dominant kernel (the warm-up launches precede them) vs bench.py's HIP-event average of its own unprofiled run"}
            json.dump(chk, open(os.path.join(dst, tag + "_roofline_check.json"), "w"), indent=1)
            print(json.dumps(chk))
    print(open(out).read())


if __name__ == "__main__":
    main()
