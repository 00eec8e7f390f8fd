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
    """per kernel: counter sums, launches per counter, and the summed duration (ms) of the profiled dispatches of that pass"""
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int)); dur = defaultdict(float); seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
            if r["Dispatch_Id"] not in seen and r.get("End_Timestamp"):
                seen.add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return acc, n, dur


# The three ceilings a kernel's counters are priced against.  Every fraction in <tag>_pmc_summary.csv is  counter / duration / ceiling  of that row.
HBM_PEAK = 8.0e12                 # B/s (MI355X_MICROARCH.md)
CLOCK_PEAK = 2.4e9                # Hz
SIMDS = 1024                      # 256 CUs x 4
VALU_ISSUE_PEAK = SIMDS * CLOCK_PEAK / 2.0   # wave64 VALU instructions / s: a SIMD-32 issues one in 2 cycles at best (transcendentals, conversions, 3-operand min / max take longer:
                                             # the fraction is a LOWER bound of the VALU pipes' busy time)
L1_LOOKUP_PEAK = 0.66e12          # scattered 16-B lane-loads / s the vector L1s sustain (tools/gather_probe.hip, L2-resident records: one lane-load per CU and cycle)


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
    rows = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC", "SQ", "SQ2", "TCP"):
        acc, n, dur = read_counters(os.path.join(src, "pmc_" + c))
        for k in acc:
            rows.setdefault(k, {})
            for name, v in acc[k].items():
                rows[k][name] = v; rows[k]["launches"] = n[k][name]
            if c == "SQ" or "duration_ms" not in rows[k]:
                rows[k]["duration_ms"] = dur[k]      # duration of the profiled dispatches (the SQ pass when there is one; the passes agree within the run-to-run spread)
    cols = ["duration_ms", "FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU",
            "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE",
            "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]
    def fractions(r):
        """the three roofline fractions of a kernel row, each = counter-derived rate / ceiling (recomputable from the row alone)"""
        t = r.get("duration_ms", 0.0) * 1e-3
        if t <= 0:
            return None, None, None, None, None
        lane = r.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * r["SQ_INSTS_VALU"]) if r.get("SQ_INSTS_VALU") else None
        hbm = (2.0 * r.get("FETCH_SIZE", 0.0) + r.get("WRITE_SIZE", 0.0)) * 1024.0 / t / HBM_PEAK if r.get("FETCH_SIZE") else None
        issue = r["SQ_INSTS_VALU"] / t / VALU_ISSUE_PEAK if r.get("SQ_INSTS_VALU") else None
        # L1 side: lane-level vector loads.  TCP_TOTAL_CACHE_ACCESSES when the counter exists on this rocprofv3; else wave-level load instructions x 64 x the kernel's VALU lane utilisation
        if r.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            l1 = r["TCP_TOTAL_CACHE_ACCESSES_sum"] / t / L1_LOOKUP_PEAK; l1_src = "TCP_TOTAL_CACHE_ACCESSES_sum"
        elif r.get("SQ_INSTS_VMEM_RD") and lane:
            l1 = r["SQ_INSTS_VMEM_RD"] * 64.0 * lane / t / L1_LOOKUP_PEAK; l1_src = "SQ_INSTS_VMEM_RD x 64 x valu_lane_utilisation"
        else:
            l1, l1_src = None, None
        clock = r["GRBM_GUI_ACTIVE"] / 8.0 / t if r.get("GRBM_GUI_ACTIVE") else None     # GRBM_GUI_ACTIVE sums the 8 XCDs
        return hbm, issue, l1, l1_src, clock
    out = os.path.join(dst, tag + "_pmc_summary.csv")
    with open(out, "w") as fh:
        fh.write("kernel,launches," + ",".join(cols) + ",memory_side_read_bytes(2xFETCH_SIZE),TCC_hit_rate,valu_lane_utilisation,hbm_frac,valu_issue_frac_min,l1_lookup_frac,l1_lookup_source,clock_GHz\n")
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
            hit, miss = r.get("TCC_HIT_sum", 0.0), r.get("TCC_MISS_sum", 0.0)
            lane = r.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * r["SQ_INSTS_VALU"]) if r.get("SQ_INSTS_VALU") else None
            hbm, issue, l1, l1_src, clock = fractions(r)
            f3 = lambda v: ("%.4f" % v) if v is not None else ""
            fh.write("\"%s\",%d,%s,%.0f,%s,%s,%s,%s,%s,%s,%s\n" % (k, r.get("launches", 0), ",".join("%.6g" % r.get(c, 0.0) for c in cols), 2 * r.get("FETCH_SIZE", 0.0) * 1024,
                                               ("%.3f" % (hit / (hit + miss))) if hit + miss > 0 else "", ("%.3f" % lane) if lane else "", f3(hbm), f3(issue), f3(l1), l1_src or "", ("%.3f" % (clock / 1e9)) if clock else ""))
        fh.write("# ceilings: hbm_frac = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / duration / %.3g B/s;  valu_issue_frac_min = SQ_INSTS_VALU / duration / %.4g (1024 SIMDs x 2.4 GHz / 2 cycles: a lower bound, slower instructions issue in more);  "
                 "l1_lookup_frac = lane-level loads / duration / %.3g (tools/gather_probe.hip)\n" % (HBM_PEAK, VALU_ISSUE_PEAK, L1_LOOKUP_PEAK))
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
        hbm, issue, l1, l1_src, clock = fractions(r)
        lane = r.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * r["SQ_INSTS_VALU"]) if r.get("SQ_INSTS_VALU") else None
        lane_loads = r.get("TCP_TOTAL_CACHE_ACCESSES_sum") or (r.get("SQ_INSTS_VMEM_RD", 0.0) * 64.0 * (lane or 0.0))
        entry = {"tag": tag, "kernel": k, "kernel_build": rf.get("kernel_build"), "rays_profiled": rays_run,
                 "fetch_bytes_per_ray_raw": fe_ray, "write_bytes_per_ray": wr_ray, "bytes_per_ray": 2 * fe_ray + wr_ray,
                 "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss > 0 else None,
                 "memory_side_read_requests_per_ray": r.get("TCC_EA0_RDREQ_sum", 0.0) / rays_run,
                 # what bench.py prices against the other two ceilings: per-ray counter quantities of the same profiled run (x the rays of ITS launches / ITS launch time)
                 "valu_insts_per_ray": r.get("SQ_INSTS_VALU", 0.0) / rays_run, "lane_loads_per_ray": lane_loads / rays_run, "lane_loads_source": l1_src,
                 "valu_lane_utilisation": round(lane, 4) if lane else None, "clock_ghz": round(clock / 1e9, 3) if clock else None,
                 "wait_any_frac": round(r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"], 4) if r.get("SQ_WAVE_CYCLES") else None,
                 "profile_fractions": {"hbm": hbm, "valu_issue_min": issue, "l1_lookup": l1},
                 "ceilings": {"hbm_bytes_per_s": HBM_PEAK, "valu_issue_insts_per_s": VALU_ISSUE_PEAK, "l1_lane_loads_per_s": L1_LOOKUP_PEAK}}
        # the shade kernel of the same run (bench.py roofline_shade): per shaded path vertex = per ray of the closest-hit traversal
        sk = [q for q in rows if q.startswith("k_shade")]
        if sk:
            cls = [q for q in sk if q.startswith("k_shade_class")]
            if cls:
                # full feature set: the product shades by model class (one launch per class and depth); bench.py's counting batch runs k_shade_full instead (no traversal keys while
                # counting) and is left out.  The class launches together ARE the shade stage: their counters and durations are summed, per vertex of the warm-up + timed passes.
                rs = {}
                for q2 in cls:
                    for k2, v2 in rows[q2].items():
                        if isinstance(v2, (int, float)): rs[k2] = rs.get(k2, 0.0) + v2
                q = "k_shade_class_* (" + " + ".join(sorted(c.replace("k_shade_class_", "") for c in cls)) + ")"
                verts = rf["closest_rays_total"] * (b["steps"] + b["warmup"]) / b["steps"] if rf.get("closest_rays_total") else None
            else:
                q = max(sk, key=lambda q: rows[q].get("duration_ms", 0.0)); rs = rows[q]
                # every closest-hit ray is one shaded vertex (or a miss that the shade kernel also handles); the profiled command shades the warm-up passes, the timed passes AND bench.py's counting batch
                verts = rf["closest_rays_total"] * (b["steps"] + b["warmup"] + rf.get("counting_passes", 0)) / b["steps"] if rf.get("closest_rays_total") else None
            sh_hbm, sh_issue, sh_l1, sh_src, _ = fractions(rs)
            sl = rs.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * rs["SQ_INSTS_VALU"]) if rs.get("SQ_INSTS_VALU") else None
            entry["shade"] = {"kernel": q, "launches": rs.get("launches"), "duration_ms": rs.get("duration_ms"), "vertices_profiled": verts,
                              "bytes_per_vertex": ((2 * rs.get("FETCH_SIZE", 0.0) + rs.get("WRITE_SIZE", 0.0)) * 1024 / verts) if verts else None,
                              "valu_insts_per_vertex": (rs.get("SQ_INSTS_VALU", 0.0) / verts) if verts else None,
                              "lane_loads_per_vertex": ((rs.get("TCP_TOTAL_CACHE_ACCESSES_sum") or rs.get("SQ_INSTS_VMEM_RD", 0.0) * 64.0 * (sl or 0.0)) / verts) if verts else None,
                              "valu_lane_utilisation": round(sl, 4) if sl else None, "profile_fractions": {"hbm": sh_hbm, "valu_issue_min": sh_issue, "l1_lookup": sh_l1},
                              "kernel_build": (b.get("roofline_shade") or {}).get("kernel_build")}   # bench.py SHADE_BUILD of the profiled run: hash of the shade sources
        t["workloads"][rf["workload_key"]] = entry
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
