#!/usr/bin/env python3
"""Fill the @TOKEN@ placeholders of tools/design_template.md from the round's final GPU run (gpurun_out/<tag>/ + profiles/<tag>_*): python tools/fill_design.py r05x > DESIGN.md"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]                                   # profile tag (profiles/<tag>_pmc_summary.csv, plugin compare, shard probe)
G = os.path.join(ROOT, "gpurun_out", tag)
G2 = os.path.join(ROOT, "gpurun_out", sys.argv[2]) if len(sys.argv) > 2 else G    # bench lines that quote the committed profile, where they exist


def line(name):
    p = os.path.join(G2, name)
    return json.loads(open(p if os.path.exists(p) else os.path.join(G, name)).read().strip().splitlines()[-1])


def brief(j):
    r = j["roofline"]
    return j["value"], j["ms_per_step"], r["ms_intersect"] / j["steps"], r["ms_shade"] / j["steps"], r["per_ray"], r


T = {"TAG": tag}
fin = line("bench_final.json"); v, ms, tr, sh, pr, r = brief(fin)
T["SM20"] = "%.0f" % v; T["SM20_MS"] = "%.2f (%.2f + %.2f)" % (ms, tr, sh); T["VIS_INNER"] = "%.1f" % pr["n_inner"]; T["VIS_TRI"] = "%.1f" % pr["n_tri"]
T["TRAV_MS"] = "%.2f" % tr; T["TRAV_SHARE"] = "%.0f" % (100 * tr / ms); T["SHADE_MS"] = "%.2f" % sh
fr = r.get("fractions", {})
T["SM_ROOF"] = "%s: hbm %.2f, valu_issue (2 cycles / instr) %.2f, l1_lookup %.2f" % (r.get("bound"), fr.get("hbm", 0), fr.get("valu_issue", 0), fr.get("l1_lookup", 0))
T["TRAV_HBM"] = "%.2f" % fr.get("hbm", 0); T["TRAV_L1"] = "%.2f" % fr.get("l1_lookup", 0); T["TRAV_L2"] = "%.2f" % (r.get("l2_hit_rate") or 0)
T["B_PATH"] = "%.0f" % r.get("bytes_per_path_ray", 0); T["B_SHADOW"] = "%.0f" % r.get("bytes_per_shadow_ray", 0); T["B_RAY"] = "%.0f" % r.get("bytes_per_ray", 0)
T["RAYS_LAUNCH"] = "%.1f" % (r.get("rays_per_launch", 0) / 1e6); T["ALG_GB"] = "%.1f" % (r.get("bytes_per_ray", 0) * r.get("rays_per_launch", 0) / 1e9)
T["TRAFFIC_GB"] = "%.1f" % ((r.get("traffic") or 0) / 1e9); T["TRAFFIC_RATIO"] = "%.2f" % ((r.get("traffic") or 0) / max(1.0, r.get("bytes_per_ray", 0) * r.get("rays_per_launch", 0)))
T["LAUNCH_MS"] = "%.2f" % r.get("avg_launch_ms", 0)
rs = fin.get("roofline_shade", {})
T["SH_ALG"] = "%.0f" % (rs.get("algorithmic_bytes_per_vertex") or 0); T["SH_TRAFFIC"] = "%.0f" % (rs.get("traffic_per_vertex") or 0); T["SH_VERT"] = "%.2f" % ((rs.get("vertices") or 0) / fin["steps"] / 1e6)
T["SHADE_HBM"] = "%.2f" % (rs.get("fractions", {}).get("hbm", 0))
cpu = fin.get("cpu_baseline", {})
T["CPU"] = "%.2f Mrays/s" % cpu.get("value", 0); T["CPU_CORES"] = str(cpu.get("cores")); T["CPU_X"] = "%.0f" % (v / cpu["value"]) if cpu.get("value") else "?"
j = line("bench_64spp.json"); v, ms, tr, sh, pr, r = brief(j); T["SM64"] = "%.0f" % v; T["SM64_MS"] = "%.2f (%.2f + %.2f)" % (ms, tr, sh)
T["LOADER"] = "%.0f" % line("bench_loader.json")["value"]; T["CORNELL"] = "%.0f" % line("bench_cornell.json")["value"]
for key, f in (("HARD", "bench_sm_hard.json"), ("BATH", "bench_bathroom.json")):
    j = line(f); v, ms, tr, sh, pr, r = brief(j)
    T[key] = "%.0f" % v; T[key + "_MS"] = "%.2f (%.2f + %.2f)" % (ms, tr, sh); T[key + "_VIS"] = "%.1f + %.1f" % (pr["n_inner"], pr["n_tri"])
    fr = r.get("fractions") or {}
    T[key + "_ROOF"] = ("%s: hbm %.2f, valu_issue %.2f, l1_lookup %.2f" % (r.get("bound"), fr.get("hbm", 0), fr.get("valu_issue", 0), fr.get("l1_lookup", 0))) if fr else str(r.get("bound"))
    if key == "BATH": T["BATH_SHADE"] = "%.2f" % sh
T["BATH128"] = "%.0f" % line("bench_bathroom_128spp.json")["value"]
try: T["BATHRED"] = "%.0f" % line("bench_bathroom_reduced.json")["value"]
except Exception: T["BATHRED"] = "?"
# SQ_WAIT_ANY of the shade kernel from the pmc summary
for row in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.csv"))):
    k = row.get("kernel") or list(row.values())[0]
    if k and k.startswith("k_shade_basic") and row.get("SQ_WAIT_ANY"):
        T["SHADE_WAIT"] = "%.2f" % (float(row["SQ_WAIT_ANY"]) / float(row["SQ_WAVE_CYCLES"]))
# plugin compare: the last line is a JSON object with both plugins' ms_per_pass
pc = json.loads(open(os.path.join(G, "plugin_compare.txt")).read().strip().splitlines()[-1])
a, b = pc["PathTracer"]["ms_per_pass"], pc["WavefrontPathTracer"]["ms_per_pass"]
T["MEGA_MS"] = "%.1f" % a; T["MEGA_X"] = "%.1f" % (a / b)
# shard probe
for l in open(os.path.join(G, "shard_time_probe.txt")):
    m = re.match(r"passes (\d+) world (\d+) FuseTraversal 1:\s+([0-9.]+) ms", l)
    if m: T["S%s_%s" % (m.group(1), m.group(2))] = m.group(3)
for p in ("20", "64", "256"):
    one, eight = float(T["S%s_1" % p]), float(T["S%s_8" % p])
    T["S%s_X" % p] = "%.2f" % (one / eight); T["S%s_XG" % p] = "%.2f" % (one / (eight + 0.3))
s = open(os.path.join(ROOT, "tools", "design_template.md")).read()
missing = sorted(set(re.findall(r"@([A-Z0-9_]+)@", s)) - set(T))
if missing: print("missing tokens:", missing, file=sys.stderr)
print(re.sub(r"@([A-Z0-9_]+)@", lambda m: T.get(m.group(1), m.group(0)), s), end="")
