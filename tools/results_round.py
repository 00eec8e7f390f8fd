#!/usr/bin/env python3
"""BASELINE.md §2's per-config result records in one file: runs on the MI355X box (through gpurun), writes gpurun_out/<tag>/results.json; `--render-md` (anywhere) turns a
results JSON into RESULTS.md.

  C1  Cornell diffuse 256x256x16 spp, depth 8: the CPU path only (the oracle): Mrays/s on 1 thread and on the calibrated thread count; GPU frame vs oracle, max abs per-pixel difference
  C2  Cornell + glass sphere 1024^2 x 64 spp, depth 8: Mrays/s CPU and GPU, speed-up, RMSE / PSNR of the GPU image against the CPU (oracle) image with the same sampler tables
  C3  San Miguel stand-ins (synthetic-SM, synthetic-sm-hard) 1920x1080: bench.py's line (Mrays/s, roofline fractions, CPU baseline, shade share, lane utilisation)
  C4  8 x MI355X: the fields bench.py --gpus N writes (rank_ms, reduce_ms, ...) — filled where an N-GPU line exists, else "not measured on hardware"
  C5  Bathroom stand-in (synthetic-bathroom) 1920x1080: Mrays/s, shade-kernel time share, active-lane % (the profile's VALU lane utilisation when a profile of this build exists)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench(args, timeout=1500):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    raise RuntimeError("bench.py %s gave no JSON line: %s" % (args, p.stderr[-500:]))


def brief(b):
    r = b["roofline"]
    out = {"Mrays_per_s": b["value"], "ms_per_step": b["ms_per_step"], "steps": b["steps"], "workload": b["config"]["workload"], "scene_source": b["config"]["scene_source"],
           "traversal_ms_per_step": round(r["ms_intersect"] / b["steps"], 3), "shade_ms_per_step": round(r["ms_shade"] / b["steps"], 3),
           "shade_time_share": round(r["ms_shade"] / (b["ms_per_step"] * b["steps"]), 4), "traversal_time_share": round(r["ms_intersect"] / (b["ms_per_step"] * b["steps"]), 4),
           "lane_utilisation_traversal": r["lane_utilisation"], "node_visits_per_path_ray": r["per_ray_gpu_visited"], "roofline_bound": r["bound"], "roofline_fractions": r.get("fractions"),
           "hbm_frac": r["frac"], "algorithmic_frac": r.get("algorithmic_frac"), "avg_launch_ms": r["avg_launch_ms"]}
    if "roofline_shade" in b:
        out["shade_roofline"] = {k: b["roofline_shade"][k] for k in ("kernel", "bound", "fractions", "valu_lane_utilisation_profiled")}
    if "cpu_baseline" in b:
        c = b["cpu_baseline"]
        out["cpu"] = {"Mrays_per_s": c["value"], "threads": c["cores"], "per_core": c.get("per_core"), "cgroup_cpu_quota": c.get("cgroup_cpu_quota"), "sample": c.get("sample"), "kind": c["kind"]}
        out["gpu_over_cpu"] = round(b["value"] / c["value"], 1) if c["value"] else None
    return out


def cornell(width, spp, glass, depth=8):
    """GPU (wavefront plugin, flattened BVH) and oracle on the same sampler tables: rates, and the image difference"""
    import numpy as np
    import cudatracerlib_amd as ctl
    from cudatracerlib_amd import scenes
    import oracle
    sc = scenes.cornell_box(width, width, glass_sphere=glass)
    orc = oracle.Oracle(shared_math=True)
    tables = orc.sequence_tables(spp)
    threads = min(32, os.cpu_count() or 1)
    t = time.time(); _, r1 = orc.render(sc.desc, width, width, n_passes=1, tables=tables[:1], max_path_length=depth, threads=1, rows=(width // 2 - 8, width // 2 + 8)); t1 = time.time() - t
    t = time.time(); want, rays_cpu = orc.render(sc.desc, width, width, n_passes=spp, tables=tables, max_path_length=depth, threads=threads); t_cpu = time.time() - t
    scene = ctl.Scene(sc.desc, flatten=True)
    tr = ctl.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", depth); tr.Resize(width, width); tr.InitializeScene(scene)
    img = ctl.Image(width, width)
    for k in range(spp):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData()
    # the rate: the tracer's own batched passes (what bench.py times), not the table-by-table parity run above
    tr2 = ctl.WavefrontPathTracer(); tr2.getParameters().setValue("MaxPathLength", depth); tr2.Resize(width, width); tr2.InitializeScene(scene); tr2.reservePasses(spp)
    img2 = ctl.Image(width, width); tr2.DoPasses(img2, min(spp, 4), new_trace=True); ctl.api._check(ctl.lib.ctl_device_synchronize())
    t = time.time(); tr2.DoPasses(img2, spp, new_trace=False); ctl.api._check(ctl.lib.ctl_device_synchronize()); t_gpu = time.time() - t
    rays_gpu = float(tr2.stats().rays_last_pass)
    a = got[..., :3] / np.maximum(got[..., 6:7], 1); b = want[..., :3] / np.maximum(want[..., 6:7], 1)
    mse = float(((a - b) ** 2).mean()); peak = float(b.max())
    return {"size": "%dx%d x %d spp, depth %d" % (width, width, spp, depth), "cpu_Mrays_per_s": round(rays_cpu / t_cpu / 1e6, 3), "cpu_threads": threads,
            "cpu_one_thread_Mrays_per_s": round(r1 / t1 / 1e6, 3), "gpu_Mrays_per_s": round(rays_gpu / t_gpu / 1e6, 1), "gpu_over_cpu": round(rays_gpu / t_gpu / (rays_cpu / t_cpu), 1),
            "image": {"rmse": mse ** 0.5, "psnr_db_peak1": (10 * np.log10(1.0 / mse)) if mse > 0 else None, "psnr_db_peak_image_max": (10 * np.log10(peak * peak / mse)) if mse > 0 else None,
                      "max_abs_diff": float(np.abs(a - b).max()), "bit_equal_pixels": float((got[..., :3] == want[..., :3]).all(axis=2).mean()),
                      "pixels_within_2e-3": float((np.abs(got[..., :3] - want[..., :3]) <= 2e-3 * (1 + np.abs(want[..., :3]))).all(axis=2).mean()),
                      "note": "GPU frame (WavefrontPathTracer, flattened BVH) vs the CPU restatement (oracle, shared-math build) on the SAME sampler tables; radiance = rgb sum / weight; an independent-seed comparison would measure Monte-Carlo noise, not the implementation"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r04"); ap.add_argument("--render-md", default=None, metavar="results.json"); ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    if a.render_md:
        return render(json.load(open(a.render_md)))
    out_dir = os.path.join(ROOT, "gpurun_out", a.tag); os.makedirs(out_dir, exist_ok=True)
    R = {"tag": a.tag, "host_cpus": os.cpu_count()}
    R["C1"] = cornell(256, 16, glass=False)
    R["C2"] = cornell(1024, 64 if not a.quick else 8, glass=True)
    R["C3"] = {"synthetic-SM": brief(bench(["--steps", "20", "--warmup", "5"])),
               "synthetic-SM 64 spp (BASELINE's own step count)": brief(bench(["--steps", "64", "--warmup", "5", "--no-cpu-baseline"])),
               "synthetic-SM via loader": brief(bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--via-loader"])),
               "synthetic-sm-hard": brief(bench(["--steps", "20", "--warmup", "5", "--workload", "synthetic-sm-hard"])),
               "synthetic-sm-hard AlphaTest": brief(bench(["--steps", "20", "--warmup", "5", "--workload", "synthetic-sm-hard", "--no-cpu-baseline", "--tracer-param", "AlphaTest=true"]))}
    R["C5"] = {"synthetic-bathroom": brief(bench(["--steps", "20", "--warmup", "5", "--workload", "synthetic-bathroom"])),
               "synthetic-bathroom 128 spp (config 5's own step count)": brief(bench(["--steps", "128", "--warmup", "5", "--workload", "synthetic-bathroom", "--no-cpu-baseline"]))}
    R["C4"] = {"status": "no multi-GPU node was available to this round's gpurun calls (1-GPU boxes): not measured on hardware.  In particular the default exchange of an N-rank run — ONE ncclGather of each rank's own tiles (ctl_image_gather) — and its fallback ncclReduce have only ever run with ONE rank against the real librccl.so (CTL_BENCH_COMM_WORLD1=1) and, with 2 / 3 / 8 ranks, through the gloo path that moves the same packed tiles: the multi-rank collective itself, its time-out and its fall-back on a fresh communicator are unverified on hardware (bench.py warms both exchanges up before the timed region, all ranks agree on the one that worked, CTL_BENCH_NO_GATHER=1 keeps the reduce)",
               "fields_bench_writes": ["value", "rank_ms[]", "reduce_ms", "reduce_ms_per_rank[]", "slowest_rank", "rays_per_rank[]", "config.framebuffer_reduce"],
               "emulated_on_one_gpu": "tools/shard_time_probe.py (one rank of N rendering its tile shard at 20 / 64 / 256 passes) and the 8-process rehearsal on one device (CTL_BENCH_SHARE_GPU=1 bench.py --gpus 8: the gathered frame equals the one-rank frame): DESIGN.md section 7, profiles/r06e_shard_time_probe.txt, profiles/r06e_bench_8ranks_shared_gpu.json, profiles/r06e_frames_8_vs_1.txt"}
    json.dump(R, open(os.path.join(out_dir, "results.json"), "w"), indent=1)
    print(json.dumps(R)[:3000])


def render(R):
    L = ["# RESULTS — round %s (`profiles/results_%s.json`, written by `tools/results_round.py` on one MI355X box)" % (R["tag"][1:3].lstrip("0"), R["tag"]), "",
         "Per-config records of BASELINE.md §2.  Every GPU number is a 1-GPU run of this tree's `bench.py` / tracer; every CPU number is the oracle (kind `port`) on the box's host CPUs (%s visible; the container's cgroup quota is in each `cpu` record)." % R["host_cpus"], ""]
    for key in ("C1", "C2"):
        c = R[key]; im = c["image"]
        L += ["## %s  Cornell box%s, %s" % (key, " + glass sphere" if key == "C2" else " (diffuse)", c["size"]), "",
              "| | |", "|---|---|", "| CPU path (oracle), %d threads | %.3f Mrays/s (one thread: %.3f) |" % (c["cpu_threads"], c["cpu_Mrays_per_s"], c["cpu_one_thread_Mrays_per_s"]),
              "| 1 x MI355X | %.1f Mrays/s (%.1fx) |" % (c["gpu_Mrays_per_s"], c["gpu_over_cpu"]),
              "| image, GPU vs CPU on the same sampler tables | RMSE %.3g, PSNR %s dB (peak 1.0) / %s dB (peak = brightest pixel), max abs diff %.3g |" % (
                  im["rmse"], "%.1f" % im["psnr_db_peak1"] if im["psnr_db_peak1"] else "inf", "%.1f" % im["psnr_db_peak_image_max"] if im["psnr_db_peak_image_max"] else "inf", im["max_abs_diff"]),
              "| pixels bit-equal / within 2e-3 (1 + ref) | %.4f / %.4f |" % (im["bit_equal_pixels"], im["pixels_within_2e-3"]), ""]
    def table(title, d):
        rows = ["## " + title, "", "| workload | Mrays/s | ms / step (traversal + shade) | shade share | lanes busy: node steps / entry tests | roofline: bound, fractions | CPU (threads) | GPU / CPU |", "|---|---|---|---|---|---|---|---|"]
        for name, b in d.items():
            cpu = b.get("cpu")
            rows.append("| %s | %.1f | %.3f (%.2f + %.2f) | %.1f %% | %.2f / %.2f | %s %s | %s | %s |" % (
                name, b["Mrays_per_s"], b["ms_per_step"], b["traversal_ms_per_step"], b["shade_ms_per_step"], 100 * b["shade_time_share"], b["lane_utilisation_traversal"]["inner"], b["lane_utilisation_traversal"]["tri"],
                b["roofline_bound"], json.dumps(b["roofline_fractions"]) if b["roofline_fractions"] else "(no profile of this workload and build)",
                ("%.2f Mrays/s (%d)" % (cpu["Mrays_per_s"], cpu["threads"])) if cpu else "—", ("%.0fx" % b["gpu_over_cpu"]) if b.get("gpu_over_cpu") else "—"))
            if b.get("shade_roofline"):
                s = b["shade_roofline"]; rows.append("| ↳ shade kernel `%s` | | | | VALU lanes busy %.2f | %s %s | | |" % (s["kernel"], s.get("valu_lane_utilisation_profiled") or 0, s["bound"], json.dumps(s["fractions"])))
        return rows + [""]
    L += table("C3  San Miguel stand-ins, 1920x1080, depth 8, 20 spp per run unless said (assets absent on every box: SURVEY §8d)", R["C3"])
    L += table("C5  Bathroom stand-in, 1920x1080, depth 8", R["C5"])
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", R["tag"][:3] + "*_bench_bathroom_reduced.json")))   # the opt-in scene flag, one bench line of the same round's build and box series (tools/r0N_results.sh)
    red = cands[-1] if cands else ""
    if red:
        b = json.loads(open(red).read().strip().splitlines()[-1])
        L[-1:] = ["Rough plastic runs the reference's own 3-D transmittance lookup (frames equal the CPU path's to the bit; DESIGN.md §4).  With the opt-in scene flag "
                  "`CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE` (a per-material 1-D reduction of the table, the behaviour of rounds 2-4: within the per-pixel tolerance in most scenes, not equal to the bit): "
                  "**%.1f Mrays/s**, %.3f ms per step (`profiles/%s`)." % (b["value"], b["ms_per_step"], os.path.basename(red)), ""]
    L += ["## C4  8 x MI355X", "", R["C4"]["status"] + ".  `bench.py --gpus N` writes: " + ", ".join("`%s`" % f for f in R["C4"]["fields_bench_writes"]) + ".  " + R["C4"]["emulated_on_one_gpu"] + ".", ""]
    open(os.path.join(ROOT, "RESULTS.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
