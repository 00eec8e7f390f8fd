#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X wavefront path tracer.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: bench.py launches the N ranks itself (one process per GPU, rank r on device r, the reference's
one-process-per-device flow, main.cpp:160-172) and forwards rank 0's JSON line; under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

metric   : Mrays/s (primary + continuation + shadow rays / wall time of the render loop; scene load, BVH build and
           image read-back excluded), SURVEY §8d, on the 1920x1080 depth-8 workload.
step     : one pass (1 sample per pixel of the 1920x1080 frame) through ray-gen -> {intersect, shade} x depth 8.
workload : "synthetic-SM" — the seeded procedural stand-in for San Miguel (BASELINE.json configs[2]; the asset is not
           in the reference tree, this repo or the GPU box), 2000 instanced meshes / 8.7 M instanced triangles.
N > 1    : image tiles (64x64) are sharded round-robin over the ranks (weak in nothing: total work fixed => "strong");
           every rank holds a full scene replica; ONE RCCL reduce of the PixelData framebuffer to rank 0 closes the timed
           region (tiles are disjoint up to film-edge jitter, so the sum is the gather).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_scene(args, rank=0, barrier=None):
    """-> (DynamicScene, how it was built).  --scene FILE: a Mitsuba XML file through ctl_parse_mitsuba_scene.  --via-loader: the synthetic workload written
    as a Mitsuba-0.5 scene (XML + .serialized meshes) and loaded back through the same loader — the reference's flow ParseMitsubaScene -> UpdateScene -> tracer."""
    from cudatracerlib_amd import scenes
    if args.scene:
        sc = scenes.load_mitsuba(args.scene, args.width, args.height)
        return sc, "Mitsuba XML %s through ctl_parse_mitsuba_scene" % os.path.basename(args.scene)
    if args.via_loader:
        if args.workload != "synthetic-sm":
            raise SystemExit("--via-loader is implemented for the synthetic-sm workload")
        d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_scene_sm_%d_%d_%dx%d" % (args.instances, args.subdiv, args.width, args.height))
        xml = os.path.join(d, "scene.xml")
        if rank == 0 and not os.path.exists(xml):
            desc = scenes.with_explicit_normals(scenes.synthetic_sm_description(args.width, args.height, n_instances=args.instances, subdiv=args.subdiv))
            scenes.export_mitsuba(desc, d)
        if barrier:
            barrier()
        return scenes.load_mitsuba(xml), "Mitsuba XML + .serialized meshes (scenes.export_mitsuba) through ctl_parse_mitsuba_scene"
    if args.workload == "synthetic-sm-hard":   # ~8.5 M unique triangles (sliver terrain, alpha-masked cards, thin beams), textured materials, sun + area lights — always through the loader
        nx, nz = (4096, 1024) if args.hard_scale == "full" else (1024, 256)
        d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_scene_sm_hard_%dx%d_%dx%d" % (nx, nz, args.width, args.height))
        if rank == 0 and not os.path.exists(os.path.join(d, "scene.xml")):
            scenes.write_sm_hard_mitsuba(d, args.width, args.height, nx=nx, nz=nz, cards=4000 if args.hard_scale == "full" else 1000, beams=3000 if args.hard_scale == "full" else 600)
        if barrier:
            barrier()
        return scenes.load_mitsuba(os.path.join(d, "scene.xml"), args.width, args.height), "Mitsuba XML + .serialized meshes + PNG textures (scenes.write_sm_hard_mitsuba) through ctl_parse_mitsuba_scene"
    if args.workload == "synthetic-sm":
        return scenes.synthetic_sm(args.width, args.height, n_instances=args.instances, subdiv=args.subdiv), "builder API (cudatracerlib_amd/scenes.py)"
    if args.workload == "cornell-glass":      # BASELINE configs[1] (quote it with --width 1024 --height 1024)
        return scenes.cornell_box(args.width, args.height, glass_sphere=True), "builder API (cudatracerlib_amd/scenes.py)"
    if args.workload == "synthetic-bathroom":  # stand-in for BASELINE configs[4]: rough BSDFs + environment emitter, the shading stress
        return scenes.synthetic_bathroom(args.width, args.height), "builder API (cudatracerlib_amd/scenes.py)"
    raise SystemExit("unknown workload " + args.workload)


def b_ray(counts, rays):
    """algorithmic bytes: 32 (ray) + 16 (result) + 64 N_inner + 52 N_tri + 108 N_inst per ray"""
    return 48.0 * rays + 64.0 * counts.n_inner + 52.0 * counts.n_tri + 108.0 * counts.n_inst


def cpu_baseline(desc, args, flat_desc):
    """The oracle (CPU restatement of PathTrace<DIRECT> over the reference's two-level BVH) on a bounded sample of the same workload:
    (1) one thread on a band of rows (per-core figure), (2) all host cores on growing bands, then whole frames, for ~10 s (the baseline),
    (3) a band in counting mode over the product's flattened BVH: N_inner / N_tri per ray for the roofline.
    The oracle runs persistent threads over 64x16 tiles handed out dynamically and is built -O3 -march=x86-64-v3."""
    import oracle
    orc = oracle.Oracle()
    cores = os.cpu_count() or 1
    kw = dict(direct=True, max_path_length=args.depth, rr_start=5)
    mid = args.height // 2
    # (1) one thread, >= ~3 s
    rows1, t1, rays1 = 4, 0.0, 0
    while t1 < 3.0 and rows1 <= args.height:
        t = time.time(); _, r = orc.render(desc, args.width, args.height, n_passes=1, threads=1, rows=(mid, min(args.height, mid + rows1)), **kw); t1 = time.time() - t; rays1 = r
        if t1 < 3.0: rows1 *= 2
    per_core = rays1 / t1 / 1e6
    # (2) as many threads as the host rewards: on the GPU box (2 x 64 cores / 256 hardware threads, and whatever CPU share its container gets) the oracle's
    # rays/s peaks far below 256 threads (tools/cpu_scaling_probe.py: 16.9 Mrays/s at 32 threads, 9.8 at 256), so the thread count is calibrated first
    host_threads = cores
    best = (0.0, 1)
    th = 8
    while th <= host_threads or th // 2 < host_threads:
        n_th = min(th, host_threads)
        t = time.time(); _, r = orc.render(desc, args.width, args.height, n_passes=1, threads=n_th, rows=(mid, min(args.height, mid + max(8, 2 * n_th))), **kw); dt = time.time() - t
        if r / dt > best[0]: best = (r / dt, n_th)
        if n_th == host_threads: break
        th *= 2
    cores = best[1]
    rows, total_rays, total_t, done_rows, extra_passes = 16, 0, 0.0, 0, 0
    while total_t < 10.0 and done_rows < args.height:
        a, b = done_rows, min(args.height, done_rows + rows)
        t = time.time(); _, rays = orc.render(desc, args.width, args.height, n_passes=1, threads=cores, rows=(a, b), **kw); total_t += time.time() - t
        total_rays += rays; done_rows += b - a; rows = min(rows * 2, 512)
    n = 2
    while total_t < 10.0 and extra_passes < 64:      # many-core hosts finish the frame in ~2 s: add whole-frame passes
        t = time.time(); _, rays = orc.render(desc, args.width, args.height, n_passes=n, threads=cores, **kw); total_t += time.time() - t
        total_rays += rays; extra_passes += n; n = min(n * 2, 16)
    value = total_rays / total_t / 1e6
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()      # "max 100000" or "<quota> <period>": the CPU share this container may use
        cpu_quota = None if quota[0] == "max" else round(float(quota[0]) / float(quota[1]), 2)
    except Exception:
        cpu_quota = None
    out = {"value": round(value, 4), "unit": "Mrays/s", "cores": cores, "host_threads": host_threads, "affinity_cpus": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": cpu_quota, "kind": "port",
           "per_core": round(per_core, 4), "effective_cores": round(min(float(cores), cpu_quota or float(cores)), 2),   # threads beyond the container's CPU quota only time-share
           "scaling_efficiency": round(value / (per_core * min(float(cores), cpu_quota or float(cores))), 3),
           "sample": "two-level BVH (the reference's layout); best thread count (calibrated over 8..%d): 1 pass over rows 0..%d of the %dx%d frame + %d more whole-frame passes, depth %d, %d threads, %.1f s wall; "
                     "per-core: 1 thread, rows %d..%d, %.1f s" % (host_threads, done_rows, args.width, args.height, extra_passes, args.depth, cores, total_t, mid, min(args.height, mid + rows1), t1)}
    counts = None
    if flat_desc is not None:
        counts = {}
        rows_c = min(args.height, max(16, int(rows1 * max(1, cores // 4))))
        t = time.time(); _, rays = orc.render(desc, args.width, args.height, n_passes=1, threads=cores, rows=(mid, min(args.height, mid + rows_c)), flat=flat_desc, counts=counts, **kw)
        out["flat_bvh_value"] = round(rays / (time.time() - t) / 1e6, 4)   # the same oracle over the product's flattened BVH (not the reference's layout; for information)
        counts["rows"] = (mid, min(args.height, mid + rows_c))
    return out, counts


def calibrated_traffic(workload_key):
    """HBM-side bytes per ray of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload and kernel build
    (profiles/roofline_traffic.json, written by tools/summarize_profile.py): 2 x FETCH_SIZE + WRITE_SIZE, the x2 calibrated on random 64-B gathers
    (profiles/README.md).  None when the profile is of another workload or another kernel."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None
    e = t.get("workloads", {}).get(workload_key)
    if not e or e.get("kernel_build") != KERNEL_BUILD:
        return None
    return e


def strip_comments(text):
    """C++ source without // and /* */ comments, trailing blanks and empty lines (string and character literals left alone): what source_hash hashes, so that an edit
    to a COMMENT is not another binary and does not throw a committed counter profile away"""
    out = []; i = 0; n = len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i); i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2); i = n if j < 0 else j + 2; out.append(" ")
        else:
            out.append(c); i += 1
    return "\n".join(l.rstrip() for l in "".join(out).split("\n") if l.strip())


def source_hash(names):
    """what ties a committed counter profile to the binary: sha256 over the named files of cudatracerlib_amd/csrc (name + content without comments) and over the compiler flags
    of cudatracerlib_amd/build.py, first 12 hex digits"""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode() + b"\0")
        with open(os.path.join(ROOT, "cudatracerlib_amd", "csrc", n), "rb") as f:
            h.update(strip_comments(f.read().decode("utf-8", "replace")).encode())
        h.update(b"\1")
    h.update(("\0".join(build_flags()) + "\2").encode())
    return h.hexdigest()[:12]


def build_flags():
    """FLAGS of cudatracerlib_amd/build.py, read from the file itself: importing the package here would load libctl_amd.so (and its HIP runtime) at bench.py's import time, before
    the ranks of an N-rank run have chosen their device and before torch brings its own — the ranks then saw no device (found by the 2-rank GPU test)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ctl_build_flags", os.path.join(ROOT, "cudatracerlib_amd", "build.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return list(m.FLAGS)


# Every file that decides what the profiled kernels DO: the kernel and its headers, the builders of the tree it walks (the flattener AND the BVH2 builder with its
# re-optimisation pass), the host code that lays the device scene out (tracer.hip: leaf keys, rough-transmittance rows), the structs both sides share, the measurement knobs,
# every build stub of the shade source.  An edit to the CODE of any of them (comments do not count) without a re-profile (tools/profile_round.sh -> tools/summarize_profile.py ->
# profiles/roofline_traffic.json) turns the roofline's counter-side numbers into "unprofiled" instead of quoting another binary's counters (tests/test_profile_hash.py).
_COMMON_SOURCES = ["tracer.hip", "tracer.h", "device_scene.h", "kernels.h", "knobs.h", "ctl_math.h", "compaction.h"]
TRAVERSAL_SOURCES = ["traverse_flat.h", "flat_slab.h", "flatten.cpp", "flatten.h", "bvh_builder.cpp", "bvh_builder.h", "traverse.h", "traverse_flat8.h", "flat8.h", "kernels.hip"] + _COMMON_SOURCES
SHADE_SOURCES = ["shade_kernel.inc", "shading.h", "bsdf_complex.h", "bsdf_more.h", "bsdf_rough.h", "spline.h", "material_factory.h", "material_textures.h", "mipmap.h", "mip_pyramid.h", "ctl_fmath.h",
                 "shade_basic.hip", "shade_full.hip", "shade_class_a.hip", "shade_class_b.hip", "shade_class_c.hip", "shade_class_g.hip", "shade_class_p.hip",
                 "shade_basic_wf.hip", "shade_full_wf.hip", "shade_class_a_wf.hip", "shade_class_b_wf.hip", "shade_class_c_wf.hip", "shade_class_g_wf.hip", "shade_class_p_wf.hip"] + _COMMON_SOURCES
KERNEL_BUILD = "trav-" + source_hash(TRAVERSAL_SOURCES)
SHADE_BUILD = "shade-" + source_hash(SHADE_SOURCES)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start N ranks of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as
    torch.distributed.run would set them), rank r on device r.  Rank 0's stdout is this process's stdout (ONE JSON line); the other ranks' stdout goes to stderr.
    Non-zero exit if any rank fails or the box has fewer than N devices (CTL_BENCH_SHARE_GPU=1, the 1-GPU test hook, puts every rank on device 0)."""
    import socket
    import subprocess
    n = args.gpus
    if not args.launch_only and os.environ.get("CTL_BENCH_SHARE_GPU") != "1":
        import cudatracerlib_amd as ctl
        have = ctl.device_count()     # hipGetDeviceCount: no context is created in the launcher
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (n, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CTL_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # Watchdog.  ONE deadline for the whole job, counted from the spawn (not from rank 0's exit): a rank wedged in a rendezvous or a collective would otherwise hold the
    # launcher until the caller's own kill, and nothing would say which rank it was.  Rank 0's stdout is drained by a thread so that a full pipe cannot block it.
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read().decode()), daemon=True); reader.start()
    deadline = time.time() + float(args.launch_timeout)
    rcs = [None] * n; verdict = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad and any(rc is None for rc in rcs):
            # a rank died: the others would wait for it in the next barrier until their own time-outs — give them a moment to report, then end them
            grace = time.time() + 20.0
            while time.time() < grace and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            verdict = "rank(s) %s failed (exit codes %s); still running and ended by the launcher: %s" % (bad, [rcs[r] for r in bad], [r for r, p in enumerate(procs) if p.poll() is None])
        elif time.time() > deadline:
            verdict = "no result within --launch-timeout %.0f s; rank(s) still running and ended by the launcher: %s (finished: %s)" % (
                float(args.launch_timeout), [r for r, p in enumerate(procs) if p.poll() is None], {r: rc for r, rc in enumerate(rcs) if rc is not None})
        if verdict:
            for p in procs:
                if p.poll() is None:
                    p.kill()                    # exactly the processes this launcher started
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    reader.join(5.0)
    for line in "".join(out0).splitlines():     # gloo's C++ side prints its connection banner on stdout: everything but the JSON line goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if verdict:
        raise SystemExit("bench.py --gpus %d: %s" % (n, verdict))
    if any(rcs):
        raise SystemExit("bench.py --gpus %d: rank exit codes %s" % (n, rcs))


def launch_only(rank, world, dist):
    """--launch-only: the rendezvous plumbing of an N-rank run without a device — gloo group, barrier, the 128-byte communicator id from rank 0 to everybody,
    the two scalar reductions — and one JSON line from rank 0.  What tests/test_bench_launcher.py runs on the CPU."""
    import torch
    hook = os.environ.get("CTL_BENCH_TEST_RANK_FAULT", "")          # tests/test_bench_launcher.py: "hang:<rank>" / "die:<rank>" — what the launcher's watchdog is for
    if hook == "hang:%d" % rank:
        time.sleep(3600)
    if hook == "die:%d" % rank:
        os._exit(7)
    dist.barrier()
    ident = [bytes((7 * i + 1) & 255 for i in range(128)) if rank == 0 else None]    # stands for ncclGetUniqueId's 128 bytes
    dist.broadcast_object_list(ident, src=0)
    ok = torch.tensor([1 if ident[0] == bytes((7 * i + 1) & 255 for i in range(128)) else 0]); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    r = torch.tensor([1.0], dtype=torch.float64); dist.all_reduce(r, op=dist.ReduceOp.SUM)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world, "ranks_joined": int(r.item()), "max_over_ranks": float(t.item()), "id_broadcast_ok": bool(ok.item()),
                          "self_launched": os.environ.get("CTL_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="passes = samples per pixel (BASELINE: 64 spp)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="synthetic-sm")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--instances", type=int, default=2000)
    ap.add_argument("--subdiv", type=int, default=4)
    ap.add_argument("--hard-scale", default="full", choices=["full", "small"], help="synthetic-sm-hard: 8.4 M (full) or 0.5 M (small) terrain triangles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scene", default=None, metavar="FILE.xml", help="render a Mitsuba-0.5 scene file (ParseMitsubaScene) instead of a built-in workload")
    ap.add_argument("--via-loader", action="store_true", help="write the synthetic workload as a Mitsuba scene and load it through ctl_parse_mitsuba_scene")
    ap.add_argument("--tracer-param", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--no-cache", action="store_true", help="do not use the compiled-geometry cache ($CTL_CACHE_DIR, default $TMPDIR/ctl_amd_cache)")
    ap.add_argument("--flatten", type=int, default=1, help="traverse one world-space BVH over all instanced triangles (64 B of HBM per triangle)")
    ap.add_argument("--reduced-rough-transmittance", action="store_true", help="CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE: rough plastic through the per-material 1-D reduction of the transmittance table (faster; equal to the reference's lookup up to fp32 rounding only)")
    ap.add_argument("--flat-format", default=None, choices=["q4", "q8"], help="node format of the flattened BVH (default: the library's)")
    ap.add_argument("--dump-frame", default=None, metavar="FILE.npy", help="rank 0 saves the reduced PixelData frame (h, w, 7) after the timed region (tests compare N-rank and 1-rank frames)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("CTL_BENCH_LAUNCH_TIMEOUT", "1500")), help="N > 1: seconds from the spawn after which the launcher ends every rank and reports which ones were stuck")
    ap.add_argument("--launch-only", action="store_true", help="N-rank rendezvous plumbing only (no device): spawn, gloo group, id broadcast, reductions; prints n_gpus")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (one rank per GPU: launch N ranks for --gpus N, or let bench.py launch them)" % (args.gpus, world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    share_gpu = os.environ.get("CTL_BENCH_SHARE_GPU") == "1"
    # CTL_BENCH_COMM_WORLD1=1: the N-rank code path with ONE rank (gloo group of one, communicator of one, gather / reduce warm-up, the exchange inside the timed region): every line of
    # the RCCL branch runs on a 1-GPU box against the real librccl.so, so that an 8-GPU node is not the first place where that Python executes (tests/test_gpu_render.py)
    multi = world > 1 or os.environ.get("CTL_BENCH_COMM_WORLD1") == "1"
    if multi and world == 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); free_port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port)); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        # torch.distributed (gloo, CPU tensors) is plumbing only: the barrier, the broadcast of the RCCL unique id and two scalar reductions.  The framebuffer —
        # the one data-path collective — goes through the library's own ncclReduce (ctl_image_reduce, csrc/comm.cpp).
        # CTL_BENCH_SHARE_GPU=1 is a test hook for a 1-GPU box: every rank on device 0 (RCCL refuses two ranks on one device, so the reduce falls back to gloo).
        import torch
        import torch.distributed as dist
        if share_gpu:
            local_rank = 0
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)   # torch.cuda.synchronize() in sync() then waits on this rank's own GPU instead of opening a context on device 0
        dist.init_process_group(backend="gloo")
        if args.launch_only:
            return launch_only(rank, world, dist)
    elif args.launch_only:
        print(json.dumps({"launch_only": True, "n_gpus": 1, "ranks_joined": 1, "self_launched": False}), flush=True)
        return None
    import cudatracerlib_amd as ctl
    if ctl.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    ctl.api._check(ctl.lib.ctl_set_device(local_rank))

    # compiled-geometry cache (the reference's .xmsh role): back-to-back runs and the other ranks of a multi-GPU run load the
    # compiled meshes and the flattened BVH instead of rebuilding them.  Scene build is outside the timed region either way.
    if not args.no_cache:
        ctl.api.set_cache_dir(os.environ.get("CTL_CACHE_DIR") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
    t_build = time.perf_counter()
    if multi and rank != 0 and not args.no_cache:
        dist.barrier()                      # rank 0 compiles and fills the cache first
    sc, scene_source = build_scene(args, rank, dist.barrier if (multi and (args.via_loader or args.workload == "synthetic-sm-hard")) else None)
    desc = sc.desc
    scene = ctl.Scene(desc, flatten=bool(args.flatten), flat_format=args.flat_format, reduced_rough_transmittance=args.reduced_rough_transmittance)
    if multi and rank == 0 and not args.no_cache:
        dist.barrier()
    t_build = time.perf_counter() - t_build
    tr = ctl.WavefrontPathTracer()
    p = tr.getParameters()
    p.setValue("Direct", True); p.setValue("MaxPathLength", args.depth); p.setValue("RRStartDepth", 5)
    for kv in args.tracer_param:                 # build-specific knobs for A/B runs, e.g. --tracer-param SortMaterials=false
        k, v = kv.split("=", 1)
        p.setValue(k, v.lower() == "true" if v.lower() in ("true", "false") else int(v))
    tr.setTileShard(rank, world)
    tr.Resize(args.width, args.height)
    tr.InitializeScene(scene)
    tr.reservePasses(args.steps)      # queue memory for the batch size the timed call will use: allocated here, not inside the timed region
    img = ctl.Image(args.width, args.height)

    comm, reduce_kind, exchange = None, "none (1 GPU)", None
    if multi:
        import torch

        def agree(flag):   # every rank takes the same path: MIN over the ranks of "it worked here" (gloo)
            t = torch.tensor([1 if flag else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN); return int(t.item()) == 1
        # (1) every rank proves that RCCL loads in its process (ncclGetUniqueId is local) BEFORE anybody enters the collective ncclCommInitRank:
        #     a rank that cannot load the library must not leave the others waiting inside it
        why = ""
        try:
            if share_gpu:
                raise RuntimeError("CTL_BENCH_SHARE_GPU: all ranks share device 0")
            my_id = ctl.Comm.unique_id()
        except Exception as e:
            why = str(e)[:120]; my_id = None
        if agree(my_id is not None):
            try:
                ident = [my_id if rank == 0 else None]
                dist.broadcast_object_list(ident, src=0)
                comm = ctl.Comm(ident[0], rank, world, timeout_ms=int(os.environ.get("CTL_BENCH_COMM_TIMEOUT_MS", "90000")))   # ncclCommInitRank with a deadline (comm.cpp)
            except Exception as e:   # never silently: the JSON line says which path ran
                why = str(e)[:120]; comm = None
            if not agree(comm is not None):
                comm = None
        # (2) BOTH exchanges run once here, outside the timed region (the FIRST collective of a communicator sets its connections up and has a time-out of its own):
        #     north_star's gather of the ranks' own tiles first; the whole-frame reduce is the agreed fallback.  A collective that timed out has aborted the communicator
        #     (comm.cpp wait_done), so the fallback of a failed gather is tried on a fresh one.
        if comm is not None:
            scratch = ctl.Image(args.width, args.height)
            for kind in ("gather", "reduce"):
                ok_here = True
                try:
                    if kind == "gather" and os.environ.get("CTL_BENCH_NO_GATHER") == "1":
                        raise RuntimeError("CTL_BENCH_NO_GATHER=1")
                    (comm.gather_to if kind == "gather" else comm.reduce_to)(img, scratch if rank == 0 else None, 0)
                except Exception as e:
                    why = "%s: %s" % (kind, str(e)[:120]); ok_here = False
                if agree(ok_here):
                    exchange = kind
                    break
                if kind == "gather":      # a fresh communicator for the fallback (the old one may be aborted on some ranks)
                    try:
                        comm = None
                        ident = [ctl.Comm.unique_id() if rank == 0 else None]
                        dist.broadcast_object_list(ident, src=0)
                        comm = ctl.Comm(ident[0], rank, world, timeout_ms=int(os.environ.get("CTL_BENCH_COMM_TIMEOUT_MS", "90000")))
                    except Exception as e:
                        why = str(e)[:120]; comm = None
                    if not agree(comm is not None):
                        comm = None
                        break
            del scratch
            if exchange is None:
                comm = None
            elif exchange == "gather":
                reduce_kind = "ncclGather of each rank's own tiles in libctl_amd.so (ctl_image_gather, %d B per rank)" % img.packedTileBytes(world)
            else:
                whys = [None] * world; dist.all_gather_object(whys, why)
                reduce_kind = "ncclReduce of the whole frames in libctl_amd.so (ctl_image_reduce; the gather failed: %s)" % next((x for x in whys if x), "on another rank")
        if comm is None:
            whys = [None] * world; dist.all_gather_object(whys, why)
            why = next((x for x in whys if x), "another rank failed")
            print("bench.py: native RCCL exchange unavailable (%s); falling back to torch.distributed" % why, file=sys.stderr, flush=True)
            exchange = "gloo-gather"
            reduce_kind = "torch.distributed gloo gather of each rank's own tiles through host memory (ctl_image_pack_tiles / _unpack_tiles; native RCCL unavailable: %s)" % why

    def sync():
        ctl.api._check(ctl.lib.ctl_device_synchronize())
        if multi:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dist.barrier()

    def reduce_frame():
        if exchange == "gather":
            comm.gather(img, 0)
        elif exchange == "reduce":
            comm.reduce(img, 0)
        else:   # the same packed tiles, moved by gloo through host memory
            import torch
            mine = torch.from_numpy(img.packTiles(rank, world))
            bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, bufs, dst=0)
            if rank == 0:
                img.unpackTiles(world, torch.stack(bufs).numpy())

    if args.warmup > 0:
        tr.DoPasses(img, args.warmup, new_trace=True)
    sync()
    t0 = time.perf_counter()
    tr.DoPasses(img, args.steps, new_trace=(args.warmup == 0))
    rank_ms = reduce_ms = None
    if multi:
        ctl.api._check(ctl.lib.ctl_device_synchronize())
        rank_ms = (time.perf_counter() - t0) * 1e3      # this rank's own render of its shard
        t_r = time.perf_counter()
        reduce_frame()   # the single framebuffer exchange of the render: PixelData sums -> rank 0 over RCCL / xGMI (includes the wait for the slowest rank)
        reduce_ms = (time.perf_counter() - t_r) * 1e3
    sync()
    elapsed = time.perf_counter() - t0
    st = tr.stats()
    rays = float(st.rays_last_pass)
    k_ms_closest, k_ms_any = st.ms_intersect, st.ms_intersect_any
    n_closest, n_any = int(st.intersect_rays), int(st.shadow_rays)
    launches_closest = int(st.intersect_launches)
    # FuseTraversal (default): bounce d's path rays and bounce d-1's shadow rays share one persistent launch; those launches are the dominant kernel then
    fused_launches, fused_any, fused_closest, k_ms_fused = int(st.fused_launches), int(st.fused_shadow_rays), int(st.fused_closest_rays), st.ms_fused
    if multi:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
        per_rank = [None] * world; dist.all_gather_object(per_rank, {"rank_ms": round(rank_ms, 3), "reduce_ms": round(reduce_ms, 3), "rays": rays})
        r = torch.tensor([rays], dtype=torch.float64); dist.all_reduce(r, op=dist.ReduceOp.SUM); rays = float(r.item())

    out = None
    if rank == 0 and args.dump_frame:
        np.save(args.dump_frame, img.getPixelData())
    if rank == 0:
        # traversal statistics of the SAME rays on the GPU (one extra, untimed batch in counting mode, as many passes per launch as the timed region had — a single
        # pass per launch is mostly ramp and drain and reports a lane utilisation the timed launches do not have): lane utilisation, visited nodes
        tr.setCounting(True)
        counting_passes = max(1, min(args.steps, 32))
        tr.DoPasses(img, counting_passes, new_trace=False)
        cs = tr.stats()
        tr.setCounting(False)
        gpu_counts = {"n_inner": cs.closest_counts.n_inner / max(1, cs.intersect_rays), "n_tri": cs.closest_counts.n_tri / max(1, cs.intersect_rays), "n_inst": cs.closest_counts.n_inst / max(1, cs.intersect_rays)}
        cpu, oc = None, None
        if world == 1 and not args.no_cpu_baseline:
            from cudatracerlib_amd import api as _api
            fb = _api.FlatBvh(desc, _api.FLAT_FORMATS[args.flat_format or _api.DEFAULT_FLAT_FORMAT]) if args.flatten else None
            cpu, oc = cpu_baseline(desc, args, fb.desc if fb is not None else None)
        # algorithmic bytes per ray from the CPU restatement in counting mode over the same BVH; the GPU's own counters when the oracle leg is off
        if oc and oc.get("path_rays"):
            per_ray = {"n_inner": oc["path_inner"] / oc["path_rays"], "n_tri": oc["path_tri"] / oc["path_rays"], "n_inst": oc["path_inst"] / oc["path_rays"]}
            count_source = "oracle (CPU restatement, counting mode, the product's flattened BVH, rows %d..%d of one pass)" % oc["rows"]
        else:
            per_ray = gpu_counts; count_source = "GPU counting kernel (the oracle leg is off)"
        per_ray_closest = 48.0 + 64.0 * per_ray["n_inner"] + 52.0 * per_ray["n_tri"] + 108.0 * per_ray["n_inst"]
        any_visits = {"n_inner": cs.any_counts.n_inner / max(1, cs.shadow_rays), "n_tri": cs.any_counts.n_tri / max(1, cs.shadow_rays), "n_inst": cs.any_counts.n_inst / max(1, cs.shadow_rays)}
        # (the oracle's shadow test is the megakernel's Occluded = a full closest-hit search, KernelDynamicScene.cu:70-80; the wavefront's any-hit traversal stops at the
        #  first hit, so the shadow rays' visits are taken from the any-hit kernel's own counting instantiation)
        per_ray_any = 48.0 + 64.0 * any_visits["n_inner"] + 52.0 * any_visits["n_tri"] + 108.0 * any_visits["n_inst"]
        if fused_launches:
            # dominant kernel = k_intersect_pair: algorithmic bytes of the path rays AND the shadow rays of a launch / average launch duration (HIP events on the tracer's stream)
            avg_launch_ms = k_ms_fused / fused_launches
            rays_per_launch = (fused_closest + fused_any) / fused_launches
            bytes_per_launch = (per_ray_closest * fused_closest + per_ray_any * fused_any) / fused_launches
            records_per_launch = ((per_ray["n_inner"] + per_ray["n_tri"]) * fused_closest + (any_visits["n_inner"] + any_visits["n_tri"]) * fused_any) / fused_launches
            launches_dom = fused_launches
        else:
            # dominant kernel = closest-hit intersect: bytes per launch / average launch duration
            avg_launch_ms = k_ms_closest / max(1, launches_closest)
            rays_per_launch = n_closest / max(1, launches_closest)
            bytes_per_launch = per_ray_closest * rays_per_launch
            records_per_launch = (per_ray["n_inner"] + per_ray["n_tri"]) * rays_per_launch
            launches_dom = launches_closest
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        records_per_s = records_per_launch / (avg_launch_ms * 1e-3) if avg_launch_ms > 0 else 0.0
        wl_key = "%s %dx%d depth %d" % (args.workload, args.width, args.height, args.depth) + ("" if args.workload != "synthetic-sm" else " %d inst subdiv %d" % (args.instances, args.subdiv)) + ((" " + args.hard_scale) if args.workload == "synthetic-sm-hard" else "") + (" flat" if args.flatten else " two-level") + ((" " + args.flat_format) if (args.flatten and args.flat_format and args.flat_format != "q4") else "")
        cal = calibrated_traffic(wl_key)
        traffic = cal["bytes_per_ray"] * rays_per_launch if cal else None
        # The roofline fractions are fractions of PHYSICAL ceilings, all from the counters of the committed profile of THIS kernel build (profiles/<tag>_pmc_summary.csv ->
        # profiles/roofline_traffic.json: per-ray HBM-side bytes, VALU instructions and lane-level vector loads) x the rays of this run's launches / this run's launch time:
        #   hbm        (2 x FETCH_SIZE + WRITE_SIZE) / t / 8 TB/s          <- `frac`, `achieved`, `traffic`: the contract's HBM roofline, by MEASURED traffic
        #   valu_issue SQ_INSTS_VALU / t / (1024 SIMDs x 2.4 GHz / 2)      a lower bound of the VALU pipes' busy time (2 cycles is the fastest a wave64 instruction issues)
        #   l1_lookup  lane-level vector loads / t / 0.66 T/s               the scattered 16-B lane-loads the vector L1s sustain (tools/gather_probe.hip)
        # `bound` names the largest.  SURVEY 8d's algorithmic figure (cache-blind: it charges HBM for every node visit, and passes 1) stays as algorithmic_*.
        t_launch = avg_launch_ms * 1e-3
        fractions = {}
        if cal and t_launch > 0:
            fractions["hbm"] = cal["bytes_per_ray"] * rays_per_launch / t_launch / (HBM_PEAK_GBS * 1e9)
            if cal.get("valu_insts_per_ray"):
                fractions["valu_issue"] = cal["valu_insts_per_ray"] * rays_per_launch / t_launch / (1024 * 2.4e9 / 2.0)
            if cal.get("lane_loads_per_ray"):
                fractions["l1_lookup"] = cal["lane_loads_per_ray"] * rays_per_launch / t_launch / 0.66e12
        bound = max(fractions, key=fractions.get) if fractions else "unprofiled"
        hbm_frac = fractions.get("hbm")
        roof = {"bound": bound, "kernel": "k_intersect<closest>" if not fused_launches else "k_intersect_pair (closest hits of bounce d + occlusion of bounce d-1 in one persistent launch)",
                "achieved": round(traffic / t_launch / 1e9, 2) if traffic and t_launch > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(hbm_frac, 4) if hbm_frac is not None else None, "traffic": traffic,
                "fractions": {k: round(v, 4) for k, v in fractions.items()}, "bound_frac": round(fractions[bound], 4) if fractions else None,
                "fractions_note": "each = per-ray counter quantity of the committed profile of this kernel build x this run's rays per launch / this run's launch time / ceiling (hbm 8 TB/s; valu_issue 1024 SIMDs x 2.4 GHz / 2 cycles, a lower bound; l1_lookup 0.66 T scattered lane-loads/s); frac = fractions.hbm",
                "algorithmic_achieved": round(achieved, 2), "algorithmic_frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_note": "SURVEY 8d: (48 + 64 N_inner + 52 N_tri) B per ray x rays / launch time: cache-blind, L2 and the Infinity Cache serve over half of those records, so it can pass 1",
                "saturated": bool(fractions and max(fractions.values()) >= 0.8),
                "valu_lane_utilisation_profiled": cal.get("valu_lane_utilisation") if cal else None, "wait_any_frac_profiled": cal.get("wait_any_frac") if cal else None, "clock_ghz_profiled": cal.get("clock_ghz") if cal else None,
                "l2_hit_rate": cal.get("l2_hit_rate") if cal else None, "traffic_profile": cal.get("tag") if cal else None, "workload_key": wl_key, "kernel_build": KERNEL_BUILD,
                "bytes_per_ray": round(bytes_per_launch / max(1.0, rays_per_launch), 1), "bytes_per_path_ray": round(per_ray_closest, 1), "bytes_per_shadow_ray": round(per_ray_any, 1),
                "per_ray": {k: round(v, 2) for k, v in per_ray.items()}, "per_shadow_ray": {k: round(v, 2) for k, v in any_visits.items()}, "per_ray_source": count_source, "per_shadow_ray_source": "GPU any-hit counting kernel",
                "per_ray_gpu_visited": {k: round(v, 2) for k, v in gpu_counts.items()},
                "records_per_s": round(records_per_s / 1e9, 2), "records_per_s_unit": "G node+leaf records/s (gather ceilings, tools/gather_probe.hip: ~58 G/s from HBM, ~170 G/s from L2)",
                "lane_utilisation": {"inner": round(cs.closest_counts.n_inner / max(1, 64 * cs.closest_counts.wave_inner_iters), 3), "tri": round(cs.closest_counts.n_tri / max(1, 64 * cs.closest_counts.wave_tri_iters), 3)},
                "rays_per_launch": int(rays_per_launch), "path_rays_per_launch": int(fused_closest / fused_launches) if fused_launches else int(rays_per_launch),
                "avg_launch_ms": round(avg_launch_ms, 4), "launches": launches_dom,
                "separate_launches": {"closest_hit": {"rays": n_closest - fused_closest, "ms": round(k_ms_closest, 3), "launches": launches_closest},
                                      "any_hit": {"rays": n_any - fused_any, "ms": round(k_ms_any, 3)}},
                "closest_rays_total": n_closest, "counting_passes": counting_passes,   # (a profile of this command also sees the counting batch: its shade launches are the product's, its traversal launches the COUNT instantiations)
                "ms_intersect": round(k_ms_closest + k_ms_fused + k_ms_any, 3), "ms_shade": round(st.ms_shade, 3), "ms_raygen": round(st.ms_raygen, 3)}
        # the second kernel of the step: shading (k_shade_basic / k_shade_full), priced the same way per shaded path vertex (= closest-hit ray)
        roof_shade = None
        sh = cal.get("shade") if cal else None
        if sh and sh.get("kernel_build") != SHADE_BUILD:
            sh = None           # the profile is of other shade sources
        if sh and sh.get("bytes_per_vertex") and st.ms_shade > 0:
            t_sh = st.ms_shade * 1e-3
            fs = {"hbm": sh["bytes_per_vertex"] * n_closest / t_sh / (HBM_PEAK_GBS * 1e9)}
            if sh.get("valu_insts_per_vertex"):
                fs["valu_issue"] = sh["valu_insts_per_vertex"] * n_closest / t_sh / (1024 * 2.4e9 / 2.0)
            if sh.get("lane_loads_per_vertex"):
                fs["l1_lookup"] = sh["lane_loads_per_vertex"] * n_closest / t_sh / 0.66e12
            bs = max(fs, key=fs.get)
            roof_shade = {"kernel": sh["kernel"], "bound": bs, "bound_frac": round(fs[bs], 4), "fractions": {k: round(v, 4) for k, v in fs.items()},
                          "achieved": round(sh["bytes_per_vertex"] * n_closest / t_sh / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fs["hbm"], 4),
                          "traffic_per_vertex": round(sh["bytes_per_vertex"], 1),
                          "algorithmic_bytes_per_vertex": 104 + 16 + 20 + 136 + 32 + 48, "algorithmic_note": "SURVEY 8d shading list: path state 104 B + hit 16 + 4 read, <= 104 + 32 written, TriangleData 32 B, instance rows 48 B (+ material / light records, L2-resident)",
                          "vertices": n_closest, "ms": round(st.ms_shade, 3), "valu_lane_utilisation_profiled": sh.get("valu_lane_utilisation"), "kernel_build": SHADE_BUILD}
        elif st.ms_shade > 0:
            roof_shade = {"bound": "unprofiled", "kernel_build": SHADE_BUILD, "vertices": n_closest, "ms": round(st.ms_shade, 3),
                          "note": "no counter profile of this workload with these shade sources in profiles/roofline_traffic.json (tools/profile_round.sh writes one)"}
        out = {
            "metric": "Mrays/s at %dx%d, %d spp (steps), depth-%d; achieved HBM GB/s vs peak" % (args.width, args.height, args.steps, args.depth),
            "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic-SM %dx%d, 1 spp/step, depth %d, NEE on, %d instances x icosphere(%d)/boxes, %d instanced triangles"
                       % (args.width, args.height, args.depth, args.instances, args.subdiv, int(_instanced_tris(desc))) if args.workload == "synthetic-sm" else
                       ("%s %dx%d depth %d, %d instanced triangles" % (args.workload, args.width, args.height, args.depth, int(_instanced_tris(desc)))),
                       "bvh": "flattened world-space BVH4 (64 B nodes with 8-bit quantised child boxes; 128 B leaf entries evaluated with the reference's object-space arithmetic; triangles much longer than their neighbours entered as several references; BVH2 re-optimised by reinsertion before the collapse)" if args.flatten else "two-level (scene BVH + instanced mesh BVHs)",
                       "scene_source": scene_source,
                       "rough_transmittance": "per-material 1-D reduction (CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE: equal to the reference's lookup up to fp32 rounding)" if args.reduced_rough_transmittance else "the reference's 3-D lookup on every call (bit-equal frames)",
                       "parallelism": "image tiles 64x64 round-robin over %d GPU(s), 1 gather of the framebuffer per render" % world, "framebuffer_reduce": reduce_kind,
                       "rays_per_step": int(rays / args.steps), "scene_build_s": round(t_build, 2)},
            "roofline": roof,
        }
        if roof_shade:
            out["roofline_shade"] = roof_shade
        if multi:
            slow = max(range(world), key=lambda r: per_rank[r]["rank_ms"])
            out["rank_ms"] = [q["rank_ms"] for q in per_rank]; out["reduce_ms"] = per_rank[0]["reduce_ms"]; out["reduce_ms_per_rank"] = [q["reduce_ms"] for q in per_rank]
            out["slowest_rank"] = slow; out["rays_per_rank"] = [int(q["rays"]) for q in per_rank]
            out["rank_ms_note"] = "rank_ms: a rank's own render of its tile shard (DoPasses + device sync); reduce_ms: the framebuffer reduce as the root saw it, including its wait for the slowest rank"
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return out


def _instanced_tris(desc):
    nodes = desc.view("nodes", np.uint32, desc.n_nodes, 6)
    meshes = desc.view("meshes", np.uint32, desc.n_meshes, 5)
    # triangles per mesh = difference of consecutive tri offsets (meshes are appended in order)
    offs = np.append(meshes[:, 0], desc.n_tri_data)
    per_mesh = np.diff(offs)
    return per_mesh[nodes[:, 0]].sum()


if __name__ == "__main__":
    main()
