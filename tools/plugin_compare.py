#!/usr/bin/env python3
"""Time the two integrator plugins (WavefrontPathTracer / PathTracer megakernel) on the bench workload — the same comparison the
reference's own WavefrontPathTracer vs PathTracer pair invites (Integrators/PathTracer.cu vs PseudoRealtime/WavefrontPathTracer.cu).
Usage: python tools/plugin_compare.py [--steps 8] [--width 1920 --height 1080]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="synthetic-sm")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--instances", type=int, default=2000)
    ap.add_argument("--subdiv", type=int, default=4)
    args = ap.parse_args()
    args.scene = None; args.via_loader = False
    import cudatracerlib_amd as ctl
    ctl.api.set_cache_dir(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
    sc, _ = bench.build_scene(args)
    scene = ctl.Scene(sc.desc, flatten=True)
    res = {}
    for cls in (ctl.WavefrontPathTracer, ctl.PathTracer):
        tr = cls()
        p = tr.getParameters(); p.setValue("Direct", True); p.setValue("MaxPathLength", args.depth); p.setValue("RRStartDepth", 5)
        tr.Resize(args.width, args.height); tr.InitializeScene(scene)
        img = ctl.Image(args.width, args.height)
        tr.DoPasses(img, args.warmup, new_trace=True)
        ctl.api._check(ctl.lib.ctl_device_synchronize())
        t0 = time.perf_counter()
        tr.DoPasses(img, args.steps, new_trace=False)
        ctl.api._check(ctl.lib.ctl_device_synchronize())
        dt = time.perf_counter() - t0
        rays = float(tr.stats().rays_last_pass)
        res[cls.__name__] = {"Mrays/s": round(rays / dt / 1e6, 1), "ms_per_pass": round(dt * 1e3 / args.steps, 3), "rays_per_pass": int(rays / args.steps),
                             "mean": float(img.getPixelData()[..., :3].mean())}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
