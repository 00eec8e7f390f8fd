#!/usr/bin/env python3
"""Where does the full shade kernel spend its time?  Renders the synthetic-bathroom workload with groups of BSDF models swapped for
plain diffuse and prints the shade-kernel time per pass (HIP events).  Usage: python tools/shade_probe.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api, scenes

W, H, STEPS = 1920, 1080, 8
real = {k: getattr(api, k) for k in ("roughplastic", "roughdielectric", "roughconductor", "coating", "roughdiffuse", "dielectric")}


def run(tag, swap):
    for k, f in real.items():
        setattr(api, k, (lambda *a, **kw: api.diffuse((0.5, 0.5, 0.5))) if k in swap else f)
    sc = scenes.synthetic_bathroom(W, H)
    scene = ctl.Scene(sc.desc, flatten=True)
    tr = ctl.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 8)
    tr.Resize(W, H); tr.InitializeScene(scene)
    img = ctl.Image(W, H)
    tr.DoPasses(img, 2, new_trace=True)
    tr.DoPasses(img, STEPS, new_trace=False)
    api._check(ctl.lib.ctl_device_synchronize())
    st = tr.stats()
    print(json.dumps({"variant": tag, "ms_shade": round(st.ms_shade / STEPS, 2), "ms_intersect": round(st.ms_intersect / STEPS, 2), "ms_shadow": round(st.ms_intersect_any / STEPS, 2),
                      "Mrays_per_pass": round(st.rays_last_pass / STEPS / 1e6, 2)}), flush=True)


run("as is", ())
run("rough plastics -> diffuse", ("roughplastic",))
run("+ rough glass / glass -> diffuse", ("roughplastic", "roughdielectric", "dielectric"))
run("+ coating, Oren-Nayar -> diffuse", ("roughplastic", "roughdielectric", "dielectric", "coating", "roughdiffuse"))
run("everything diffuse (textures, height map, env emitter stay)", tuple(real))
