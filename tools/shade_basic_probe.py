#!/usr/bin/env python3
"""Where does k_shade_basic spend its time (synthetic-SM 1080p depth 8)?  Shade ms per pass with parts of the work taken away at scene / parameter level; library variants
(CTL_AMD_LIB) take code away.  Usage: python tools/shade_basic_probe.py [tag]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api, scenes

W, H, STEPS = 1920, 1080, 20
real_mat = scenes._material_of


def run(tag, all_diffuse=False, **params):
    scenes._material_of = (lambda m: api.diffuse(m[1] if m[0] == "diffuse" else (0.6, 0.6, 0.6))) if all_diffuse else real_mat
    sc = scenes.synthetic_sm(W, H)
    scene = ctl.Scene(sc.desc, flatten=True)
    tr = ctl.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 8)
    for k, v in params.items():
        p.setValue(k, v)
    tr.Resize(W, H); tr.InitializeScene(scene); tr.reservePasses(STEPS)
    img = ctl.Image(W, H)
    tr.DoPasses(img, 5, new_trace=True)
    tr.DoPasses(img, STEPS, new_trace=False)
    api._check(ctl.lib.ctl_device_synchronize())
    st = tr.stats()
    print(json.dumps({"lib": os.path.basename(os.environ.get("CTL_AMD_LIB", "libctl_amd.so")), "variant": tag, "ms_shade_per_pass": round(st.ms_shade / STEPS, 3), "ms_intersect": round((st.ms_intersect + st.ms_fused + st.ms_intersect_any) / STEPS, 3),
                      "Mrays_per_pass": round(st.rays_last_pass / STEPS / 1e6, 2), "path_vertices_per_pass_M": round(st.intersect_rays / STEPS / 1e6, 2)}), flush=True)


ctl.api.set_cache_dir(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
run("as is")
if os.environ.get("PROBE_SCENE_VARIANTS", "1") == "1":
    run("every material diffuse", all_diffuse=True)
    run("Direct = false (no NEE)", Direct=False)
    run("every material diffuse, no NEE", all_diffuse=True, Direct=False)
    run("atomics instead of the ordered stage", OrderedAccumulation=False)
