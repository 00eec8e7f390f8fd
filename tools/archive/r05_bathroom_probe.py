#!/usr/bin/env python3
"""What the rough-plastic class of synthetic-bathroom waits for: shade ms per pass with parts of the floor's material taken away at scene level; library
variants (CTL_AMD_LIB) move the reduced transmittance tables into LDS.  Usage: python tools/archive/r05_bathroom_probe.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api, scenes

W, H, STEPS = 1920, 1080, 20
real_rp, real_hm, real_img = api.roughplastic, api.set_height_map, api.image_texture


def run(tag, no_height=False, constant_floor=False, plastic_as_diffuse=False):
    api.set_height_map = (lambda m, t: None) if no_height else real_hm
    api.image_texture = (lambda image, *a, **k: (0.6, 0.6, 0.55) if image == 0 else real_img(image, *a, **k)) if constant_floor else real_img   # image 0 = the floor's tiles, 1 = its height map
    api.roughplastic = (lambda refl=(0.5, 0.5, 0.5), **k: api.diffuse(refl)) if plastic_as_diffuse else real_rp
    try:
        sc = scenes.synthetic_bathroom(W, H)
    finally:
        api.set_height_map, api.image_texture, api.roughplastic = real_hm, real_img, real_rp
    scene = ctl.Scene(sc.desc, flatten=True)
    tr = ctl.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 8)
    tr.Resize(W, H); tr.InitializeScene(scene); tr.reservePasses(STEPS)
    img = ctl.Image(W, H)
    tr.DoPasses(img, 5, new_trace=True); tr.DoPasses(img, STEPS, new_trace=False)
    api._check(ctl.lib.ctl_device_synchronize())
    st = tr.stats()
    print(json.dumps({"lib": os.path.basename(os.environ.get("CTL_AMD_LIB", "libctl_amd.so")), "variant": tag, "ms_shade_per_pass": round(st.ms_shade / STEPS, 3),
                      "ms_intersect_per_pass": round((st.ms_intersect + st.ms_fused + st.ms_intersect_any) / STEPS, 3), "Mrays_per_pass": round(st.rays_last_pass / STEPS / 1e6, 2)}), flush=True)


ctl.api.set_cache_dir(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
run("as is")
if os.environ.get("PROBE_SCENE_VARIANTS", "1") == "1":
    run("floor without its height map", no_height=True)
    run("floor with a constant diffuse colour instead of the bitmap (height map kept)", constant_floor=True)
    run("floor: neither", no_height=True, constant_floor=True)
    run("every rough plastic replaced by diffuse (same colours, maps kept)", plastic_as_diffuse=True)
