#!/usr/bin/env python3
"""k_shade_full against the model-class launches (shade_class_*.hip) on synthetic-bathroom 1080p depth 8 and on its single-class variants ("basic": every material of class a,
"single": every material of class b) — on those a class launch has full waves by construction, so the difference is what the per-class register allocation is worth.
Usage: python tools/shade_class_probe.py [sets, default all,basic,single]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api, scenes

W, H, STEPS = 1920, 1080, 20
ctl.api.set_cache_dir(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
for mset in (sys.argv[1] if len(sys.argv) > 1 else "all,basic,single").split(","):
    sc = scenes.synthetic_bathroom(W, H, material_set=mset)
    scene = ctl.Scene(sc.desc, flatten=True)
    for by_class in ((True,) if os.environ.get("PROBE_CLASS_ONLY") == "1" else (True, False)):
        tr = ctl.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 8); p.setValue("ShadeByModelClass", by_class)
        tr.Resize(W, H); tr.InitializeScene(scene); tr.reservePasses(STEPS)
        img = ctl.Image(W, H)
        tr.DoPasses(img, 5, new_trace=True)
        tr.DoPasses(img, STEPS, new_trace=False)
        api._check(ctl.lib.ctl_device_synchronize())
        st = tr.stats()
        print(json.dumps({"lib": os.path.basename(os.environ.get("CTL_AMD_LIB", "libctl_amd.so")), "materials": mset, "ShadeByModelClass": by_class, "ms_shade_per_pass": round(st.ms_shade / STEPS, 3),
                          "ms_intersect": round((st.ms_intersect + st.ms_fused + st.ms_intersect_any) / STEPS, 3), "Mrays_per_pass": round(st.rays_last_pass / STEPS / 1e6, 2),
                          "path_vertices_per_pass_M": round(st.intersect_rays / STEPS / 1e6, 2)}), flush=True)
