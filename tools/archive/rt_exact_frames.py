#!/usr/bin/env python3
"""bit-equal pixel fractions GPU vs oracle (shared-math build) of scenes with rough plastic: the bathroom miniature and fuzz seeds (RT_MODE = rows / generic / reduced, tools/archive/r05_rt_exact_probe.sh)"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as gpu
from cudatracerlib_amd import scenes
import oracle
orc = oracle.Oracle(shared_math=True)
cases = [("bathroom 96x64", scenes.synthetic_bathroom(96, 64, n_instances=40, subdiv=2), 96, 64)] + [("fuzz %d" % s, scenes.fuzz_scene(s, 96, 64), 96, 64) for s in (200, 230, 110, 143)]
for name, sc, W, H in cases:
    d = sc.desc; P = 3
    tables = orc.sequence_tables(P)
    want, _ = orc.render(d, W, H, n_passes=P, tables=tables, max_path_length=8, rr_start=5)
    tr = gpu.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 8)
    tr.Resize(W, H); tr.InitializeScene(gpu.Scene(d, flatten=True, reduced_rough_transmittance=os.environ.get("RT_MODE") == "reduced")); img = gpu.Image(W, H)
    for k in range(P):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData(); g, w = got[..., :3], want[..., :3]
    off = ~(np.abs(g - w) <= 2e-3 * (1 + np.abs(w))).all(axis=2)
    print(json.dumps({"scene": name, "exact": round(float((g == w).all(axis=2).mean()), 4), "off_pixels": int(off.sum()), "weights_equal": bool(np.array_equal(got[..., 6], want[..., 6]))}), flush=True)
