#!/usr/bin/env python3
"""Diagnosis of a parity-fuzz seed on the GPU box: single-pass frames GPU vs oracle (shared-math build) -> the samples that differ; for each, the first path length at which they
differ and the oracle's vertex-by-vertex log of that sample (oracle_capi.cpp orc_path_log).  Usage: python tools/fuzz_diag.py SEED [SEED ...]"""
import os, sys, json, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as gpu
from cudatracerlib_amd import scenes
import oracle

W, H, PASSES, DEPTH, RR = 96, 64, 4, 8, 5
orc = oracle.Oracle(shared_math=True)
lib = orc.lib
lib.orc_path_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
MODEL = {1: "diffuse", 2: "roughdiffuse", 3: "dielectric", 4: "thindielectric", 5: "roughdielectric", 6: "conductor", 7: "roughconductor", 8: "plastic", 9: "roughplastic", 10: "phong", 11: "ward", 13: "coating", 14: "roughcoating", 15: "blend"}


def gpu_pass(scene, tab, depth):
    tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", depth); p.setValue("RRStartDepth", RR)
    tr.Resize(W, H); tr.InitializeScene(scene); img = gpu.Image(W, H)
    tr.setSamplerTables(*tab); tr.DoPass(img, new_trace=True)
    return img.getPixelData()


for seed in [int(a) for a in sys.argv[1:]]:
    sc = scenes.fuzz_scene(seed, W, H); d = sc.desc
    scene = gpu.Scene(d, flatten=os.environ.get("FLAT", "1") == "1"); scene2 = gpu.Scene(d, flatten=False); fb = gpu.api.FlatBvh(d, gpu.api.FLAT_Q4)
    tables = orc.sequence_tables(PASSES)
    found = []
    for k in range(PASSES):
        want, _ = orc.render(d, W, H, n_passes=1, tables=tables[k:k + 1], max_path_length=DEPTH, rr_start=RR)
        got = gpu_pass(scene, tables[k], DEPTH)
        off = ~(np.abs(got[..., :3] - want[..., :3]) <= 2e-3 * (1 + np.abs(want[..., :3]))).all(axis=2) | (got[..., 6] != want[..., 6])
        for y, x in zip(*np.nonzero(off)):
            found.append((k, int(x), int(y), got[y, x].tolist(), want[y, x].tolist()))
    print(json.dumps({"seed": seed, "differing_samples": len(found)}), flush=True)
    for k, x, y, g, w in found[:4]:
        per_len = []
        for L in range(1, DEPTH + 1):
            wl, _ = orc.render(d, W, H, n_passes=1, tables=tables[k:k + 1], max_path_length=L, rr_start=RR, rows=(y, y + 1))
            gl = gpu_pass(scene, tables[k], L)
            per_len.append((L, [round(float(v), 6) for v in gl[y, x, :3]] + [float(gl[y, x, 6])], [round(float(v), 6) for v in wl[y, x, :3]] + [float(wl[y, x, 6])]))
        log = np.zeros(26 * 16, np.float32); rgb = np.zeros(3, np.float32)
        t1, t2 = tables[k]
        n = lib.orc_path_log(C.addressof(d), W, H, t1.ctypes.data, t2.ctypes.data, x, y, 1, DEPTH, RR, log.ctypes.data, len(log), rgb.ctypes.data)
        print(json.dumps({"pass": k, "pixel": [x, y], "gpu": g, "cpu": w, "oracle_sample": rgb.tolist()}))
        for L, a, b in per_len:
            print("   len %d  gpu %s  cpu %s %s" % (L, a, b, "" if np.allclose(a, b, rtol=2e-3, atol=2e-3) else "  <-- differ"))
        for r in log[:n].reshape(-1, 26):
            m = d.materials[int(r[3])]
            extra = ""
            if int(r[4]) in (13, 14, 15):
                extra = " nested " + "/".join(MODEL.get(d.materials[int(m.u[i])].bsdf_type, "?") for i in ((2, 3) if int(r[4]) == 15 else (2,)))
            print("   v%d tri %d node %d mat %d %s%s map %d tex0 type %d light %d | f %s pdf %.6g sampled 0x%x | cf %s cl %s | t %.6g uv %.4f %.4f" % (
                r[0], r[1], r[2], r[3], MODEL.get(int(r[4]), "?"), extra, m.map_kind, m.tex[0].type, r[5], np.round(r[6:9], 5).tolist(), r[9], int(r[10]), np.round(r[11:14], 5).tolist(), np.round(r[14:17], 5).tolist(), r[17], r[18], r[19]))
            # the same ray through ctl_intersect (both layouts) and the oracle's two traversals
            ray = np.zeros((1, 8), np.float32); ray[0, :3] = r[20:23]; ray[0, 3] = d.ray_trace_eps; ray[0, 4:7] = r[23:26]; ray[0, 7] = np.float32(3.402823466e+38)
            hits = {"gpu (the rendering layout)": gpu.intersect(scene, ray), "gpu two-level": gpu.intersect(scene2, ray), "oracle two-level": orc.intersect(d, ray), "oracle flat": orc.intersect(d, ray, flat=fb.desc)}
            print("      ray o %s d %s | %s" % (r[20:23].tolist(), r[23:26].tolist(), " | ".join("%s: tri %d node %d t %.9g u %.6f v %.6f" % (k, h["tri_idx"][0], h["node_idx"][0], h["dist"][0], h["u"][0], h["v"][0]) for k, h in hits.items())))
