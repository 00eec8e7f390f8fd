#!/usr/bin/env python3
"""One-off wide parity fuzz on the GPU box: seeds A..B of scenes.fuzz_scene at 96x64, 4 passes, depth 8, both BVH layouts (+ the Wavefront path rules on every fourth seed), GPU vs the
oracle's shared-math build on the same sampler tables.  Prints one line per seed that is not clean and a summary.  Usage: python tools/fuzz_sweep.py [first] [last]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as gpu
from cudatracerlib_amd import scenes
import oracle

W, H, PASSES, DEPTH, RR = 96, 64, 4, 8, 5
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 300)
MODE = sys.argv[3] if len(sys.argv) > 3 else "default"      # default | wavefront (PathSemantics = Wavefront, +u16 on odd seeds) | plugin (the megakernel PathTracer: first-hit ray differentials) | sensors (thin lens / orthographic / telecentric / spherical by seed) | alpha (alpha maps on a third of the materials, AlphaTest = true) | nodirect (Direct = false, depth 5, RRStartDepth 2) | wild (textured roughness / exponents / specular colours, two-sidedness, texture offsets varied in place)
orc = oracle.Oracle(shared_math=True)
worst = []; n_bad = 0; zs_all = []
for seed in range(first, last):
    sc = scenes.fuzz_scene(seed, W, H); d = sc.desc
    if MODE == "sensors":
        cam = gpu.api.ctl_sensor.from_buffer_copy(d.camera)
        kind = seed % 4
        cam.type = (3, 4, 5, 1)[kind]
        cam.aperture_radius, cam.focus_distance = (0.2, 12.0) if kind == 0 else (0.05, 8.0)
        cam.screen_scale[:] = [6.0, 6.0]
        if kind in (1, 2): cam.near_depth, cam.far_depth = 1e-5, 1e5
        sc.setSensor(cam); sc.UpdateScene(); d = sc.desc
    if MODE == "alpha":
        rs = np.random.RandomState(seed)
        for i in range(d.n_materials):
            if rs.randint(3) == 0 and d.materials[i].bsdf_type not in (3, 4, 5):
                m = d.materials[i]
                k = rs.randint(3)
                t = gpu.api.checker_texture(1.0, 0.0, uv_scale=(float(rs.choice([2.0, 4.0, 7.0])), float(rs.choice([2.0, 3.0])))) if k < 2 else gpu.api.checker_texture((0.9, 0.1, 0.1), (0.1, 0.1, 0.9), uv_scale=(3.0, 3.0))
                m.alpha_state = 1 if k < 2 else 3; m.alpha_test_scalar = 0.5 if k < 2 else 0.25; m.alpha_test_color[:] = [1.0, 0.0, 0.0]; m.alpha_tex = t
    if MODE == "wild":      # parameters the scene generator leaves constant, varied in place (both sides read the same records): textured roughness / exponents / specular colours, two-sidedness, texture offsets
        rs = np.random.RandomState(7000 + seed)
        def chk(a, b):
            return gpu.api.checker_texture(a, b, uv_scale=(float(rs.choice([1.0, 3.0, 6.0])), float(rs.choice([2.0, 5.0]))), uv_offset=(float(rs.uniform(0, 1)), float(rs.uniform(0, 1))))
        slots = {7: (1, 2), 5: (2, 3), 9: (2,), 14: (2,), 2: (1,), 11: (2, 3)}
        for i in range(d.n_materials):
            m = d.materials[i]; t = m.bsdf_type
            if t in slots and rs.randint(2):
                for k in slots[t]:
                    a0, a1 = float(rs.uniform(0.03, 0.5)), float(rs.uniform(0.03, 0.5))
                    m.tex[k] = chk((a0, a0, a0), (a1, a1, a1))
            if t == 10 and rs.randint(2): e0, e1 = float(rs.uniform(3, 150)), float(rs.uniform(3, 150)); m.tex[2] = chk((e0, e0, e0), (e1, e1, e1))
            if t in (6, 7) and rs.randint(2): m.tex[0] = chk(tuple(rs.uniform(0.3, 1.0, 3)), tuple(rs.uniform(0.3, 1.0, 3)))
            if t not in (3, 4, 5) and rs.randint(3) == 0: m.two_sided = 1
            for k in range(4):
                if m.tex[k].type in (3, 4) and rs.randint(2): m.tex[k].uv_offset[:] = [float(rs.uniform(-1, 1)), float(rs.uniform(-1, 1))]
    tables = orc.sequence_tables(PASSES)
    kw = {}
    if MODE == "wavefront": kw = dict(wavefront_rules=True, u16_barycentrics=bool(seed & 1))
    elif MODE == "plugin": kw = dict(partials=True)
    elif MODE == "alpha": kw = dict(alpha_test=True)
    elif MODE == "nodirect": kw = dict(direct=False)
    depth_, rr_ = (5, 2) if MODE == "nodirect" else (DEPTH, RR)
    # round 6: the oracle also returns the samples the reference drops after a path's throughput died, as the radiance the kernels' zero-throughput cut counts (zero_stop):
    # the kernels' frame is held to frame + zero_stop in EVERY pixel, weights exactly (rounds 5's sweeps masked the pixels whose weights differ)
    zero_stop = np.zeros((H, W, 7), np.float32)
    want, want_rays = orc.render(d, W, H, n_passes=PASSES, tables=tables, max_path_length=depth_, rr_start=rr_, zero_stop=zero_stop, **kw)
    n_zero_stop = int(zero_stop[..., 6].sum()); untouched = zero_stop[..., 6] == 0
    want = want + zero_stop
    for flatten in ((True,) if MODE == "plugin" else (False, True)):
        scene = gpu.Scene(d, flatten=flatten)
        tr = gpu.PathTracer() if MODE == "plugin" else gpu.WavefrontPathTracer()
        p = tr.getParameters(); p.setValue("MaxPathLength", depth_); p.setValue("RRStartDepth", rr_)
        if MODE == "alpha": p.setValue("AlphaTest", True)
        if MODE == "nodirect": p.setValue("Direct", False)
        if MODE == "wavefront": p.setValue("PathSemantics", "Wavefront"); p.setValue("U16Barycentrics", bool(seed & 1))
        tr.Resize(W, H); tr.InitializeScene(scene); img = gpu.Image(W, H)
        for k in range(PASSES):
            tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
        got = img.getPixelData()
        g, w = got[..., :3], want[..., :3]
        same_w = got[..., 6] == want[..., 6]      # every pixel's weight must equal the oracle's + its zero-stop samples
        bad_w = bool((~same_w).any())
        off = ~(np.abs(g - w) <= 2e-3 * (1 + np.abs(w))).all(axis=2)
        exact = float((g == w).all(axis=2)[untouched].mean())
        rel = abs(float(g.mean()) - float(w.mean())) / max(float(w.mean()), 1e-9)
        fin = bool(np.isfinite(g).all())
        rec = {"seed": seed, "flat": flatten, "off_pixels": int(off.sum()), "exact": round(exact, 4), "mean_rel": rel, "weights_differ_pixels": int((~same_w).sum()), "weights_bad": bad_w, "zero_stop_samples": n_zero_stop, "finite": fin,
               "models": sorted(set(d.materials[i].bsdf_type for i in range(d.n_materials)))}
        worst.append((int(off.sum()), rel, seed, flatten)); zs_all.append(n_zero_stop)
        if off.sum() > 0 or rel > 1e-3 or bad_w or not fin or exact < 0.98:
            n_bad += 1
            ys, xs = np.nonzero(off)
            rec["where"] = [[int(x), int(y)] for x, y in zip(xs[:4], ys[:4])]; rec["gpu"] = g[off][:2].tolist(); rec["cpu"] = w[off][:2].tolist()
            print(json.dumps(rec), flush=True)
worst.sort(reverse=True)
print(json.dumps({"zero_stop_samples_total": int(sum(zs_all)), "renders_with_zero_stop": int(sum(1 for z in zs_all if z))}), flush=True)
print(json.dumps({"mode": MODE, "seeds": [first, last], "renders": len(worst), "not_clean": n_bad, "worst_off_pixels": worst[:5]}), flush=True)
