#!/usr/bin/env python3
"""One material of a fuzz scene alone: a floor quad and a sphere of it under a point light and an area light, GPU vs oracle (shared-math build), and the same with parts of the
material taken away (texture -> constant, map off, flags off) to see which part carries a difference.  Usage: python tools/fuzz_material_probe.py SEED MATERIAL_INDEX"""
import os, sys, json, copy, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudatracerlib_amd as gpu
from cudatracerlib_amd import api, scenes
import oracle

W, H, PASSES = 64, 48, 2
seed, mi = int(sys.argv[1]), int(sys.argv[2])
orc = oracle.Oracle(shared_math=True)
src = scenes.fuzz_scene(seed, 96, 64); sd = src.desc
M0 = api.ctl_material.from_buffer_copy(C.string_at(C.addressof(sd.materials[mi]), C.sizeof(api.ctl_material)))
if os.environ.get("MAT") == "diffuse_checker":
    M0 = api.diffuse(api.checker_texture((0.63, 0.6, 0.89), (0.16, 0.65, 0.1), uv_scale=(4.0, 2.0)))
elif os.environ.get("MAT") == "roughplastic_checker_linear":
    M0.u[0] = 0
elif os.environ.get("MAT") == "roughplastic_const":
    M0.tex[0].type = 2
print(json.dumps({"seed": seed, "material": mi, "bsdf_type": M0.bsdf_type, "map_kind": M0.map_kind, "tex_types": [M0.tex[i].type for i in range(4)], "f": [round(x, 5) for x in M0.f], "u": list(M0.u), "two_sided": M0.two_sided,
                  "tex0": [list(M0.tex[0].value), list(M0.tex[0].value1), list(M0.tex[0].uv_scale)], "map_tex": [M0.map_tex.type, M0.map_tex.image, list(M0.map_tex.value), list(M0.map_tex.uv_scale)]}))


def build(mat):
    from cudatracerlib_amd import rough_tables
    sc = api.DynamicScene()
    for slot in (0, 1):
        tr, df, er, ar = rough_tables.make_table(slot, n_eta=4, n_alpha=4, n_theta=8, quad=12)
        sc.setRoughTransmittance(slot, tr, df, er, ar)
    sc.add_image(scenes.checker_image(), api.TEXEL_RGBCOL, api.WRAP_REPEAT, api.FILTER_BILINEAR if seed % 2 else api.FILTER_POINT)
    sc.add_image(api.float3_to_rgbcol(scenes.bump_image(32, seed=seed)), api.TEXEL_RGBCOL, api.WRAP_REPEAT, api.FILTER_BILINEAR)
    sc.add_image(api.float3_to_rgbcol(scenes.normal_image(32)), api.TEXEL_RGBCOL, api.WRAP_REPEAT, api.FILTER_BILINEAR)
    m = api.ctl_material.from_buffer_copy(C.string_at(C.addressof(mat), C.sizeof(api.ctl_material)))
    if m.bsdf_type in (13, 14, 15):      # nested models: re-register the nested records
        for slot in ((2, 3) if m.bsdf_type == 15 else (2,)):
            n = api.ctl_material.from_buffer_copy(C.string_at(C.addressof(sd.materials[m.u[slot]]), C.sizeof(api.ctl_material)))
            m.u[slot] = sc.add_material(n)
    P, I, N = scenes._quad([[-6, 0, -6], [-6, 0, 6], [6, 0, 6], [6, 0, -6]], [0, 1, 0])
    sc.CreateNode(sc.add_mesh(P, I, normals=N, uvs=np.array([[0, 0], [0, 4], [4, 4], [4, 0]], np.float32), materials=[m]))
    V, F = scenes.icosphere(2)
    uv_s = np.stack([np.arctan2(V[:, 2], V[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(V[:, 1], -1, 1)) / np.pi], axis=1).astype(np.float32)
    xf = np.eye(4, dtype=np.float32); xf[:3, :3] *= 1.5; xf[:3, 3] = [0, 1.6, 0]
    mode = os.environ.get("XF", "none")
    A = np.diag([1.5, 1.1, 0.8]) if mode != "none" else np.eye(3) * 1.5
    if "rot" in mode:
        a = 0.7; A = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ A
    if "mirror" in mode: A = A @ np.diag([1.0, 1.0, -1.0])
    if "shear" in mode: A = A @ np.array([[1, 0.4, 0], [0, 1, 0], [0, 0.3, 1.0]])
    xf[:3, :3] = A.astype(np.float32)
    if os.environ.get("SHAPE") == "box":
        Pb, Ib, Nb = scenes.unit_box(); uv_b = (Pb[:, [0, 2]] * 0.5 + 0.5).astype(np.float32)
        sc.CreateNode(sc.add_mesh(Pb, Ib, normals=Nb, uvs=uv_b, materials=[m]), xf)
    else:
        sc.CreateNode(sc.add_mesh(V, F, normals=V, uvs=uv_s, materials=[m]), xf)
    P, I, N = scenes._quad([[-2, 7.9, -2], [2, 7.9, -2], [2, 7.9, 2], [-2, 7.9, 2]], [0, -1, 0])
    sc.CreateLight(sc.CreateNode(sc.add_mesh(P, I, normals=N, materials=[api.diffuse((0.5, 0.5, 0.5))])), 0, (20.0, 18.0, 15.0))
    sc.CreatePointLight((4, 5, 3), (60, 60, 80))
    sc.setCamera((0, 4.5, 9), (0, 1.2, 0), (0, 1, 0), 50.0, W, H)
    sc.UpdateScene()
    return sc


def compare(tag, mat, depth):
    sc = build(mat); d = sc.desc
    tables = orc.sequence_tables(PASSES)
    want, _ = orc.render(d, W, H, n_passes=PASSES, tables=tables, max_path_length=depth, rr_start=5)
    tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", depth); p.setValue("RRStartDepth", 5)
    tr.Resize(W, H); tr.InitializeScene(gpu.Scene(d, flatten=True)); img = gpu.Image(W, H)
    for k in range(PASSES):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData()
    g, w = got[..., :3], want[..., :3]
    off = ~(np.abs(g - w) <= 2e-3 * (1 + np.abs(w))).all(axis=2)
    ys, xs = np.nonzero(off)
    print(json.dumps({"variant": tag, "depth": depth, "off_pixels": int(off.sum()), "exact": round(float((g == w).all(axis=2).mean()), 4), "weights_equal": bool(np.array_equal(got[..., 6], want[..., 6])),
                      "max_rel": float((np.abs(g - w) / (1 + np.abs(w))).max()), "where": [[int(x), int(y)] for x, y in zip(xs[:3], ys[:3])], "gpu": g[off][:2].round(5).tolist(), "cpu": w[off][:2].round(5).tolist()}), flush=True)


def variant(**kw):
    m = api.ctl_material.from_buffer_copy(C.string_at(C.addressof(M0), C.sizeof(api.ctl_material)))
    for k, v in kw.items():
        if k == "const_tex":
            for i in v:
                m.tex[i].type = 2
        elif k == "u":
            for i, x in v.items(): m.u[i] = x
        else:
            setattr(m, k, v)
    return m


for depth in (1, 2):
    compare("as is " + os.environ.get("XF", "none"), M0, depth)
if os.environ.get("XF"): sys.exit(0)
compare("map off", variant(map_kind=0), 2)
compare("textures constant", variant(const_tex=[0, 1, 2, 3]), 2)
compare("map off + textures constant", variant(map_kind=0, const_tex=[0, 1, 2, 3]), 2)
compare("u[0] = 0", variant(u={0: 0}), 2)
compare("u[1] = 0", variant(u={1: 0}), 2)
compare("two_sided", variant(two_sided=1), 2)
