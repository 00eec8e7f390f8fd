#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE'S OWN code (oracle/_ref/libctlref.so, compiled from /root/reference
by `make -C oracle ref`).  Run in the build container only; the fixtures are data (inputs + the reference's outputs),
committed so that the oracle stays pinned on boxes where /root/reference does not exist.

    python tests/golden/generate.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

f32 = C.c_float


def main():
    r = oracle.load_ref()
    if r is None:
        raise SystemExit("oracle/_ref/libctlref.so missing: run `make -C oracle ref` (needs /root/reference)")
    rs = np.random.RandomState(20260929)

    # ---- Woop triangles: setData rows, getData round trip, Intersect (Engine/TriIntersectorData.cu)
    n = 256
    tris = (rs.normal(size=(n, 3, 3)) * rs.uniform(0.05, 50, size=(n, 1, 1))).astype(np.float32)
    rows = np.zeros((n, 12), np.float32); back = np.zeros((n, 3, 3), np.float32)
    rays_o = np.zeros((n, 3), np.float32); rays_d = np.zeros((n, 3), np.float32); hits = np.zeros((n, 4), np.float32)
    for i in range(n):
        r.ref_woop_set_data(tris[i, 0].ctypes.data_as(C.c_void_p), tris[i, 1].ctypes.data_as(C.c_void_p), tris[i, 2].ctypes.data_as(C.c_void_p), rows[i].ctypes.data_as(C.c_void_p))
        r.ref_woop_get_data(rows[i].ctypes.data_as(C.c_void_p), back[i, 0].ctypes.data_as(C.c_void_p), back[i, 1].ctypes.data_as(C.c_void_p), back[i, 2].ctypes.data_as(C.c_void_p))
        b = rs.dirichlet([1, 1, 1]) if i % 3 else rs.uniform(-0.5, 1.5, size=3)
        target = (b[0] * tris[i, 0] + b[1] * tris[i, 1] + (1 - b[0] - b[1]) * tris[i, 2]).astype(np.float32)
        o = (target + rs.normal(size=3) * 5).astype(np.float32)
        d = target - o; d = (d / np.linalg.norm(d)).astype(np.float32)
        rays_o[i], rays_d[i] = o, d
        tuv = np.zeros(3, np.float32)
        h = r.ref_woop_intersect(rows[i].ctypes.data, o.ctypes.data, d.ctypes.data, f32(1e30), tuv.ctypes.data)
        hits[i] = [h, tuv[0], tuv[1], tuv[2]]
    np.savez_compressed(os.path.join(HERE, "woop.npz"), tris=tris, rows=rows, back=back, rays_o=rays_o, rays_d=rays_d, hits=hits)

    # ---- codecs: half (host branch), spherical normals (Math/half.h, Math/Compression.h)
    h2f = np.array([r.ref_half_to_float(h) for h in range(65536)], np.float32)
    xs = np.concatenate([rs.normal(size=4096).astype(np.float32) * np.float32(10.0) ** rs.randint(-8, 6, 4096).astype(np.float32),
                         np.array([0, -0.0, 1e-8, 6e-5, 65504, 65520, 1e6, np.inf, -np.inf], np.float32)]).astype(np.float32)
    f2h = np.array([r.ref_float_to_half(float(x)) for x in xs], np.uint16)
    dec = np.zeros((65536, 3), np.float32)
    for v in range(65536):
        r.ref_normal_decode(v, dec[v].ctypes.data)
    nrm = rs.normal(size=(4096, 3)).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    enc = np.array([r.ref_normal_encode(nrm[i].ctypes.data_as(C.c_void_p)) for i in range(len(nrm))], np.uint16)
    np.savez_compressed(os.path.join(HERE, "codecs.npz"), half_to_float_host=h2f, f2h_in=xs, f2h_out=f2h, normal_decode=dec, normal_in=nrm, normal_encode=enc)

    # ---- TriangleData pack + fillDG (Engine/TriangleData.cu)
    n = 128
    P = (rs.normal(size=(n, 9)) * 3).astype(np.float32)
    N = rs.normal(size=(n, 3, 3)).astype(np.float32); N /= np.linalg.norm(N, axis=2, keepdims=True); N = N.reshape(n, 9).copy()
    T = rs.uniform(-2, 2, size=(n, 6)).astype(np.float32); T[: n // 4] = 0
    packed = np.zeros((n, 8), np.uint32); dg = np.zeros((n, 21), np.float32)
    M = np.zeros((n, 16), np.float32); uv = rs.dirichlet([1, 1, 1], size=n).astype(np.float32)
    for i in range(n):
        r.ref_triangle_data_pack(P[i].ctypes.data, N[i].ctypes.data, T[i].ctypes.data, i % 7, packed[i].ctypes.data)
        m = np.eye(4, dtype=np.float32)
        if i % 2:
            q = rs.normal(size=(3, 3)); q, _ = np.linalg.qr(q); m[:3, :3] = q * rs.uniform(0.5, 3); m[:3, 3] = rs.normal(size=3) * 10
        M[i] = m.reshape(16)
        r.ref_triangle_fill_dg(packed[i].ctypes.data, M[i].ctypes.data, f32(uv[i, 0]), f32(uv[i, 1]), dg[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "triangle_data.npz"), P=P, N=N, T=T, packed=packed, M=M, uv=uv[:, :2].copy(), dg=dg)

    # ---- warps / Fresnel / frames (Math/Warp.h, FresnelHelper.h, Frame.h, float4x4.h)
    s = rs.uniform(0, 1, size=(1024, 2)).astype(np.float32); s[0] = [0.5, 0.5]; s[1] = [0, 0]; s[2] = [0.999999, 0.5]
    cosh = np.zeros((len(s), 3), np.float32); tri = np.zeros((len(s), 2), np.float32); disk = np.zeros((len(s), 2), np.float32)
    for i in range(len(s)):
        r.ref_square_to_cosine_hemisphere(f32(s[i, 0]), f32(s[i, 1]), cosh[i].ctypes.data)
        r.ref_square_to_uniform_triangle(f32(s[i, 0]), f32(s[i, 1]), tri[i].ctypes.data)
        r.ref_square_to_uniform_disk_concentric(f32(s[i, 0]), f32(s[i, 1]), disk[i].ctypes.data)
    ci = rs.uniform(-1, 1, size=512).astype(np.float32); eta = rs.choice(np.array([1.0, 1.5, 1.5046 / 1.000277, 1 / 1.5, 2.4], np.float32), size=512)
    fd = np.zeros((512, 2), np.float32)
    for i in range(512):
        ct = f32()
        fd[i, 0] = r.ref_fresnel_dielectric_ext(f32(ci[i]), f32(eta[i]), C.byref(ct)); fd[i, 1] = ct.value
    ce = rs.uniform(0, 1, size=256).astype(np.float32); ek = rs.uniform(0.05, 5, size=(256, 6)).astype(np.float32); fc = np.zeros((256, 3), np.float32)
    for i in range(256):
        r.ref_fresnel_conductor_exact(f32(ce[i]), ek[i, :3].ctypes.data, ek[i, 3:].ctypes.data, fc[i].ctypes.data)
    cs_in = nrm[:512].copy(); cs_s = np.zeros((512, 3), np.float32); cs_t = np.zeros((512, 3), np.float32)
    for i in range(512):
        r.ref_coordinate_system(cs_in[i].ctypes.data_as(C.c_void_p), cs_s[i].ctypes.data_as(C.c_void_p), cs_t[i].ctypes.data_as(C.c_void_p))
    mats = rs.normal(size=(128, 16)).astype(np.float32); mats[::2, 12:] = [0, 0, 0, 1]; inv = np.zeros((128, 16), np.float32)
    for i in range(128):
        r.ref_matrix_inverse(mats[i].ctypes.data_as(C.c_void_p), inv[i].ctypes.data_as(C.c_void_p))
    # FresnelHelper::fresnelDiffuseReflectance(eta, fast=false) (Math/FresnelHelper.cu:13-60), used by plastic::Update()
    r.ref_fresnel_diffuse_reflectance.restype = C.c_float; r.ref_fresnel_diffuse_reflectance.argtypes = [C.c_float, C.c_int]
    fdr_eta = np.array([1.49 / 1.000277, 1.000277 / 1.49, 1.5, 1 / 1.5, 1.33, 1 / 1.33, 1.9, 1 / 1.9, 2.4, 1.05], np.float32)
    fdr_val = np.array([r.ref_fresnel_diffuse_reflectance(float(e), 0) for e in fdr_eta], np.float32)
    np.savez_compressed(os.path.join(HERE, "math.npz"), s=s, cosine_hemisphere=cosh, uniform_triangle=tri, disk_concentric=disk, fdr_eta=fdr_eta, fdr_value=fdr_val,
                        fd_cos=ci, fd_eta=eta, fd_out=fd, fc_cos=ce, fc_eta_k=ek, fc_out=fc, cs_in=cs_in, cs_s=cs_s, cs_t=cs_t, mat_in=mats, mat_inv=inv)

    # ---- microfacet distribution (Engine/MicrofacetDistribution.cu): Beckmann/GGX eval, G1, pdf, sample
    n = 512
    wi = rs.normal(size=(n, 3)).astype(np.float32); wi[:, 2] = np.abs(wi[:, 2]) + 0.05; wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    m = rs.normal(size=(n, 3)).astype(np.float32); m[:, 2] = np.abs(m[:, 2]) + 0.2; m /= np.linalg.norm(m, axis=1, keepdims=True)
    cfg = np.stack([rs.randint(0, 3, n), rs.randint(0, 2, n)], axis=1).astype(np.int32)   # type (Beckmann, GGX, Phong), sampleVisible
    alpha = rs.uniform(0.02, 0.6, size=(n, 2)).astype(np.float32); alpha[::2, 1] = alpha[::2, 0]
    cfg[(cfg[:, 0] == 2), 1] = 0   # MicrofacetDistribution::getSampleVisible: Phong never samples visible normals (MicrofacetDistribution.h:45-48)
    ev = np.zeros((n, 3), np.float32); sm = np.zeros((n, 4), np.float32); su = rs.uniform(0.01, 0.99, size=(n, 2)).astype(np.float32)
    for i in range(n):
        r.ref_microfacet_eval(int(cfg[i, 0]), f32(alpha[i, 0]), f32(alpha[i, 1]), int(cfg[i, 1]), wi[i].ctypes.data, m[i].ctypes.data, ev[i].ctypes.data)
        r.ref_microfacet_sample(int(cfg[i, 0]), f32(alpha[i, 0]), f32(alpha[i, 1]), int(cfg[i, 1]), wi[i].ctypes.data, f32(su[i, 0]), f32(su[i, 1]), sm[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "microfacet.npz"), wi=wi, m=m, cfg=cfg, alpha=alpha, eval=ev, su=su, sample=sm)

    # ---- PerspectiveSensor::sampleRay (SceneTypes/Sensor.cu:76-128)
    n = 256
    cams = []
    px = rs.uniform(0, 1, size=(n, 2)).astype(np.float32); out = np.zeros((n, 6), np.float32); par = np.zeros((n, 5), np.float32); tw = np.zeros((n, 16), np.float32)
    for i in range(n):
        w, h = int(rs.choice([64, 256, 1920])), int(rs.choice([64, 256, 1080]))
        fov = np.float32(np.radians(rs.uniform(20, 100)))
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        m = np.eye(4, dtype=np.float32); m[:3, :3] = q; m[:3, 3] = rs.normal(size=3) * 20
        tw[i] = m.reshape(16); par[i] = [fov, 1e-2, 1e4, w, h]
        p = (px[i] * [w, h]).astype(np.float32); px[i] = p
        r.ref_sensor_sample_ray(tw[i].ctypes.data, f32(fov), f32(1e-2), f32(1e4), w, h, f32(p[0]), f32(p[1]), out[i, :3].ctypes.data, out[i, 3:].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "sensor.npz"), to_world=tw, params=par, pixel=px, ray=out)
    # ---- ConstructBVH = SplitBVHBuilder with spatial splits (Engine/MeshLoader/BVHBuilderHelper.cpp:116-127): the reference's node,
    #      Woop and index arrays for a few small meshes (inputs are regenerated by tests/test_sbvh.py::sbvh_cases from the same seeds)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from test_sbvh import sbvh_cases
    out = {}
    for name, (V, F) in sbvh_cases().items():
        V = np.ascontiguousarray(V, np.float32); F = np.ascontiguousarray(F, np.uint32)
        nn, nt = C.c_uint32(), C.c_uint32()
        r.ref_construct_bvh(V.ctypes.data, F.ctypes.data, len(V), F.size, C.byref(nn), C.byref(nt))
        nodes = np.zeros((nn.value, 16), np.uint32); tris = np.zeros((nt.value, 12), np.uint32); idx = np.zeros(nt.value, np.uint32)
        r.ref_construct_bvh_fetch(nodes.ctypes.data, tris.ctypes.data, idx.ctypes.data)
        n_entries = nt.value - 2                                  # the callback allocates two spare entries (BVHBuilderHelper.cpp:34-39)
        n_nodes = max(1, nn.value - 2) if n_entries else 0        # a one-leaf mesh uses one of the spare nodes for its (leaf, none) root
        out[name + "_V"] = V; out[name + "_F"] = F
        out[name + "_nodes"] = nodes[:n_nodes]; out[name + "_woop"] = tris[:n_entries]; out[name + "_index"] = idx[:n_entries]
    np.savez_compressed(os.path.join(HERE, "sbvh.npz"), **out)
    print("golden fixtures written to", HERE)


    # ---- first-hit ray differentials (added in round 2; own random stream, so that the fixtures above stay byte-identical):
    #      PerspectiveSensor::sampleRayDifferential (SceneTypes/Sensor.cu:130-144) and DifferentialGeometry::computePartials (Engine/DifferentialGeometry.cu:9-90)
    rs2 = np.random.RandomState(20260930)
    n = 256
    px = np.zeros((n, 2), np.float32); par = np.zeros((n, 5), np.float32); tw = np.zeros((n, 16), np.float32); rays = np.zeros((n, 12), np.float32)
    dgs = np.zeros((n, 12), np.float32); parts = np.zeros((n, 4), np.float32)
    for i in range(n):
        w, h = int(rs2.choice([64, 256, 1920])), int(rs2.choice([64, 256, 1080]))
        fov = np.float32(np.radians(rs2.uniform(20, 100)))
        q, _ = np.linalg.qr(rs2.normal(size=(3, 3)))
        m = np.eye(4, dtype=np.float32); m[:3, :3] = q; m[:3, 3] = rs2.normal(size=3) * 20
        tw[i] = m.reshape(16); par[i] = [fov, 1e-2, 1e4, w, h]
        p = (rs2.uniform(0, 1, size=2) * [w, h]).astype(np.float32); px[i] = p
        r.ref_sensor_sample_ray_differential(tw[i].ctypes.data, f32(fov), f32(1e-2), f32(1e4), w, h, f32(p[0]), f32(p[1]),
                                             rays[i, 0:3].ctypes.data, rays[i, 3:6].ctypes.data, rays[i, 6:9].ctypes.data, rays[i, 9:12].ctypes.data)
        # a surface point in front of the camera with a random frame; every 16th case is degenerate (zero dpdu / dpdv, or a normal at right angles to a differential ray)
        t = rs2.uniform(1, 50)
        P = (rays[i, 0:3] + t * rays[i, 3:6]).astype(np.float32)
        nrm = rs2.normal(size=3); nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
        dpdu = (rs2.normal(size=3) * rs2.uniform(0.01, 10)).astype(np.float32); dpdv = (rs2.normal(size=3) * rs2.uniform(0.01, 10)).astype(np.float32)
        if i % 16 == 5: dpdu[:] = 0; dpdv[:] = 0
        if i % 16 == 9: dpdv = (dpdu * np.float32(2)).astype(np.float32)          # singular 2x2 system
        if i % 16 == 13: nrm = np.array([1, 0, 0], np.float32); rays[i, 6:9] = [0, 1, 0]   # n . rx.dir == 0
        dgs[i] = np.concatenate([P, nrm, dpdu, dpdv])
        r.ref_compute_partials(dgs[i, 0:3].ctypes.data, dgs[i, 3:6].ctypes.data, dgs[i, 6:9].ctypes.data, dgs[i, 9:12].ctypes.data,
                               rays[i, 0:3].ctypes.data, rays[i, 3:6].ctypes.data, rays[i, 6:9].ctypes.data, rays[i, 9:12].ctypes.data, parts[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "partials.npz"), to_world=tw, params=par, pixel=px, rays=rays, dg=dgs, partials=parts)

    # ---- the four projective sensors (added in round 2, own random stream): PerspectiveSensor, ThinLensSensor, OrthographicSensor, TelecentricSensor
    #      sampleRay + sampleRayDifferential (SceneTypes/Sensor.cu), and computePartials fed with the differential rays' own origins
    rs3 = np.random.RandomState(20260931)
    n = 384
    typ = np.zeros(n, np.int32); par = np.zeros((n, 8), np.float32); tw = np.zeros((n, 16), np.float32); smp = np.zeros((n, 4), np.float32)
    rays = np.zeros((n, 18), np.float32); ray_d = np.zeros((n, 6), np.float32); dgs = np.zeros((n, 12), np.float32); parts = np.zeros((n, 4), np.float32)
    for i in range(n):
        typ[i] = 2 + i % 4
        w, h = int(rs3.choice([64, 256, 1920])), int(rs3.choice([64, 256, 1080]))
        fov = np.float32(np.radians(rs3.uniform(20, 100)))
        q, _ = np.linalg.qr(rs3.normal(size=(3, 3)))
        m = np.eye(4, dtype=np.float32); m[:3, :3] = q; m[:3, 3] = rs3.normal(size=3) * 20
        tw[i] = m.reshape(16)
        aperture, focus, sscale = np.float32(rs3.uniform(0.0, 0.5)), np.float32(rs3.uniform(0.5, 30.0)), np.float32(rs3.choice([1.0, 2.0, 0.5]))
        nearD, farD = (np.float32(1e-2), np.float32(1e4)) if typ[i] in (2, 3) else (np.float32(1e-5), np.float32(1e5))
        par[i] = [fov, nearD, farD, w, h, aperture, focus, sscale]
        smp[i, :2] = (rs3.uniform(0, 1, size=2) * [w, h]).astype(np.float32); smp[i, 2:] = rs3.uniform(0, 1, size=2).astype(np.float32)
        r.ref_sensor_rays(int(typ[i]), tw[i].ctypes.data, f32(fov), f32(nearD), f32(farD), w, h, f32(aperture), f32(focus), f32(sscale),
                          f32(smp[i, 0]), f32(smp[i, 1]), f32(smp[i, 2]), f32(smp[i, 3]), rays[i].ctypes.data, ray_d[i].ctypes.data)
        t = rs3.uniform(1, 50)
        P = (ray_d[i, 0:3] + t * ray_d[i, 3:6]).astype(np.float32)
        nrm = rs3.normal(size=3); nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
        dpdu = (rs3.normal(size=3) * rs3.uniform(0.01, 10)).astype(np.float32); dpdv = (rs3.normal(size=3) * rs3.uniform(0.01, 10)).astype(np.float32)
        dgs[i] = np.concatenate([P, nrm, dpdu, dpdv])
        r.ref_compute_partials_origins(dgs[i, 0:3].ctypes.data, dgs[i, 3:6].ctypes.data, dgs[i, 6:9].ctypes.data, dgs[i, 9:12].ctypes.data,
                                       ray_d[i, 0:3].ctypes.data, ray_d[i, 3:6].ctypes.data, rays[i, 6:9].ctypes.data, rays[i, 9:12].ctypes.data,
                                       rays[i, 12:15].ctypes.data, rays[i, 15:18].ctypes.data, parts[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "sensors.npz"), type=typ, to_world=tw, params=par, samples=smp, rays=rays, ray_diff=ray_d, dg=dgs, partials=parts)

    # ---- reconstruction filters of the image pipeline (SceneTypes/Filter.h), own random stream
    rs4 = np.random.RandomState(20260932)
    n = 600
    cfg = np.zeros((n, 5), np.float32); xy = np.zeros((n, 2), np.float32); val = np.zeros(n, np.float32)
    for i in range(n):
        t = 1 + i % 5
        xw, yw = np.float32(rs4.choice([1.0, 2.0, 3.0, 6.0])), np.float32(rs4.choice([1.0, 2.0, 3.0, 6.0]))
        p0, p1 = {1: (0, 0), 2: (rs4.choice([2.0, 0.5, 1.0]), 0), 3: (rs4.choice([1 / 3.0, 0.0, 1.0]), rs4.choice([1 / 3.0, 0.5])), 4: (rs4.choice([3.0, 2.0]), 0), 5: (0, 0)}[t]
        cfg[i] = [t, xw, yw, p0, p1]
        xy[i] = (rs4.uniform(0, 1.1, size=2) * [xw, yw]).astype(np.float32)          # |dx|, |dy| as CanonicalFilter passes them, a little beyond the support
        if i % 50 < 5: xy[i, 0] = 0.0                                                # the sinc's small-argument branch
        val[i] = r.ref_filter_evaluate(int(t), f32(xw), f32(yw), f32(cfg[i, 3]), f32(cfg[i, 4]), f32(xy[i, 0]), f32(xy[i, 1]))
    np.savez_compressed(os.path.join(HERE, "filters.npz"), cfg=cfg, xy=xy, value=val)

    # ---- texel / frame codecs (Math/Spectrum.h:521-565), own random stream: RGBE and RGBCOL both ways
    rs5 = np.random.RandomState(20260933)
    n = 4096
    rgb = (rs5.uniform(0, 1, size=(n, 3)) * np.exp2(rs5.randint(-20, 21, size=(n, 1)))).astype(np.float32)
    rgb[:16] = 0; rgb[16:32] = [[1e-33, 0, 0]] * 16; rgb[32] = [1.0, 0.5, 0.25]; rgb[33] = [255.9999, 256.0, 1.0]; rgb[34] = [-1.0, 0.5, 2.0]; rgb[35] = [0.5, 0.5, 0.5]
    rgb[36:300] = rs5.uniform(-0.2, 1.2, size=(264, 3)).astype(np.float32)                     # the RGBCOL clamp range
    enc_e = np.zeros(n, np.uint32); enc_c = np.zeros(n, np.uint32); dec_e = np.zeros((n, 3), np.float32); dec_c = np.zeros((n, 3), np.float32)
    words = rs5.randint(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    for i in range(n):
        enc_e[i] = r.ref_float3_to_rgbe(f32(rgb[i, 0]), f32(rgb[i, 1]), f32(rgb[i, 2])) if rgb[i].min() >= 0 else 0
        enc_c[i] = r.ref_float3_to_rgbcol(f32(rgb[i, 0]), f32(rgb[i, 1]), f32(rgb[i, 2]))
        r.ref_rgbe_to_float3(int(words[i]), dec_e[i].ctypes.data); r.ref_rgbcol_to_float3(int(words[i]), dec_c[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "spectrum_codecs.npz"), rgb=rgb, rgbe=enc_e, rgbcol=enc_c, words=words, from_rgbe=dec_e, from_rgbcol=dec_c)

    # ---- texture addressing and the checkerboard (Engine/MIPMap_device.h:33-55, SceneTypes/Texture.h:10-41, :127-146), own random stream
    rs6 = np.random.RandomState(20260934)
    n = 2048
    uv = rs6.uniform(-3, 3, size=(n, 2)).astype(np.float32); uv[:64] = rs6.randint(-3, 4, size=(64, 2)).astype(np.float32)   # exact integers: frac == 0
    dim = rs6.choice([1.0, 7.0, 64.0, 300.0], size=(n, 2)).astype(np.float32); mode = (np.arange(n) % 4).astype(np.int32)
    loc = np.zeros((n, 2), np.float32); ok = np.zeros(n, np.int32)
    mp = np.stack([rs6.choice([1.0, 2.0, 0.5, 12.0], size=n), rs6.choice([1.0, 3.0, 0.25], size=n), rs6.uniform(-1, 1, size=n), rs6.uniform(-1, 1, size=n)], axis=1).astype(np.float32)
    sel = np.zeros(n, np.int32)
    for i in range(n):
        ok[i] = r.ref_wrap_coordinates(f32(uv[i, 0]), f32(uv[i, 1]), f32(dim[i, 0]), f32(dim[i, 1]), int(mode[i]), loc[i].ctypes.data)
        sel[i] = r.ref_checkerboard_select(f32(uv[i, 0]), f32(uv[i, 1]), f32(mp[i, 0]), f32(mp[i, 1]), f32(mp[i, 2]), f32(mp[i, 3]))
    np.savez_compressed(os.path.join(HERE, "texture_addressing.npz"), uv=uv, dim=dim, mode=mode, loc=loc, ok=ok, mapping=mp, checker=sel)

    # ---- SphericalSensor::sampleRay (SceneTypes/Sensor.cu:6-17), own random stream
    rs7 = np.random.RandomState(20260935)
    n = 256
    tw = np.zeros((n, 16), np.float32); res = np.zeros((n, 2), np.float32); px = np.zeros((n, 2), np.float32); rays = np.zeros((n, 18), np.float32); tmp6 = np.zeros(6, np.float32)
    for i in range(n):
        w, h = int(rs7.choice([64, 512, 2048])), int(rs7.choice([32, 256, 1024]))
        q, _ = np.linalg.qr(rs7.normal(size=(3, 3)))
        m = np.eye(4, dtype=np.float32); m[:3, :3] = q; m[:3, 3] = rs7.normal(size=3) * 20
        tw[i] = m.reshape(16); res[i] = [w, h]; px[i] = (rs7.uniform(0, 1, size=2) * [w, h]).astype(np.float32)
        r.ref_sensor_rays(1, tw[i].ctypes.data, f32(1.0), f32(1e-2), f32(1e4), w, h, f32(0), f32(0), f32(1), f32(px[i, 0]), f32(px[i, 1]), f32(0.5), f32(0.5), rays[i].ctypes.data, tmp6.ctypes.data)
    np.savez_compressed(os.path.join(HERE, "sensor_spherical.npz"), to_world=tw, resolution=res, pixel=px, ray=rays[:, :6])

def traceray_scenes():
    """the scenes of traceray.npz, built by the product's host code (deterministic): (key, DynamicScene, rays)"""
    from cudatracerlib_amd import scenes
    out = []
    for key, sc, seed in (("sm", scenes.synthetic_sm(32, 32, n_instances=60, subdiv=2), 31), ("cornell", scenes.cornell_box(32, 32, glass_sphere=True), 32)):
        d = sc.desc
        rs = np.random.RandomState(seed)
        lo, hi = np.array(d.box_min[:]), np.array(d.box_max[:])
        n = 3000
        rays = np.zeros((n, 8), np.float32)
        rays[:, :3] = rs.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), size=(n, 3))
        dd = rs.normal(size=(n, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        rays[:, 4:7] = dd; rays[:, 3] = np.float32(1e-4); rays[:, 7] = np.float32(3.402823466e+38)
        rays[:6, 4:7] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)   # axis-parallel: the 2^-80 guard of the slab test
        out.append((key, sc, rays))
    return out


def traceray_input_digest(d):
    """sha256 over the arrays the traversal reads: the fixture's outputs belong to exactly these inputs"""
    import hashlib
    h = hashlib.sha256()
    for name, dt, cnt, wd in (("scene_bvh_nodes", np.uint32, d.n_scene_bvh_nodes, 16), ("bvh_nodes", np.uint32, d.n_bvh_nodes, 16), ("woop", np.uint32, d.n_woop, 12),
                              ("woop_index", np.uint32, d.n_woop, 1), ("nodes", np.uint32, d.n_nodes, 6), ("meshes", np.uint32, d.n_meshes, 5), ("node_inv_transforms", np.uint32, d.n_nodes, 16)):
        h.update(np.ascontiguousarray(d.view(name, dt, cnt, wd)).tobytes())
    h.update(np.int32(d.scene_start_node).tobytes())
    return h.hexdigest()


def gen_traceray(r):
    # ---- two-level single-ray traversal through the reference's own TracerayTemplate (Engine/SpatialStructures/BVH/BVHTraversal.h:122-232), float4x4 transforms and
    #      TriIntersectorData::Intersect (oracle/ref_driver.cpp ref_trace_two_level) over scenes compiled by the product's host code; own random streams
    out = {}
    for key, sc, rays in traceray_scenes():
        want = oracle.ref_trace_two_level(r, sc.desc, rays)
        out[key + "_rays"] = rays; out[key + "_digest"] = np.array(traceray_input_digest(sc.desc))
        for f in ("dist", "u", "v", "tri_idx", "node_idx"):
            out[key + "_" + f] = np.ascontiguousarray(want[f])
    np.savez_compressed(os.path.join(HERE, "traceray.npz"), **out)


def gen_math2(r):
    # ---- second batch of small functions (Math/Warp.h intervalToTent / squareToTent, squareToCosineHemispherePdf, squareToUniformSphere; AlgebraHelper::Barycentric), own random stream
    rs = np.random.RandomState(20260940)
    n = 512
    s = np.concatenate([rs.uniform(0, 1, size=(n - 8, 2)), np.array([[0, 0], [0.5, 0.5], [1, 1], [0.25, 0.75], [0.49999997, 0.50000006], [1e-8, 1 - 1e-7], [0.999, 0.001], [0.5, 0]])]).astype(np.float32)
    r.ref_interval_to_tent.restype = f32; r.ref_interval_to_tent.argtypes = [f32]
    r.ref_cosine_hemisphere_pdf.restype = f32; r.ref_cosine_hemisphere_pdf.argtypes = [C.c_void_p]
    r.ref_square_to_uniform_sphere.argtypes = [f32, f32, C.c_void_p]
    r.ref_barycentric.argtypes = [C.c_void_p] * 5
    tent = np.array([r.ref_interval_to_tent(f32(x)) for x in s[:, 0]], np.float32)
    sph = np.zeros((n, 3), np.float32)
    for i in range(n):
        r.ref_square_to_uniform_sphere(f32(s[i, 0]), f32(s[i, 1]), sph[i].ctypes.data)
    dirs = rs.normal(size=(n, 3)); dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    cpdf = np.array([r.ref_cosine_hemisphere_pdf(dirs[i].ctypes.data) for i in range(n)], np.float32)
    # Barycentric: points in / near / outside random triangles, and on their edges
    tri = (rs.normal(size=(n, 3, 3)) * rs.uniform(0.1, 20, size=(n, 1, 1))).astype(np.float32)
    w = rs.dirichlet([1, 1, 1], size=n); w[n // 2:] = rs.uniform(-0.3, 1.3, size=(n - n // 2, 3)); w[:16, 2] = 0; w[:16, 1] = 1 - w[:16, 0]
    pts = (w[:, :1] * tri[:, 0] + w[:, 1:2] * tri[:, 1] + w[:, 2:3] * tri[:, 2]).astype(np.float32)
    uv = np.zeros((n, 2), np.float32); inside = np.zeros(n, np.int32)
    for i in range(n):
        inside[i] = r.ref_barycentric(pts[i].ctypes.data, tri[i, 0].ctypes.data, tri[i, 1].ctypes.data, tri[i, 2].ctypes.data, uv[i].ctypes.data)
    # FresnelHelper::reflect / refract about a normal (drawn after everything above, so that the earlier arrays stay as they were)
    r.ref_reflect_about.argtypes = [C.c_void_p] * 3; r.ref_refract_about.argtypes = [C.c_void_p, C.c_void_p, f32, f32, C.c_void_p]
    wi = rs.normal(size=(n, 3)); wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    nn = rs.normal(size=(n, 3)); nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    eta = rs.uniform(1.01, 2.5, size=n).astype(np.float32); ct = rs.uniform(-1, 1, size=n).astype(np.float32)
    refl = np.zeros((n, 3), np.float32); refr = np.zeros((n, 3), np.float32)
    for i in range(n):
        r.ref_reflect_about(wi[i].ctypes.data, nn[i].ctypes.data, refl[i].ctypes.data)
        r.ref_refract_about(wi[i].ctypes.data, nn[i].ctypes.data, f32(eta[i]), f32(ct[i]), refr[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "math2.npz"), s=s, tent=tent, sphere=sph, dirs=dirs, cosine_pdf=cpdf, tri=tri, pts=pts, bary_uv=uv, bary_inside=inside,
                        rr_wi=wi, rr_n=nn, rr_eta=eta, rr_cos_t=ct, reflect=refl, refract=refr)


def mipmap_cases():
    """the images and queries of the KernelMIPMap fixture (shared with tests/test_oracle_golden.py: the test rebuilds the inputs from the stored arrays, not from here)"""
    rs = np.random.RandomState(20261003)
    imgs = []
    for (w, h, typ) in ((32, 16, 1), (64, 64, 0), (20, 12, 1), (8, 128, 0)):
        if typ == 1:
            tex = rs.randint(0, 256, size=(h, w, 4)).astype(np.uint32)
        else:   # RGBE: mantissas + exponents around 128 (values 2^-6 .. 2^5), some zero texels
            tex = rs.randint(0, 256, size=(h, w, 4)).astype(np.uint32); tex[..., 3] = rs.randint(122, 134, size=(h, w)); tex[rs.uniform(size=(h, w)) < 0.05] = 0
        imgs.append((w, h, typ, (tex[..., 0] | (tex[..., 1] << 8) | (tex[..., 2] << 16) | (tex[..., 3] << 24)).astype(np.uint32)))
    n = 96
    q = {}
    uv = rs.uniform(-1.6, 2.6, size=(8, n, 2)).astype(np.float32); uv[:, :8] = np.array([[0, 0], [1, 1], [0.5, 0.5], [1, 0], [0, 1], [-1, -1], [2, 2], [0.999999, 1e-7]], np.float32)
    for what in range(8):
        a = np.zeros((n, 8), np.float32); a[:, :2] = uv[what]
        a[:, 6] = rs.randint(0, 9, size=n)                                                                  # level (beyond the last one too)
        if what == 2:      # ellipse coefficients as eval() forms them from texel-space derivatives
            d = rs.normal(size=(n, 4)) * 10.0 ** rs.uniform(-1.5, 1.2, size=(n, 1))
            A = d[:, 1] ** 2 + d[:, 3] ** 2; B = -2 * (d[:, 0] * d[:, 1] + d[:, 2] * d[:, 3]); Cc = d[:, 0] ** 2 + d[:, 2] ** 2; F = A * Cc - B * B * 0.25
            ok = F > 1e-6; A[~ok] = 1; B[~ok] = 0; Cc[~ok] = 1; F[~ok] = 1
            a[:, 2] = A / F; a[:, 3] = B / F; a[:, 4] = Cc / F
        elif what == 3:    # uv derivatives over four decades, some axis-aligned, some degenerate
            d = (rs.normal(size=(n, 4)) * 10.0 ** rs.uniform(-4.0, -0.3, size=(n, 1))).astype(np.float32)
            d[8:16, 1] = 0; d[8:16, 2] = 0; d[16:20] = 0; d[20:24, 2:] = d[20:24, :2] * 1e-4
            a[:, 2:6] = d
        elif what in (5, 7):
            a[:, 2] = 10.0 ** rs.uniform(-4, 0.3, size=n); a[:8, 2] = [0, 1e-9, 1, 2, 0.5, 0.25, 1e-3, 0.1]
            if what == 7:
                a[:, 3] = rs.randint(-3, 70, size=n); a[:, 4] = rs.randint(-3, 140, size=n)
        q[what] = a
    return imgs, q


def gen_mipmap(r):
    # ---- KernelMIPMap (Engine/MIPMap.cu:13-278, compiled from the reference by `make -C oracle ref`): Texel / triangle / evalEWA / eval / Sample / SampleAlpha on four
    # images x four wrap modes x the filter modes; the pyramid levels and the EWA weight table are inputs (their builder, MIPMap.cpp, needs FreeImage and is restated only)
    from cudatracerlib_amd import api
    orc = oracle.Oracle()
    lib = orc.lib
    lib.orc_mip_pyramid.restype = C.c_uint32; lib.orc_mip_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.orc_mip_query.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    r.ref_mipmap_query.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    imgs, q = mipmap_cases()
    out = {}
    for ii, (w, h, typ, tex) in enumerate(imgs):
        tex = np.ascontiguousarray(tex)
        M = api.ctl_mipmap(tex.ctypes.data, w, h, typ, 0, 1)
        levels = C.c_uint32(); offs = (C.c_uint32 * 16)()
        total = lib.orc_mip_pyramid(C.byref(M), C.byref(levels), offs, None, 0)
        pyr = np.zeros(total, np.uint32)
        lib.orc_mip_pyramid(C.byref(M), C.byref(levels), offs, pyr.ctypes.data_as(C.c_void_p), total)
        lut = np.zeros(64, np.float32); dummy = np.zeros(8, np.float32); o3 = np.zeros(3, np.float32)
        lib.orc_mip_query(C.byref(M), 0, 1, dummy.ctypes.data_as(C.c_void_p), o3.ctypes.data_as(C.c_void_p), lut.ctypes.data_as(C.c_void_p))
        out["img%d_texels" % ii] = tex; out["img%d_pyramid" % ii] = pyr; out["img%d_offsets" % ii] = np.array(list(offs), np.uint32)
        out["img%d_hdr" % ii] = np.array([w, h, typ, levels.value], np.uint32); out["lut"] = lut
        for wrap in range(4):
            for what in range(8):
                filters = (2, 3, 0, 1) if what == 3 else ((0, 1) if what == 4 else (1,))
                for filt in filters:
                    hdr = np.array([w, h, typ, wrap, filt, levels.value], np.uint32)
                    a = q[what].copy(); res = np.zeros((len(a), 3), np.float32)
                    if what == 0:
                        a[:, 6] = np.minimum(a[:, 6], levels.value - 1)      # Texel() is only ever called with a level that exists (triangle / evalEWA clamp before)
                    r.ref_mipmap_query(pyr.ctypes.data_as(C.c_void_p), hdr.ctypes.data_as(C.c_void_p), offs, lut.ctypes.data_as(C.c_void_p), what, len(a), a.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p))
                    out["img%d_wrap%d_what%d_filter%d" % (ii, wrap, what, filt)] = res
    for what in range(8):
        out["args%d" % what] = q[what]
    np.savez_compressed(os.path.join(HERE, "mipmap.npz"), **out)


def bsdf_cases():
    """materials of the BSDF fixture: name -> list of ctl_material (the LAST one is queried; earlier ones are what it nests)"""
    from cudatracerlib_amd import api
    chk = api.checker_texture((0.9, 0.2, 0.1), (0.1, 0.3, 0.8), uv_scale=(3.0, 2.0), uv_offset=(0.25, 0.5))
    rc = api.roughconductor(alpha=0.2, distribution=1, sample_visible=True)
    dchk = api.diffuse((0.5, 0.5, 0.5)); dchk.tex[0] = chk
    cases = {
        "diffuse": [api.diffuse((0.8, 0.5, 0.3))], "diffuse_checker": [dchk],
        "roughdiffuse": [api.roughdiffuse((1, 0.9, 0.8), alpha=0.5)], "roughdiffuse_fast": [api.roughdiffuse((0.8, 0.8, 0.8), alpha=0.3, use_fast_approx=True)],
        "dielectric": [api.dielectric(int_ior=1.5, ext_ior=1.0)], "dielectric_tinted": [api.dielectric(int_ior=1.33, ext_ior=1.0, specular_transmittance=(0.9, 0.95, 1.0), specular_reflectance=(1.0, 0.9, 0.8))],
        "thindielectric": [api.thindielectric(int_ior=1.5, ext_ior=1.0)],
        "roughdielectric_beck_vis": [api.roughdielectric(alpha=0.2, int_ior=1.5, ext_ior=1.0, distribution=0, sample_visible=True)],
        "roughdielectric_ggx_vis": [api.roughdielectric(alpha=0.15, int_ior=1.5, ext_ior=1.0, distribution=1, sample_visible=True)],
        "roughdielectric_beck": [api.roughdielectric(alpha=0.3, int_ior=1.33, ext_ior=1.0, distribution=0, sample_visible=False)],
        "roughdielectric_aniso": [api.roughdielectric(alpha=0.3, alpha_v=0.1, int_ior=1.5, ext_ior=1.0, distribution=1, sample_visible=True)],
        "roughdielectric_phong": [api.roughdielectric(alpha=0.25, int_ior=1.5, ext_ior=1.0, distribution=2, sample_visible=False)],
        "conductor": [api.conductor(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14))],
        "roughconductor_ggx_vis": [rc], "roughconductor_beck": [api.roughconductor(alpha=0.3, distribution=0, sample_visible=False)],
        "roughconductor_beck_vis_aniso": [api.roughconductor(alpha=0.25, alpha_v=0.1, distribution=0, sample_visible=True)],
        "roughconductor_phong": [api.roughconductor(alpha=0.2, alpha_v=0.35, distribution=2, sample_visible=False)],
        "roughconductor_ggx": [api.roughconductor(alpha=0.1, distribution=1, sample_visible=False)],
        "plastic": [api.plastic(diffuse_reflectance=(1, 1, 1), int_ior=1.49)], "plastic_nonlinear": [api.plastic(diffuse_reflectance=(0.5, 0.4, 0.3), int_ior=1.9, nonlinear=True)],
        "plastic_tinted": [api.plastic(diffuse_reflectance=(0.2, 0.6, 0.3), int_ior=1.33, ext_ior=1.0, specular_reflectance=(0.9, 0.8, 0.7))],
        "phong": [api.phong(diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(0.5, 0.5, 0.5), exponent=25.0)], "phong_sharp": [api.phong((0.1, 0.2, 0.3), (0.6, 0.5, 0.4), exponent=300.0)],
        "ward_balanced": [api.ward((0.4, 0.4, 0.4), (0.5, 0.5, 0.5), 0.15, 0.15, variant=2)], "ward_duer_aniso": [api.ward((0.4, 0.4, 0.4), (0.3, 0.3, 0.3), 0.1, 0.3, variant=1)],
        "ward": [api.ward((0.4, 0.3, 0.4), (0.3, 0.3, 0.2), 0.2, 0.3, variant=0)],
    }
    d = api.diffuse((0.6, 0.7, 0.8)); pl = api.plastic(diffuse_reflectance=(0.5, 0.4, 0.3), int_ior=1.5, ext_ior=1.0)
    cases["coating_diffuse"] = [d, api.coating(0, d, int_ior=1.5, ext_ior=1.0, thickness=1.0, sigma_a=(0.1, 0.4, 0.9))]
    cases["coating_roughconductor"] = [rc, api.coating(0, rc, int_ior=1.33, ext_ior=1.0, thickness=0.5, sigma_a=0.0, specular_reflectance=(0.9, 0.9, 1.0))]
    cases["coating_plastic"] = [pl, api.coating(0, pl, int_ior=1.6, ext_ior=1.0, thickness=2.0, sigma_a=(0.3, 0.2, 0.1))]
    cases["blend_diffuse_roughconductor"] = [d, rc, api.blend(0, d, 1, rc, weight=0.3)]
    cases["blend_plastic_dielectric"] = [pl, api.dielectric(int_ior=1.5, ext_ior=1.0), api.blend(0, pl, 1, api.dielectric(int_ior=1.5, ext_ior=1.0), weight=0.6)]
    return cases


def gen_bsdf(r):
    # ---- the reference's own BSDFs (SceneTypes/BSDF_Simple.cu without its unused curand include, BSDF_Complex.cu as it lies; oracle/ref_bsdf_driver.cpp):
    # sample (weight, pdf, wo, sampled lobe, eta), f + pdf with the solid-angle measure under three lobe masks, f + pdf with the discrete measure, and the
    # constants the reference's constructors derive (m_combinedType, fdrInt, fdrExt, invEta2 / invEta, specular sampling weight)
    from cudatracerlib_amd import api
    r.ref_bsdf_query.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_bsdf_derived.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    rs = np.random.RandomState(20261004)
    n = 192
    out = {}
    for name, mats in bsdf_cases().items():
        arr = (api.ctl_material * len(mats))(*mats); idx = len(mats) - 1
        wi = rs.normal(size=(n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
        if "dielectric" not in name:
            wi[:, 2] = np.abs(wi[:, 2])
        wi[:4] = [[0, 0, 1], [0.6, 0, 0.8], [0.999, 0, 0.0447101778], [-0.6, 0.64, 0.48]]
        q = np.zeros((n, 8), np.float32); q[:, :3] = wi; q[:, 3:5] = rs.rand(n, 2); q[:, 6:8] = rs.uniform(-1, 2, size=(n, 2))
        smp = np.zeros((n, 9), np.float32)
        assert r.ref_bsdf_query(C.addressof(arr), idx, 0, 0x1ff, n, q.ctypes.data, smp.ctypes.data) == 0, name
        # evaluation directions: the sampled ones of OTHER queries (so that both hemispheres and the specular directions occur), and the exact mirror / refraction directions
        q2 = q.copy(); wo = np.roll(smp[:, 4:7], 1, axis=0).copy()
        bad = ~(np.linalg.norm(wo, axis=1) > 0.5); alt = rs.normal(size=(n, 3)); alt /= np.linalg.norm(alt, axis=1, keepdims=True); wo[bad] = alt[bad]
        own = smp[:, 4:7].copy(); keep = (np.arange(n) % 3 == 0) & (np.linalg.norm(own, axis=1) > 0.5); wo[keep] = own[keep]      # every third query: its own sampled direction
        q2[:, 3:6] = wo
        out[name + "_materials"] = np.frombuffer(bytes(arr), np.uint8).copy(); out[name + "_sample_q"] = q; out[name + "_sample"] = smp; out[name + "_eval_q"] = q2
        for mask in (0x1ff, 0x2 | 0x4, 0x8 | 0x10, 0x20 | 0x40):
            for mode in (1, 2):
                ev = np.zeros((n, 9), np.float32)
                assert r.ref_bsdf_query(C.addressof(arr), idx, mode, mask, n, q2.ctypes.data, ev.ctypes.data) == 0
                out["%s_eval_mode%d_mask%x" % (name, mode, mask)] = ev[:, :4].copy()
        d5 = np.zeros(5, np.float32); assert r.ref_bsdf_derived(C.addressof(arr), idx, d5.ctypes.data) == 0
        out[name + "_derived"] = d5
    np.savez_compressed(os.path.join(HERE, "bsdf.npz"), **out)


def rough_cases():
    """materials of bsdf_rough.npz (the LAST one is queried) and the three transmittance tables (slot 0 / 1 / 2 = beckmann.dat / phong.dat / ggx.dat, RoughTransmittance.cu:126-128;
    looked up by the distribution TYPE 0 Beckmann / 1 GGX / 2 Phong, :140-158)"""
    from cudatracerlib_amd import api, rough_tables
    tables = [rough_tables.make_table(slot, n_eta=4, n_alpha=5, n_theta=8, quad=16) for slot in (0, 1, 2)]
    chk = api.checker_texture((0.05, 0.05, 0.05), (0.35, 0.35, 0.35), uv_scale=(3.0, 2.0), uv_offset=(0.25, 0.5))      # a roughness that varies over the surface: the 3-D lookup
    d = api.diffuse((0.6, 0.7, 0.8)); rc = api.roughconductor(alpha=0.2, distribution=1, sample_visible=True)
    cases = {
        "roughplastic_beck": [api.roughplastic((0.2, 0.5, 0.25), alpha=0.15, distribution=0)],
        "roughplastic_ggx_nonlinear": [api.roughplastic((0.6, 0.25, 0.2), alpha=0.3, int_ior=1.6, distribution=1, nonlinear=True)],
        "roughplastic_phong_tinted": [api.roughplastic((0.3, 0.3, 0.6), alpha=0.25, int_ior=1.33, ext_ior=1.0, distribution=2, specular_reflectance=(0.9, 0.8, 0.7))],
        "roughplastic_checker_alpha": [api.roughplastic((0.5, 0.5, 0.5), alpha=chk, distribution=1)],
        "roughcoating_diffuse": [d, api.roughcoating(0, d, alpha=0.2, int_ior=1.5, ext_ior=1.0, thickness=1.0, sigma_a=(0.1, 0.4, 0.9), distribution=0)],
        "roughcoating_roughconductor_ggx": [rc, api.roughcoating(0, rc, alpha=0.1, int_ior=1.33, ext_ior=1.0, thickness=0.5, sigma_a=0.0, distribution=1, specular_reflectance=(0.9, 0.9, 1.0))],
    }
    return cases, tables


def gen_bsdf_rough(r):
    """roughplastic (SceneTypes/BSDF_Simple.cu:890-1057) and roughcoating (BSDF_Complex.cu) of the reference build, through RoughTransmittanceManager -> RoughTransmittance::Evaluate /
    EvaluateDiffuse (Engine/RoughTransmittance.cu:55-121, 140-158) -> Math/Spline.cu, over three synthetic tables; plus the two lookups by themselves (cos theta < 0, eta < 1,
    eta below the table, alpha at and beyond the table's ends).  Same query scheme as gen_bsdf."""
    from cudatracerlib_amd import api
    r.ref_bsdf_query.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_bsdf_derived.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    r.ref_rough_manager_set.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, f32, f32, f32, f32]
    r.ref_rough_transmittance_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, f32, f32, f32, f32, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_rough_manager_eval.restype = C.c_float; r.ref_rough_manager_eval.argtypes = [C.c_int, f32, f32, f32]
    cases, tables = rough_cases()
    out = {}; keep = []
    for slot, (tr, df, er, ar) in enumerate(tables):
        tr = np.ascontiguousarray(tr, np.float32); df = np.ascontiguousarray(df, np.float32); keep += [tr, df]
        assert r.ref_rough_manager_set(slot, tr.ctypes.data, df.ctypes.data, tr.shape[0] // 2, tr.shape[1], tr.shape[2], f32(er[0]), f32(er[1]), f32(ar[0]), f32(ar[1])) == 0
        out["table%d_trans" % slot] = tr; out["table%d_diff" % slot] = df; out["table%d_ranges" % slot] = np.array([er[0], er[1], ar[0], ar[1]], np.float32)
    rs = np.random.RandomState(20261011)
    # the lookups alone: {cos theta, alpha, eta}
    nq = 256
    q = np.stack([rs.uniform(-1, 1, nq), rs.uniform(tables[0][3][0], tables[0][3][1], nq), rs.choice([1.49, 1 / 1.49, 1.6, 1.05, 1.0001, 2.4, 1 / 1.33], nq)], axis=1).astype(np.float32)
    q[0] = [1, tables[0][3][0], 1.5]; q[1] = [0, tables[0][3][1], 1.5]; q[2] = [0.5, tables[0][3][1] * 1.5, 1.5]; q[3] = [-0.3, 0.1, 1.5]; q[4] = [0.7, tables[0][3][0] * 0.5, 1.5]
    for slot, (tr, df, er, ar) in enumerate(tables):
        res = np.zeros((nq, 2), np.float32)
        assert r.ref_rough_transmittance_eval(out["table%d_trans" % slot].ctypes.data, out["table%d_diff" % slot].ctypes.data, tr.shape[0] // 2, tr.shape[1], tr.shape[2],
                                               f32(er[0]), f32(er[1]), f32(ar[0]), f32(ar[1]), nq, q.ctypes.data, res.ctypes.data) == 0
        out["lookup%d" % slot] = res
    out["lookup_q"] = q
    out["manager_by_type"] = np.array([[r.ref_rough_manager_eval(t, float(q[i, 0]), float(q[i, 1]), float(q[i, 2])) for i in range(16)] for t in range(3)], np.float32)   # type t reads slot t
    n = 192
    for name, mats in cases.items():
        arr = (api.ctl_material * len(mats))(*mats); idx = len(mats) - 1
        wi = rs.normal(size=(n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True); wi[:, 2] = np.abs(wi[:, 2])
        wi[:4] = [[0, 0, 1], [0.6, 0, 0.8], [0.999, 0, 0.0447101778], [-0.6, 0.64, 0.48]]; wi[4] = [0.6, 0, -0.8]
        q1 = np.zeros((n, 8), np.float32); q1[:, :3] = wi; q1[:, 3:5] = rs.rand(n, 2); q1[:, 6:8] = rs.uniform(-1, 2, size=(n, 2))
        smp = np.zeros((n, 9), np.float32)
        assert r.ref_bsdf_query(C.addressof(arr), idx, 0, 0x1ff, n, q1.ctypes.data, smp.ctypes.data) == 0, name
        q2 = q1.copy(); wo = np.roll(smp[:, 4:7], 1, axis=0).copy()
        bad = ~(np.linalg.norm(wo, axis=1) > 0.5); alt = rs.normal(size=(n, 3)); alt /= np.linalg.norm(alt, axis=1, keepdims=True); wo[bad] = alt[bad]
        own = smp[:, 4:7].copy(); kp = (np.arange(n) % 3 == 0) & (np.linalg.norm(own, axis=1) > 0.5); wo[kp] = own[kp]
        q2[:, 3:6] = wo
        out[name + "_materials"] = np.frombuffer(bytes(arr), np.uint8).copy(); out[name + "_sample_q"] = q1; out[name + "_sample"] = smp; out[name + "_eval_q"] = q2
        for mask in (0x1ff, 0x2 | 0x4, 0x8 | 0x10):
            for mode in (1, 2):
                ev = np.zeros((n, 9), np.float32)
                assert r.ref_bsdf_query(C.addressof(arr), idx, mode, mask, n, q2.ctypes.data, ev.ctypes.data) == 0
                out["%s_eval_mode%d_mask%x" % (name, mode, mask)] = ev[:, :4].copy()
        d5 = np.zeros(5, np.float32); assert r.ref_bsdf_derived(C.addressof(arr), idx, d5.ctypes.data) == 0
        out[name + "_derived"] = d5
    np.savez_compressed(os.path.join(HERE, "bsdf_rough.npz"), **out)


def scene_light_cases():
    """scenes whose area / environment emitters the reference build samples through the product's own scene description (name -> DynamicScene)"""
    from cudatracerlib_amd import scenes
    return {"cornell": scenes.cornell_box(64, 64), "panel_checker": scenes.area_lights_scene(kind="checker"), "panel_orthogonal": scenes.area_lights_scene(kind="orthogonal"),
            "panel_image": scenes.area_lights_scene(kind="image"), "panel_orthogonal_image": scenes.area_lights_scene(kind="orthogonal_image"),
            "env": scenes.env_scene(), "env_rotated": scenes.env_scene(rotate_env=True), "bathroom": scenes.synthetic_bathroom(64, 64, n_instances=12, subdiv=1),
            "sm": scenes.synthetic_sm(64, 64, n_instances=20, subdiv=1)}


def scene_light_digest(d):
    """what the reference's outputs depend on: the lights, the anim blob, TriangleData and the images' level-0 texels of the description"""
    import hashlib
    h = hashlib.sha256()
    h.update(C.string_at(d.lights, d.n_lights_buf * C.sizeof(type(d.lights.contents)))); h.update(C.string_at(d.anim, d.n_anim_bytes)); h.update(C.string_at(d.tri_data, d.n_tri_data * 32))
    for i in range(d.n_images):
        m = d.images[i]; h.update(C.string_at(m.texels, m.width * m.height * 4)); h.update(bytes([m.texel_type, m.wrap_mode, m.filter_mode]))
    return h.hexdigest()


def gen_scene_lights(r):
    """DiffuseLight::sampleDirect / pdfDirect through ShapeSet::SamplePosition (Engine/ShapeSet.cu:24-105) and InfiniteLight::sampleDirect / pdfDirect through
    internalSampleDirection / internalPdfDirection (SceneTypes/Light.cu:420-479) of the reference build, over scenes compiled by the product's host code (oracle/ref_scene_light_driver.cpp)"""
    r.ref_scene_light_sample_direct.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_scene_light_pdf_direct.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_scene_light_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_scene_image_texture_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rs = np.random.RandomState(20261019)
    out = {}
    for name, sc in scene_light_cases().items():
        d = sc.desc
        out[name + "_digest"] = np.frombuffer(scene_light_digest(d).encode(), np.uint8).copy()
        lo, hi = np.array(d.box_min[:], np.float32), np.array(d.box_max[:], np.float32)
        for li in range(d.n_lights_buf):
            if d.lights[li].type not in (2, 5):
                continue
            n = 256
            q = np.zeros((n, 8), np.float32)
            q[:, :3] = lo + (hi - lo) * rs.uniform(0.02, 0.98, size=(n, 3)); nrm = rs.normal(size=(n, 3)); q[:, 3:6] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
            q[:, 6:8] = rs.rand(n, 2); q[0, 6:8] = [0, 0]; q[1, 6:8] = [0.99999994, 0.99999994]; q[2, 6:8] = [0.5, 0.5]; q[3, 6:8] = [0, 0.99999994]
            res = np.zeros((n, 14), np.float32)
            assert r.ref_scene_light_sample_direct(C.addressof(d), li, n, q.ctypes.data, res.ctypes.data) == 0, (name, li)
            # pdfDirect for the sampled directions, and for directions that miss the emitter / random ones
            q2 = np.zeros((n, 14), np.float32); q2[:, :6] = q[:, :6]; q2[:, 6:9] = res[:, 4:7]; q2[:, 9] = res[:, 7]; q2[:, 10:13] = res[:, 11:14]
            bad = ~(np.linalg.norm(q2[:, 6:9], axis=1) > 0.5); alt = rs.normal(size=(n, 3)); alt /= np.linalg.norm(alt, axis=1, keepdims=True)
            q2[bad, 6:9] = alt[bad]; q2[bad, 9] = 1.0; q2[bad, 10:13] = -alt[bad]
            swap = np.arange(n) % 4 == 0; q2[swap, 6:9] = alt[swap]       # every fourth: an unrelated direction (environment: any direction has a density)
            pdf = np.zeros(n, np.float32)
            assert r.ref_scene_light_pdf_direct(C.addressof(d), li, n, q2.ctypes.data, pdf.ctypes.data) == 0
            out["%s_light%d_q" % (name, li)] = q; out["%s_light%d_sample" % (name, li)] = res; out["%s_light%d_pdf_q" % (name, li)] = q2; out["%s_light%d_pdf" % (name, li)] = pdf
            if d.lights[li].type == 2:      # DiffuseLight::eval at the sampled emitter points, seen from the reference points (and from behind: every fourth)
                live = res[:, 3] > 0
                q3 = np.zeros((int(live.sum()), 9), np.float32); q3[:, :3] = res[live, 8:11]; q3[:, 3:6] = res[live, 11:14]; q3[:, 6:9] = -res[live, 4:7]
                q3[::4, 6:9] *= -1
                ev = np.zeros((len(q3), 3), np.float32)
                if len(q3):
                    assert r.ref_scene_light_eval(C.addressof(d), li, len(q3), q3.ctypes.data, ev.ctypes.data) == 0
                out["%s_light%d_eval_q" % (name, li)] = q3; out["%s_light%d_eval" % (name, li)] = ev
        # ImageTexture::Evaluate(uv) / Average() for every image texture among the scene's materials and lights (SceneTypes/Texture.cu:6-13, 32-38)
        texs = []
        for mi in range(d.n_materials):
            texs += [(("mat%d_tex%d" % (mi, k)), d.materials[mi].tex[k]) for k in range(4) if d.materials[mi].tex[k].type == 4]
            if d.materials[mi].map_kind != 0 and d.materials[mi].map_tex.type == 4: texs.append(("mat%d_map" % mi, d.materials[mi].map_tex))
            if d.materials[mi].alpha_state != 0 and d.materials[mi].alpha_tex.type == 4: texs.append(("mat%d_alpha" % mi, d.materials[mi].alpha_tex))
        texs += [("light%d_rad" % li, d.lights[li].rad_texture) for li in range(d.n_lights_buf) if d.lights[li].type == 2 and d.lights[li].rad_texture.type == 4]
        for tn, t in texs:
            uvq = rs.uniform(-1.5, 2.5, size=(192, 2)).astype(np.float32); uvq[0] = [0, 0]; uvq[1] = [1, 1]; uvq[2] = [0.5, 0.5]; uvq[3] = [-0.25, 1.75]
            tv = np.zeros((193, 3), np.float32)
            assert r.ref_scene_image_texture_eval(C.addressof(d), C.byref(t), 192, uvq.ctypes.data, tv.ctypes.data) == 0, (name, tn)
            out["%s_%s_uv" % (name, tn)] = uvq; out["%s_%s_value" % (name, tn)] = tv
    np.savez_compressed(os.path.join(HERE, "scene_lights.npz"), **out)


def material_map_cases():
    """scenes whose materials carry normal / height / alpha maps (name -> DynamicScene), and hand-made materials without a scene (name -> ctl_material)"""
    from cudatracerlib_amd import scenes, api
    sc = {"maps_normal_luminance": scenes.maps_scene(32, 24, "normal", "luminance"), "maps_height_alpha": scenes.maps_scene(32, 24, "height", "alpha"),
          "maps_none_color": scenes.maps_scene(32, 24, None, "color"), "bathroom": scenes.synthetic_bathroom(64, 64, n_instances=12, subdiv=1)}
    hand = {"normal_constant": api.set_normal_map(api.diffuse(), (0.8, 0.4, 0.9)), "normal_flat": api.set_normal_map(api.diffuse(), (0.5, 0.5, 1.0)),
            "normal_checker": api.set_normal_map(api.diffuse(), api.checker_texture((0.6, 0.5, 0.9), (0.4, 0.55, 0.8), uv_scale=(3.0, 2.0))),
            "height_constant": api.set_height_map(api.diffuse(), 0.7), "plain": api.diffuse(),
            "alpha_checker_luminance": api.set_alpha_map(api.diffuse(), api.checker_texture(1.0, 0.0, uv_scale=(2.0, 5.0)), api.ALPHA_MAP_LUMINANCE, 0.5),
            "alpha_constant_color": api.set_alpha_map(api.diffuse(), (0.2, 0.4, 0.6), api.ALPHA_MAP_COLOR, 0.15, (0.3, 0.3, 0.6)),
            "alpha_checker_color": api.set_alpha_map(api.diffuse(), api.checker_texture((0.9, 0.1, 0.1), (0.1, 0.1, 0.9), uv_scale=(3.0, 3.0)), api.ALPHA_MAP_COLOR, 0.25, (1.0, 0.0, 0.0)),
            "alpha_reflectance_luminance_bright": api.set_alpha_map(api.diffuse((0.9, 0.9, 0.9)), 0.0, api.ALPHA_REFLECTANCE_LUMINANCE, 0.5),
            "alpha_reflectance_luminance_dark": api.set_alpha_map(api.diffuse((0.1, 0.1, 0.1)), 1.0, api.ALPHA_REFLECTANCE_LUMINANCE, 0.5),
            "alpha_reflectance_checker_color": api.set_alpha_map(api.diffuse(api.checker_texture((0.8, 0.2, 0.2), (0.2, 0.2, 0.8), uv_scale=(2.0, 2.0))), 0.0, api.ALPHA_REFLECTANCE_COLOR, 0.3, (0.8, 0.2, 0.2)),
            "alpha_mode_on_constant": api.set_alpha_map(api.diffuse(), 0.0, api.ALPHA_MAP_ALPHA, 0.5)}
    return sc, hand


def material_map_queries(rs, n):
    """n shading points for Material::SampleNormalMap: uv, an orthonormal shading frame, a geometric normal near it (every eighth: on the other side), dpdu / dpdv near the tangents"""
    q = np.zeros((n, 20), np.float32)
    q[:, :2] = rs.uniform(-0.5, 2.5, size=(n, 2)); q[0, :2] = [0, 0]; q[1, :2] = [1, 1]; q[2, :2] = [0.5, 0.5]
    for i in range(n):
        f, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        if np.linalg.det(f) < 0:
            f[:, 2] = -f[:, 2]
        s, t, nn = f[:, 0], f[:, 1], f[:, 2]
        if i < 4:
            s, t, nn = np.array([1.0, 0, 0]), np.array([0, 0, -1.0]), np.array([0, 1.0, 0])
        g = nn + 0.2 * rs.normal(size=3); g /= np.linalg.norm(g)
        if i % 8 == 7:
            g = -g
        q[i, 2:5], q[i, 5:8], q[i, 8:11], q[i, 11:14] = s, t, nn, g
        q[i, 14:17] = s * rs.uniform(0.2, 5) + 0.1 * rs.normal(size=3); q[i, 17:20] = t * rs.uniform(0.2, 5) + 0.1 * rs.normal(size=3)
    return q


def material_map_digest(d):
    """what the reference's outputs depend on: the materials and the images' level-0 texels of the description"""
    import hashlib
    h = hashlib.sha256()
    h.update(C.string_at(d.materials, d.n_materials * C.sizeof(type(d.materials.contents))))
    for i in range(d.n_images):
        m = d.images[i]; h.update(C.string_at(m.texels, m.width * m.height * 4)); h.update(bytes([m.texel_type, m.wrap_mode, m.filter_mode]))
    return h.hexdigest()


def gen_material_maps(r):
    """Material::SampleNormalMap / AlphaTest of the reference build (Engine/Material.cu, compiled whole; oracle/ref_material_driver.cpp) over the product's materials"""
    r.ref_material_sample_normal_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_material_alpha_test.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rs = np.random.RandomState(20261101)
    out = {}
    scs, hand = material_map_cases()

    def run(key, desc_addr, mat):
        if mat.map_kind != 0 or key.endswith("plain"):
            q = material_map_queries(rs, 256); res = np.zeros((256, 10), np.float32)
            assert r.ref_material_sample_normal_map(desc_addr, C.byref(mat), 256, q.ctypes.data, res.ctypes.data) == 0
            out[key + "_frame_q"] = q; out[key + "_frame"] = res
        if mat.alpha_state != 0 or key.endswith("plain"):
            q = rs.uniform(-0.5, 2.5, size=(1024, 4)).astype(np.float32); q[:, :2] = rs.dirichlet([1, 1, 1], size=1024)[:, :2]
            q[0, 2:] = [0, 0]; q[1, 2:] = [1, 1]; q[2, 2:] = [0.5, 0.5]; q[3, 2:] = [0.25, 0.75]
            res = np.zeros(1024, np.int32)
            assert r.ref_material_alpha_test(desc_addr, C.byref(mat), 1024, q.ctypes.data, res.ctypes.data) == 0
            out[key + "_alpha_q"] = q; out[key + "_alpha"] = res
    for name, sc in scs.items():
        d = sc.desc
        out[name + "_digest"] = np.frombuffer(material_map_digest(d).encode(), np.uint8).copy()
        for mi in range(d.n_materials):
            run("%s_mat%d" % (name, mi), C.addressof(d), d.materials[mi])
    for name, m in hand.items():
        out["hand_" + name + "_bytes"] = np.frombuffer(C.string_at(C.addressof(m), C.sizeof(m)), np.uint8).copy()
        run("hand_" + name, None, m)
    np.savez_compressed(os.path.join(HERE, "material_maps.npz"), **out)


def spline_cases():
    """inputs of spline.npz: tables of 2..32 knots per axis, query points inside, exactly on knots, at 0 and 1, just outside [0, 1] and NaN (own random stream)"""
    rs = np.random.RandomState(20260941)

    def points(n, sizes):
        p = rs.uniform(0, 1, size=(n, len(sizes))).astype(np.float32)
        for a, s in enumerate(sizes):
            k = rs.randint(0, s, size=n // 4).astype(np.float32) / np.float32(s - 1)       # exactly on knots (as fp32 allows)
            p[: n // 4, a] = k
        p[n // 4] = 0; p[n // 4 + 1] = 1; p[n // 4 + 2, 0] = np.nextafter(np.float32(1), np.float32(2)); p[n // 4 + 3, -1] = np.nextafter(np.float32(0), np.float32(-1))
        p[n // 4 + 4, 0] = np.nan; p[n // 4 + 5] = np.nextafter(np.float32(1), np.float32(0)); p[n // 4 + 6, 0] = -0.0
        return p
    cases = []
    for dim in (1, 2, 3):
        for _ in range(24 if dim < 3 else 16):
            sizes = [int(x) for x in rs.randint(2, 33, size=dim)]
            if _ == 0: sizes = [2] * dim
            if _ == 1: sizes = [3] * dim
            vals = (rs.uniform(0, 1, size=int(np.prod(sizes))) ** 2).astype(np.float32)      # transmittance-like: in [0, 1]
            if _ % 5 == 2: vals = rs.normal(size=vals.size).astype(np.float32) * np.float32(100)
            cases.append((dim, sizes, vals, points(48, sizes)))
    return cases


def gen_spline(r):
    """Spline::evalCubicInterp1D / 2D / 3D (Math/Spline.cu:6-44, 223-296, 376-453) of the reference build: min = 0, max = 1 per axis and extrapolate = false — the way
    RoughTransmittance calls it — plus, for the 1-D and 2-D functions, a shifted knot range with extrapolation on (pins the driver's argument order)"""
    r.ref_spline_eval_1d.restype = C.c_float; r.ref_spline_eval_1d.argtypes = [f32, C.c_void_p, C.c_uint32, f32, f32, C.c_int]
    r.ref_spline_eval_2d.restype = C.c_float; r.ref_spline_eval_2d.argtypes = [C.c_void_p] * 5 + [C.c_int]
    r.ref_spline_eval_3d.restype = C.c_float; r.ref_spline_eval_3d.argtypes = [C.c_void_p] * 5 + [C.c_int]
    out = {}
    for i, (dim, sizes, vals, pts) in enumerate(spline_cases()):
        sz = np.array(sizes, np.uint32); lo = np.zeros(dim, np.float32); hi = np.ones(dim, np.float32)
        res = np.zeros(len(pts), np.float32); res_x = np.zeros(len(pts), np.float32)
        lo2 = np.full(dim, -1.5, np.float32); hi2 = np.full(dim, 2.25, np.float32)
        for j, q in enumerate(pts):
            q = np.ascontiguousarray(q, np.float32); q2 = (np.nan_to_num(q, nan=0.5) * np.float32(5) - np.float32(2)).astype(np.float32)   # (a NaN with extrapolation on reads outside the table in the reference)
            if dim == 1:
                res[j] = r.ref_spline_eval_1d(f32(q[0]), vals.ctypes.data, int(sz[0]), f32(0), f32(1), 0)
                res_x[j] = r.ref_spline_eval_1d(f32(q2[0]), vals.ctypes.data, int(sz[0]), f32(lo2[0]), f32(hi2[0]), 1)
            else:
                fn = r.ref_spline_eval_2d if dim == 2 else r.ref_spline_eval_3d
                res[j] = fn(q.ctypes.data, vals.ctypes.data, sz.ctypes.data, lo.ctypes.data, hi.ctypes.data, 0)
                res_x[j] = fn(q2.ctypes.data, vals.ctypes.data, sz.ctypes.data, lo2.ctypes.data, hi2.ctypes.data, 1)
        out["c%02d_size" % i] = sz; out["c%02d_values" % i] = vals; out["c%02d_points" % i] = pts; out["c%02d_result" % i] = res; out["c%02d_result_shifted_extrapolated" % i] = res_x
    np.savez_compressed(os.path.join(HERE, "spline.npz"), n_cases=np.int32(len(spline_cases())), **out)


def gen_lights(r):
    # ---- the reference's own PointLight / SpotLight / DistantLight (SceneTypes/Light.cu, compiled by `make ref` without its two g_SceneData functions; oracle/ref_light_driver.cpp):
    # constructed by the reference from primary parameters, sampleDirect from random reference points
    r.ref_light_sample_direct.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rs = np.random.RandomState(20261005)
    n = 128; out = {}
    lights = []
    for k in range(4):
        lights.append((1, np.concatenate([rs.uniform(-5, 5, 3), rs.uniform(0.5, 40, 3)]).astype(np.float32)))
    for k in range(6):
        p = rs.uniform(-5, 5, 3); t = p + rs.normal(size=3) * 3; cutoff = rs.uniform(10, 60); beam = cutoff * rs.uniform(0.3, 0.95)
        lights.append((4, np.concatenate([p, t, rs.uniform(0.5, 40, 3), [cutoff, beam]]).astype(np.float32)))
    for k in range(4):
        d = rs.normal(size=3); d /= np.linalg.norm(d)
        lights.append((3, np.concatenate([d, rs.uniform(0.5, 5, 3), [rs.uniform(3, 30)]]).astype(np.float32)))
    for i, (typ, par) in enumerate(lights):
        q = np.zeros((n, 8), np.float32); q[:, :3] = rs.uniform(-8, 8, size=(n, 3)); nn = rs.normal(size=(n, 3)); q[:, 3:6] = nn / np.linalg.norm(nn, axis=1, keepdims=True); q[:, 6:8] = rs.rand(n, 2)
        if typ == 4:   # half of the reference points inside the cone
            axis = par[3:6] - par[:3]; axis = axis / np.linalg.norm(axis)
            q[: n // 2, :3] = par[:3] + axis * rs.uniform(0.5, 6, size=(n // 2, 1)) + rs.normal(size=(n // 2, 3)) * 0.4
        if typ == 3:   # three quarters of the reference points beyond the light's disk (centre d * 1.1 r), where the reference accepts the sample (Light.cu:224-233)
            m = 3 * n // 4; dn = par[:3] / np.linalg.norm(par[:3])
            q[:m, :3] = dn * (1.1 * par[6] + rs.uniform(0.05, 12, size=(m, 1))) + rs.normal(size=(m, 3)) * 3
        res = np.zeros((n, 14), np.float32)
        assert r.ref_light_sample_direct(typ, par.ctypes.data, n, q.ctypes.data, res.ctypes.data) == 0
        out["light%d_type" % i] = np.int32(typ); out["light%d_params" % i] = par; out["light%d_q" % i] = q; out["light%d_out" % i] = res
    np.savez_compressed(os.path.join(HERE, "lights.npz"), **out)


def gen_emitters(r):
    # ---- the reference's own emitter SELECTION: KernelDynamicScene::sampleEmitter / pdfEmitter / sampleEmitterDirect (Engine/KernelDynamicScene.cu:25-46, 98-117; `make ref`
    # compiles those line ranges, oracle/ref_emitter_driver.cpp) over light lists given as (CDF, index list into a light buffer).  Case "deleted": the buffer holds one light
    # more than the list (a deleted light in the middle), so list position and buffer slot differ — where pdfEmitter's indexing of the CDF by BUFFER slot shows.
    r.ref_emitter_select.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    r.ref_sample_emitter_direct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rs = np.random.RandomState(20261006)
    out = {}
    def light_params(n_buf):
        types = np.zeros(n_buf, np.int32); params = np.zeros((n_buf, 12), np.float32)
        for i in range(n_buf):
            t = (1, 4, 3)[i % 3]; types[i] = t
            if t == 1: params[i, :6] = np.concatenate([rs.uniform(-5, 5, 3), rs.uniform(0.5, 40, 3)])
            elif t == 4:
                pp = rs.uniform(-5, 5, 3); cutoff = rs.uniform(20, 60)
                params[i, :11] = np.concatenate([pp, pp + rs.normal(size=3) * 3, rs.uniform(0.5, 40, 3), [cutoff, cutoff * rs.uniform(0.3, 0.95)]])
            else:
                d = rs.normal(size=3); params[i, :7] = np.concatenate([d / np.linalg.norm(d), rs.uniform(0.5, 5, 3), [rs.uniform(3, 30)]])
        return types, params
    cases = {"one": (1, [0]), "three": (3, [0, 1, 2]), "seven": (7, list(range(7))), "sixteen": (16, list(range(16))), "deleted": (5, [0, 1, 3, 4]), "deleted_first": (4, [1, 2, 3])}
    for name, (n_buf, idx) in cases.items():
        n = len(idx)
        w = rs.uniform(0.2, 3.0, n); cdf = (np.cumsum(w) / w.sum()).astype(np.float32); cdf[-1] = np.float32(1.0)
        indices = np.array(idx, np.uint32)
        nq = 192
        smp = rs.rand(nq, 2).astype(np.float32)
        edge = np.concatenate([[0.0, np.float32(0.99999994)], cdf[:-1], np.nextafter(cdf[:-1], np.float32(0)), np.nextafter(cdf[:-1], np.float32(2))]).astype(np.float32)
        smp[:len(edge), 0] = edge[:nq]
        slot = np.zeros(nq, np.int32); pdf = np.zeros(nq, np.float32); res = np.zeros(nq, np.float32); pe = np.zeros(16, np.float32)
        assert r.ref_emitter_select(cdf.ctypes.data, indices.ctypes.data, n, n_buf, nq, smp.ctypes.data, slot.ctypes.data, pdf.ctypes.data, res.ctypes.data, pe.ctypes.data) == 0
        types, params = light_params(n_buf)
        q = np.zeros((nq, 8), np.float32); q[:, :3] = rs.uniform(-8, 8, size=(nq, 3)); nn = rs.normal(size=(nq, 3)); q[:, 3:6] = nn / np.linalg.norm(nn, axis=1, keepdims=True); q[:, 6:8] = smp
        o15 = np.zeros((nq, 15), np.float32)
        assert r.ref_sample_emitter_direct(cdf.ctypes.data, indices.ctypes.data, n, n_buf, types.ctypes.data, params.ctypes.data, nq, q.ctypes.data, o15.ctypes.data) == 0
        for k, v in dict(cdf=cdf, indices=indices, n_buf=np.int32(n_buf), samples=smp, slot=slot, pdf=pdf, resampled=res, pdf_emitter=pe[:min(n_buf, 16)], types=types, params=params, q=q, direct=o15).items():
            out["%s_%s" % (name, k)] = v
    np.savez_compressed(os.path.join(HERE, "emitters.npz"), **out)


def image_samples():
    """inputs of image.npz: samples {sx, sy, r, g, b} for a 16 x 12 frame — positions inside, on pixel borders, one ulp below an integer (rounds into the next pixel when the
    jitter is added in float), outside on every side, -0.0, NaN, infinite; radiance positive, negative (clampNegative), -0.0, NaN in one channel, +-inf, huge (own random stream)"""
    rs = np.random.RandomState(20261001)
    W, H, n = 16, 12, 6000
    s = np.zeros((n, 5), np.float32)
    s[:, 0] = rs.uniform(-1.5, W + 1.5, n); s[:, 1] = rs.uniform(-1.5, H + 1.5, n)
    s[:, 2:] = rs.uniform(0, 4, (n, 3)) ** 2
    k = rs.randint(0, n, 900); s[k[:300], 0] = np.floor(s[k[:300], 0]); s[k[300:600], 1] = np.floor(s[k[300:600], 1])
    s[k[600:750], 0] = np.nextafter(np.floor(s[k[600:750], 0]).astype(np.float32), np.float32(-1e9)); s[k[750:900], 1] = np.nextafter(np.ceil(s[k[750:900], 1]).astype(np.float32), np.float32(-1e9))
    special_pos = np.array([-0.0, np.nan, np.inf, -np.inf, W, H, W - 2.0 ** -20, -2.0 ** -30, 0.0, 3.4e38, -3.4e38, 2147483648.0, -2147483904.0], np.float32)
    k = rs.randint(0, n, 400); s[k[:200], 0] = special_pos[rs.randint(0, len(special_pos), 200)]; s[k[200:], 1] = special_pos[rs.randint(0, len(special_pos), 200)]
    special_col = np.array([-0.0, -1.0, -1e-30, np.nan, np.inf, -np.inf, 1e36, 1e-45, 0.0], np.float32)
    k = rs.randint(0, n, 1500); s[k, 2 + rs.randint(0, 3, 1500)] = special_col[rs.randint(0, len(special_col), 1500)]
    return W, H, s


def gen_image(r):
    """Image::AddSample (Engine/Image.cu:22-44) run by the reference's own code (oracle/ref_image_driver.cpp) over image_samples(), in order"""
    W, H, s = image_samples()
    r.ref_image_add_samples.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]; r.ref_image_add_samples.restype = C.c_int
    px = np.zeros((H, W, 7), np.float32)
    assert r.ref_image_add_samples(px.ctypes.data, W, H, len(s), s.ctypes.data) == 0
    with np.errstate(invalid="ignore"):
        assert np.isfinite(px).all() and 0.5 * len(s) < px[..., 6].sum() < len(s) and (px[..., 3:6] == 0).all()
    np.savez_compressed(os.path.join(HERE, "image.npz"), width=np.int32(W), height=np.int32(H), samples=s, pixels=px)


if __name__ == "__main__":
    if sys.argv[1:] == ["image"]:
        gen_image(oracle.load_ref())
    elif sys.argv[1:] == ["emitters"]:
        gen_emitters(oracle.load_ref())
    elif sys.argv[1:] == ["lights"]:
        gen_lights(oracle.load_ref())
        gen_spline(oracle.load_ref())
        gen_bsdf_rough(oracle.load_ref())
        gen_scene_lights(oracle.load_ref())
    elif sys.argv[1:] == ["bsdf"]:
        gen_bsdf(oracle.load_ref())
        gen_lights(oracle.load_ref())
        gen_spline(oracle.load_ref())
        gen_bsdf_rough(oracle.load_ref())
        gen_scene_lights(oracle.load_ref())
    elif sys.argv[1:] == ["mipmap"]:
        gen_mipmap(oracle.load_ref())
    elif sys.argv[1:] == ["math2"]:
        gen_math2(oracle.load_ref())
    elif sys.argv[1:] == ["scene_lights"]:
        gen_scene_lights(oracle.load_ref())
    elif sys.argv[1:] == ["bsdf_rough"]:
        gen_bsdf_rough(oracle.load_ref())
        gen_scene_lights(oracle.load_ref())
    elif sys.argv[1:] == ["spline"]:
        gen_spline(oracle.load_ref())
        gen_bsdf_rough(oracle.load_ref())
        gen_scene_lights(oracle.load_ref())
    elif sys.argv[1:] == ["material_maps"]:
        gen_material_maps(oracle.load_ref())
    elif sys.argv[1:] == ["traceray"]:      # only this fixture (the others stay byte-identical)
        gen_traceray(oracle.load_ref())
    else:
        main()
        gen_traceray(oracle.load_ref())
        gen_math2(oracle.load_ref())
        gen_mipmap(oracle.load_ref())
        gen_bsdf(oracle.load_ref())
        gen_emitters(oracle.load_ref())
        gen_lights(oracle.load_ref())
        gen_spline(oracle.load_ref())
        gen_bsdf_rough(oracle.load_ref())
        gen_scene_lights(oracle.load_ref())
        gen_material_maps(oracle.load_ref())
        gen_image(oracle.load_ref())
