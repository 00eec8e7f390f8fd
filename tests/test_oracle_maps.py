"""Material maps of the oracle: Material::SampleNormalMap (normal / height map, Engine/Material.cu:96-138), Material::AlphaTest
(:141-190) and the alpha test inside single-ray traversal (Kernel/TraceHelper.cu:135-153).

Since round 5 Engine/Material.cu itself is part of oracle/_ref and tests/test_oracle_golden.py holds the restatement on it bit for bit
(tests/golden/material_maps.npz); the closed forms here say what the functions MEAN: a flat normal-map texel leaves the frame alone, a tilted one gives toWorld(c - 0.5); a height ramp gives the analytic
normal of the displaced plane; the alpha tests are threshold functions of hand-made textures; rays through the holes of an
alpha-mapped card reach the wall behind it.
"""
import ctypes as C
import numpy as np
import pytest
import oracle
from cudatracerlib_amd import api, scenes


@pytest.fixture(scope="module")
def lib():
    return oracle.load()


def _frame(lib, desc, mat, uv=(0.3, 0.6), s=(1, 0, 0), t=(0, 0, -1), n=(0, 1, 0), dpdu=None, dpdv=None):
    f = np.array([*s, *t, *n], np.float32)
    geo = np.array([*n, *(dpdu if dpdu is not None else s), *(dpdv if dpdv is not None else t)], np.float32)
    used = lib.orc_sample_normal_map(C.byref(desc) if desc is not None else None, C.byref(mat), float(uv[0]), float(uv[1]), f.ctypes.data, geo.ctypes.data)
    return used, f.reshape(3, 3)


def test_normal_map_constant_texels(lib):
    m = api.set_normal_map(api.diffuse(), (0.5, 0.5, 1.0))        # flat: n stays, the frame is rebuilt around it
    used, f = _frame(lib, None, m)
    assert used == 1
    assert np.allclose(f[2], (0, 1, 0), atol=1e-6)
    assert np.allclose(f @ f.T, np.eye(3), atol=1e-6)
    c = np.array((0.8, 0.4, 0.9), np.float32)
    m = api.set_normal_map(api.diffuse(), tuple(c))
    used, f = _frame(lib, None, m)
    local = c - 0.5
    want = local[0] * np.array((1, 0, 0)) + local[1] * np.array((0, 0, -1)) + local[2] * np.array((0, 1, 0))
    assert np.allclose(f[2], want / np.linalg.norm(want), atol=1e-6)
    # t = normalize(n x s_old), s = normalize(n x t)  (Material.cu:104-105)
    t = np.cross(f[2], (1, 0, 0)); t /= np.linalg.norm(t)
    assert np.allclose(f[1], t, atol=1e-6) and np.allclose(f[0], np.cross(f[2], t) / np.linalg.norm(np.cross(f[2], t)), atol=1e-6)


def test_no_map_and_constant_height_texture_are_noops(lib):
    used, f = _frame(lib, None, api.diffuse())
    assert used == 0 and np.array_equal(f, np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32))
    m = api.set_height_map(api.diffuse(), 0.7)                   # HeightMap.tex.Is<ImageTexture>() fails (Material.cu:109)
    used, f = _frame(lib, None, m)
    assert used == 0
    with pytest.raises(api.CtlError):
        api.set_normal_map(m, 0.5)


def test_height_ramp_gives_the_analytic_normal(lib):
    sc = api.DynamicScene()
    W = 64
    ramp = np.repeat((np.arange(W, dtype=np.float32) / 255.0)[None, :, None], 16, axis=0).repeat(3, axis=2)   # exactly representable texels: x / 255
    img = sc.add_image(api.float3_to_rgbcol(ramp), api.TEXEL_RGBCOL, api.WRAP_CLAMP, api.FILTER_BILINEAR)
    m = api.set_height_map(api.diffuse(), api.image_texture(img))
    sc.CreateNode(sc.add_mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32), None, materials=[m]))
    sc.setCamera((0, 3, 0), (0, 0, 0), (0, 0, 1), 40.0, 8, 8); sc.UpdateScene()
    d = sc.desc
    used, f = _frame(lib, d, m, uv=(0.5, 0.5))
    assert used == 1
    g_u = W * (1.0 / 255.0)                                        # luminance of d(height)/du: (texel step) * width, weights sum to 1
    # dpdu' = s + n g_u, dpdv' = t  ->  n' = normalize(dpdu' x dpdv') with s = x, t = -z, n = y: (x + y g) x (-z) = y - g x ... sign per cross
    want = np.cross(np.array((1, g_u, 0.0)), np.array((0, 0, -1.0)))
    want /= np.linalg.norm(want)
    if np.dot(want, (0, 1, 0)) < 0:
        want = -want
    assert np.allclose(f[2], want, atol=2e-5)
    assert np.allclose(f @ f.T, np.eye(3), atol=1e-5)
    assert abs(np.dot(f[0], f[2])) < 1e-6 and f[0][0] > 0          # s follows dpdu'


def test_alpha_test_modes(lib):
    def survives(mat, u, v, desc=None):
        return lib.orc_alpha_test(C.byref(desc) if desc is not None else None, C.byref(mat), float(u), float(v)) == 1
    plain = api.diffuse()
    assert survives(plain, 0.1, 0.1)                               # Disabled
    chk = api.checker_texture(1.0, 0.0, uv_scale=(1.0, 1.0))
    m = api.set_alpha_map(api.diffuse(), chk, api.ALPHA_MAP_LUMINANCE, 0.5)
    assert survives(m, 0.1, 0.1) and not survives(m, 0.6, 0.1) and survives(m, 0.6, 0.6)
    m = api.set_alpha_map(api.diffuse(), (0.2, 0.4, 0.6), api.ALPHA_MAP_COLOR, 0.15, (0.3, 0.3, 0.6))
    assert survives(m, 0, 0)                                       # max |d| = 0.1 <= 0.15
    m.alpha_test_scalar = 0.05
    assert not survives(m, 0, 0)
    m = api.set_alpha_map(api.diffuse((0.9, 0.9, 0.9)), 0.0, api.ALPHA_REFLECTANCE_LUMINANCE, 0.5)   # bit 2: test the BSDF's first texture
    assert survives(m, 0, 0)
    m = api.set_alpha_map(api.diffuse((0.1, 0.1, 0.1)), 1.0, api.ALPHA_REFLECTANCE_LUMINANCE, 0.5)
    assert not survives(m, 0, 0)
    m = api.set_alpha_map(api.diffuse(), 0.0, api.ALPHA_MAP_ALPHA, 0.5)     # alpha mode on a non-image texture: always passes (Material.cu:186)
    assert survives(m, 0, 0)
    # alpha channel of an RGBCOL bitmap; RGBE bitmaps are opaque (MIPMap.cu:135-137)
    sc = api.DynamicScene()
    tex = np.array([[0xff000000, 0x10000000], [0x80000000, 0x7f000000]], np.uint32)
    i0 = sc.add_image(tex, api.TEXEL_RGBCOL, api.WRAP_REPEAT, api.FILTER_POINT)
    i1 = sc.add_image(tex, api.TEXEL_RGBE, api.WRAP_REPEAT, api.FILTER_POINT)
    sc.CreateNode(sc.add_mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32), None))
    sc.setCamera((0, 3, 0), (0, 0, 0), (0, 0, 1), 40.0, 8, 8); sc.UpdateScene()
    d = sc.desc
    m = api.set_alpha_map(api.diffuse(), api.image_texture(i0), api.ALPHA_MAP_ALPHA, 0.5)
    # REPEAT wrap flips v: uv (0.25, 0.75) -> texel row 0 (MIPMap_device.h:36)
    got = [[survives(m, 0.25, 0.75, d), survives(m, 0.75, 0.75, d)], [survives(m, 0.25, 0.25, d), survives(m, 0.75, 0.25, d)]]
    assert got == [[True, False], [True, False]]                  # 255, 16 / 128, 127 of 255 against 0.5
    m = api.set_alpha_map(api.diffuse(), api.image_texture(i1), api.ALPHA_MAP_ALPHA, 0.5)
    assert survives(m, 0.75, 0.75, d)


def test_trace_ray_alpha_test_lets_rays_through_the_holes(orc):
    sc = scenes.maps_scene(32, 24, None, "luminance")
    d = sc.desc
    # rays from the camera side straight at the card (z = -1, x in [-3, 3], y in [0.2, 4.2]); checker 4 x 3 over uv in [0, 1]
    xs, ys = np.meshgrid(np.linspace(-2.9, 2.9, 24), np.linspace(0.3, 4.1, 16))
    rays = np.zeros((xs.size, 8), np.float32)
    rays[:, 0] = xs.ravel(); rays[:, 1] = ys.ravel(); rays[:, 2] = 5.0; rays[:, 3] = 1e-3
    rays[:, 6] = -1.0; rays[:, 7] = 1e30
    plain = orc.intersect(d, rays)
    alpha = orc.intersect(d, rays, alpha_test=True)
    assert np.allclose(plain["dist"], 6.0, atol=1e-4)              # every ray stops at the card without the test
    u, v = (xs.ravel() + 3) / 6, (ys.ravel() - 0.2) / 4
    u, v = v, u      # the alpha test interpolates getUVSetData's pairs (TraceHelper.cu:149), which come out (v, u) (TriangleData.cu:27 against fillDG's :94): the reference's own exchange
    solid = ((np.floor(u * 4 * 2).astype(int) % 2) * 2 - 1) * ((np.floor(v * 3 * 2).astype(int) % 2) * 2 - 1) == 1   # CheckerboardTexture (Texture.h:136-146)
    edge = (np.abs(u * 8 - np.round(u * 8)) < 1e-3) | (np.abs(v * 6 - np.round(v * 6)) < 1e-3)
    ok = ~edge
    assert np.allclose(alpha["dist"][ok & solid], 6.0, atol=1e-4)
    assert np.allclose(alpha["dist"][ok & ~solid], 9.0, atol=1e-4)   # the wall at z = -4
    # any-hit rays (shadow rays) see the same holes
    occ_plain = orc.intersect(d, rays, any_hit=True)
    occ_alpha = orc.intersect(d, rays, any_hit=True, alpha_test=True)
    assert (occ_plain["tri_idx"] >= 0).all() and (occ_alpha["tri_idx"] >= 0).all()   # the wall still stops them
    rays[:, 7] = 7.0                                               # tmax between card and wall
    occ_alpha = orc.intersect(d, rays, any_hit=True, alpha_test=True)
    assert ((occ_alpha["tri_idx"] >= 0) == solid)[ok].all()


def test_render_differs_only_where_maps_act(orc):
    plain = scenes.maps_scene(48, 32, None, None)                  # (the descriptor points into the scene object: keep it alive)
    base, _ = orc.render(plain.desc, 48, 32, n_passes=1, max_path_length=3)
    card = scenes.maps_scene(48, 32, None, "luminance")
    off, _ = orc.render(card.desc, 48, 32, n_passes=1, max_path_length=3, alpha_test=False)
    on, _ = orc.render(card.desc, 48, 32, n_passes=1, max_path_length=3, alpha_test=True)
    assert np.array_equal(base, off)                               # an alpha map without the test changes nothing
    assert not np.array_equal(off, on)
