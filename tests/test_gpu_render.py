"""Wavefront HIP path tracer vs the oracle's PathTrace<DIRECT> (Integrators/PathTracer.cu:10-113) on identical
scenes and identical sampler tables.

Tolerance (north_star: "within a stated per-pixel float tolerance"): both sides run the same fp32 expressions without FMA contraction AND the same
transcendental functions — the kernels and the checker (the oracle's shared-math build, oracle/liboracle_sm.so) both compile csrc/ctl_fmath.h — so the bar is
  * >= 99.95 % of pixels: |gpu - cpu| <= 2e-3 * (1 + cpu) per channel of the accumulated radiance sum,
  * >= 98 % of pixels equal to the BIT (what is left: sums of float atomics in another order, texture filtering through the device's own exp2 / log2), and
  * the image means agree to 1e-3 relative.
The glibc build of the oracle — the reference's CPU path, pinned on the reference's own code — differs from the shared-math build by <= 1 ulp per function
call (tests/test_fmath.py), which is what used to separate GPU and checker.
"""
import os
import numpy as np
import pytest
from cudatracerlib_amd import scenes

pytestmark = pytest.mark.gpu


def render_pair(gpu, orc, sc, w, h, n_passes, max_len=8, rr=5, direct=True):
    d = sc.desc
    tables = orc.sequence_tables(n_passes)
    want, want_rays = orc.render(d, w, h, n_passes=n_passes, tables=tables, direct=direct, max_path_length=max_len, rr_start=rr)
    scene = gpu.Scene(d)
    tr = gpu.WavefrontPathTracer()
    p = tr.getParameters()
    p.setValue("Direct", direct); p.setValue("MaxPathLength", max_len); p.setValue("RRStartDepth", rr)
    tr.Resize(w, h); tr.InitializeScene(scene)
    img = gpu.Image(w, h)
    for k in range(n_passes):
        tr.setSamplerTables(*tables[k])
        tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData()
    return got, want, tr, want_rays


def assert_close(got, want, exact_min=0.98):
    assert np.array_equal(got[..., 6], want[..., 6]), "weightSum differs"
    g, w = got[..., :3], want[..., :3]
    ok = np.abs(g - w) <= 2e-3 * (1 + np.abs(w))
    frac = ok.all(axis=2).mean()
    assert frac >= 0.9995, frac
    exact = (g == w).all(axis=2).mean()     # the shading code and the checker (the oracle's shared-math build) run the same fp32 functions: most pixels agree to the bit
    assert exact >= exact_min, exact
    assert abs(g.mean() - w.mean()) <= 1e-3 * w.mean()


def test_cornell_diffuse(gpu, orc):
    sc = scenes.cornell_box(64, 64)
    got, want, tr, rays = render_pair(gpu, orc, sc, 64, 64, 4)
    assert_close(got, want)
    assert tr.getNumPassesDone() == 4


def test_cornell_glass_sphere(gpu, orc):
    sc = scenes.cornell_box(96, 96, glass_sphere=True)
    got, want, tr, rays = render_pair(gpu, orc, sc, 96, 96, 3)
    assert_close(got, want)


def test_cornell_microfacet_and_conductor(gpu, orc):
    sc = scenes.cornell_box(64, 64, extra_materials=True)
    got, want, tr, rays = render_pair(gpu, orc, sc, 64, 64, 3)
    assert_close(got, want)


@pytest.mark.parametrize("variant", [8, 9, 10])
def test_cornell_coating_roughcoating_blend(gpu, orc, variant):
    """the nesting BSDFs (BSDF_Complex.cu) over diffuse / metal / plastic children, incl. delta children (discrete-measure f / pdf)"""
    sc = scenes.cornell_box(64, 64, extra_materials=variant)
    got, want, tr, rays = render_pair(gpu, orc, sc, 64, 64, 3)
    assert_close(got, want)


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7, 11, 12])
def test_cornell_plastic_roughdielectric_phong_thindielectric(gpu, orc, variant):
    """2: plastic + GGX rough glass, 3: phong + thin glass, 4: nonlinear plastic + anisotropic Beckmann rough glass,
    5: Oren-Nayar + balanced Ward, 6: Beckmann / GGX rough plastic (rough-transmittance tables), 7: fast Oren-Nayar + Ward,
    11 / 12: Beckmann sampled from the visible normals (erf / erfinv iteration) and the Phong microfacet distribution"""
    sc = scenes.cornell_box(64, 64, extra_materials=variant)
    got, want, tr, rays = render_pair(gpu, orc, sc, 64, 64, 3)
    # 6: the rough plastics run the reference's own 3-D table lookup on the device (round 5; the per-material 1-D reduction is opt-in, test_reduced_rough_transmittance_is_opt_in)
    assert_close(got, want)


def test_reduced_rough_transmittance_is_opt_in(gpu, orc):
    """CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE: rough plastic through the per-material 1-D reduction of RoughTransmittanceManager's table — within the per-pixel tolerance here, NOT equal
    to the bit (the default scene of the same description is: test_cornell_plastic_roughdielectric_phong_thindielectric[6])"""
    sc = scenes.cornell_box(64, 64, extra_materials=6)
    d = sc.desc
    tables = orc.sequence_tables(3)
    want, _ = orc.render(d, 64, 64, n_passes=3, tables=tables)
    frames = []
    for reduced in (False, True):
        tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("Direct", True); p.setValue("MaxPathLength", 8); p.setValue("RRStartDepth", 5)
        tr.Resize(64, 64); tr.InitializeScene(gpu.Scene(d, reduced_rough_transmittance=reduced))
        img = gpu.Image(64, 64)
        for k in range(3):
            tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
        frames.append(img.getPixelData())
    assert_close(frames[0], want)
    assert_close(frames[1], want, exact_min=0.3)
    assert not np.array_equal(frames[0][..., :3], frames[1][..., :3])


@pytest.mark.parametrize("kw", [dict(), dict(rotate_env=True, point_filter=True), dict(extra_lights=True)])
def test_environment_map_bitmap_texture_and_delta_lights(gpu, orc, kw):
    """InfiniteLight NEE + miss MIS, ImageTexture (bilinear / point), spot + distant + point lights, 6 BSDF types in one frame"""
    sc = scenes.env_scene(96, 64, **kw)
    got, want, tr, rays = render_pair(gpu, orc, sc, 96, 64, 3)
    assert_close(got, want)
    assert want[..., :3].mean() > 0.1


@pytest.mark.parametrize("kind", ["checker", "image", "orthogonal", "orthogonal_image"])
def test_textured_and_orthogonal_area_lights(gpu, orc, kind):
    """DiffuseLight::m_rad_texture that needs uv (NEE: uv of the sampled triangle; emitter hit: ShapeSet::getPosition finds the triangle again) and
    m_bOrthogonal (discrete-measure NEE straight along the normal, eval only within DeltaEpsilon of it) — wavefront and megakernel plugins"""
    sc = scenes.area_lights_scene(96, 64, kind)
    got, want, tr, rays = render_pair(gpu, orc, sc, 96, 64, 3)
    assert_close(got, want)
    assert want[..., :3].mean() > 0.02
    tables = orc.sequence_tables(3)
    mk = gpu.PathTracer(); mk.Resize(96, 64); mk.InitializeScene(gpu.Scene(sc.desc, flatten=True))
    img = gpu.Image(96, 64)
    for k in range(3):
        mk.setSamplerTables(*tables[k]); mk.DoPass(img, new_trace=(k == 0))
    assert_close(img.getPixelData(), want)


def test_synthetic_bathroom_workload(gpu, orc):
    """the stand-in for BASELINE config 5 in miniature: nine BSDF models (both distributions, visible normals, coating, rough glass),
    bitmap texture + height map, environment emitter + area light, instanced meshes — two-level layout, per-pixel bar"""
    sc = scenes.synthetic_bathroom(96, 54, n_instances=60, subdiv=2)
    got, want, tr, rays = render_pair(gpu, orc, sc, 96, 54, 3, max_len=6)
    assert_close(got, want)      # rough plastic everywhere: bit-equal frames since the device runs the reference's 3-D transmittance lookup (round 5: 5 % of the pixels with the reduced tables, 100 % without)
    assert want[..., :3].mean() > 0.1


def test_queue_ordering_options_do_not_change_the_image(gpu, orc):
    """FuseTraversal (path rays of a bounce and shadow rays of the previous one in one persistent launch, or in two), BlockSort (each workgroup of the full shade kernel regroups its paths by BSDF model), ShadeByModelClass (one shade launch per model class of the scene, or the one kernel with every model), SortMaterials (shade in BSDF-model order) and SortOctants (append new rays grouped by direction octant) only reorder work:
    same frame (up to the order of the float atomics) and same ray count as the default order"""
    sc = scenes.synthetic_bathroom(96, 54, n_instances=60, subdiv=2)
    scene = gpu.Scene(sc.desc, flatten=True)
    tables = orc.sequence_tables(3)
    out = []
    for params in (dict(), dict(SortMaterials=True), dict(SortOctants=True), dict(SortMaterials=True, SortOctants=True), dict(BlockSort=False), dict(BlockSort=False, SortMaterials=True), dict(FuseTraversal=False), dict(ShadeByModelClass=False), dict(ShadeByModelClass=False, BlockSort=False)):
        tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 6)
        for k, v in params.items():
            p.setValue(k, v)
        tr.Resize(96, 54); tr.InitializeScene(scene)
        img = gpu.Image(96, 54)
        for k in range(3):
            tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
        out.append((img.getPixelData(), tr.stats().rays_total))
    for got, rays in out[1:]:
        assert rays == out[0][1]
        assert np.array_equal(got[..., 6], out[0][0][..., 6])
        assert np.allclose(got[..., :3], out[0][0][..., :3], rtol=1e-5, atol=1e-5)


def test_mitsuba_xml_scene(gpu, orc, tmp_path):
    """ParseMitsubaScene -> UpdateScene -> render: Cornell box + glass sphere authored as Mitsuba XML (BASELINE config 2 geometry)"""
    path = scenes.write_cornell_mitsuba(str(tmp_path), 64, 64, glass_sphere=True)
    sc = gpu.DynamicScene()
    assert sc.ParseMitsubaScene(path) == (64, 64)
    sc.UpdateScene()
    got, want, tr, rays = render_pair(gpu, orc, sc, 64, 64, 3)
    assert_close(got, want)


def test_tungsten_style_interior_through_the_loader(gpu, orc, tmp_path):
    """A scene file in the form the BASELINE distributions take (scenes.write_interior_mitsuba: matrix transforms, twosided wrappers by id, OBJ shapes, trilinear
    bitmap on a rough plastic, glass, rough conductor, rectangle emitter, rotated .hdr environment) -> ParseMitsubaScene -> both plugins against the oracle: the
    wavefront plugin at texture level 0, the PathTracer plugin with first-hit ray differentials (trilinear lookup of the floor's bitmap)."""
    from cudatracerlib_amd import rough_tables
    w, h, n = 128, 72, 3
    path = scenes.write_interior_mitsuba(str(tmp_path), w, h)
    sc = gpu.DynamicScene()
    tr_, df, er, ar = rough_tables.make_table(1, n_eta=3, n_alpha=3, n_theta=4, quad=8)
    sc.setRoughTransmittance(1, tr_, df, er, ar)
    assert sc.ParseMitsubaScene(path) == (w, h)
    d = sc.UpdateScene()
    tables = orc.sequence_tables(n)
    scene = gpu.Scene(d, flatten=True)
    for cls, partials in ((gpu.WavefrontPathTracer, False), (gpu.PathTracer, True)):
        want, _ = orc.render(d, w, h, n_passes=n, tables=tables, max_path_length=8, partials=partials)
        got = _render(gpu, cls, scene, tables, w, h, max_len=8)
        assert_close(got, want, exact_min=0.98 if not partials else 0.5)    # (the PathTracer plugin filters textures through the device's own arithmetic order)
        assert want[..., :3].mean() > 0.05


@pytest.mark.parametrize("scene_kw", [dict(glass_sphere=True), dict(extra_materials=6), dict(extra_materials=8)])
def test_megakernel_path_tracer_plugin(gpu, orc, scene_kw):
    """ctl_tracer_create("PathTracer"): PathTrace<DIRECT> as one kernel — same image as the oracle (and as the wavefront tracer),
    same ray count as the wavefront tracer"""
    sc = scenes.cornell_box(64, 64, **scene_kw)
    d = sc.desc
    tables = orc.sequence_tables(3)
    want, _ = orc.render(d, 64, 64, n_passes=3, tables=tables, max_path_length=8, partials=True)
    scene = gpu.Scene(d, flatten=True)
    out = {}
    for cls in (gpu.PathTracer, gpu.WavefrontPathTracer):
        tr = cls(); tr.getParameters().setValue("MaxPathLength", 8)
        tr.Resize(64, 64); tr.InitializeScene(scene)
        img = gpu.Image(64, 64)
        for k in range(3):
            tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
        out[cls.__name__] = (img.getPixelData(), tr.stats().rays_total)
    got, rays_mega = out["PathTracer"]
    g, w = got[..., :3], want[..., :3]
    assert np.array_equal(got[..., 6], want[..., 6])
    assert (np.abs(g - w) <= 2e-3 * (1 + np.abs(w))).all(axis=2).mean() >= 0.995       # the flattened layout reports the reference's t,u,v
    assert abs(g.mean() - w.mean()) <= 1e-3 * w.mean()
    wave, rays_wave = out["WavefrontPathTracer"]
    assert np.isclose(got[..., :3], wave[..., :3], rtol=1e-3, atol=1e-3).all(axis=2).mean() >= 0.995
    assert abs(int(rays_mega) - int(rays_wave)) <= 1e-3 * rays_wave
    with pytest.raises(gpu.CtlError):
        tr = gpu.PathTracer(); tr.Resize(64, 64); tr.InitializeScene(gpu.Scene(d))        # needs the flattened layout


def _render(gpu, cls, scene, tables, w, h, max_len=5, **params):
    tr = cls()
    p = tr.getParameters(); p.setValue("MaxPathLength", max_len)
    for k, v in params.items():
        p.setValue(k, v)
    tr.Resize(w, h); tr.InitializeScene(scene)
    img = gpu.Image(w, h)
    for k in range(len(tables)):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    return img.getPixelData()


@pytest.mark.parametrize("surface_map,alpha", [("normal", "luminance"), ("height", "alpha"), (None, "color")])
def test_material_maps(gpu, orc, surface_map, alpha):
    """Normal / height maps in the shading stage (Material::SampleNormalMap) and the alpha test in traversal (Material::AlphaTest):
    the wavefront tracer ignores alpha maps by default as the reference's intersectKernel does, tests them with AlphaTest=true, and
    the megakernel PathTracer always tests them (TraceHelper.cu:135-153, 179)."""
    w, h = 96, 64
    sc = scenes.maps_scene(w, h, surface_map, alpha)
    d = sc.desc
    tables = orc.sequence_tables(2)
    want_off, _ = orc.render(d, w, h, n_passes=2, tables=tables, max_path_length=5, alpha_test=False)
    want_on, _ = orc.render(d, w, h, n_passes=2, tables=tables, max_path_length=5, alpha_test=True)
    assert not np.array_equal(want_off, want_on)
    two_level, flat = gpu.Scene(d), gpu.Scene(d, flatten=True)
    assert_close(_render(gpu, gpu.WavefrontPathTracer, two_level, tables, w, h), want_off)
    assert_close(_render(gpu, gpu.WavefrontPathTracer, two_level, tables, w, h, AlphaTest=True), want_on)

    def close_flat(got, want):                                       # flattened layout: the same bar as the two-level one
        g, wv = got[..., :3], want[..., :3]
        assert np.array_equal(got[..., 6], want[..., 6])
        assert (np.abs(g - wv) <= 2e-3 * (1 + np.abs(wv))).all(axis=2).mean() >= 0.995
        assert abs(g.mean() - wv.mean()) <= 1e-3 * wv.mean()
    close_flat(_render(gpu, gpu.WavefrontPathTracer, flat, tables, w, h, AlphaTest=True), want_on)
    close_flat(_render(gpu, gpu.WavefrontPathTracer, flat, tables, w, h), want_off)
    want_mega, _ = orc.render(d, w, h, n_passes=2, tables=tables, max_path_length=5, alpha_test=True, partials=True)   # the megakernel integrator: first-hit ray differentials
    close_flat(_render(gpu, gpu.PathTracer, flat, tables, w, h), want_mega)


def test_image_pipeline_filters_and_tonemap(gpu):
    """applyImagePipeline with a CanonicalFilter and / or the Reinhard tone mapper against the numpy oracle (oracle/pipeline.py).
    8-bit outputs may differ by one step where expf / powf / the atomic log-average differ in the last ulp."""
    from oracle import pipeline as P
    sc = scenes.cornell_box(48, 40, glass_sphere=True)
    scene = gpu.Scene(sc.desc)
    tr = gpu.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 4)
    tr.Resize(48, 40); tr.InitializeScene(scene)
    img = gpu.Image(48, 40)
    tr.DoPasses(img, 4, new_trace=True)
    px = img.getPixelData()
    api = gpu.api
    cases = [(api.box_filter(1.0, 2.0), None), (api.gaussian_filter(2.0, 2.0, 2.0), None), (api.mitchell_filter(), None), (api.lanczos_filter(3.0, 3.0, 3.0), None),
             (api.triangle_filter(2.0, 1.5), None), (None, api.tonemap()), (api.triangle_filter(), api.tonemap(0.3, 0.2)), (api.gaussian_filter(1.5, 1.5, 1.0), api.tonemap(0.18, 0.0))]
    for flt, proc in cases:
        got = img.applyImagePipeline(0.25, flt, proc)
        f = None if flt is None else dict(type=flt.type, xw=flt.x_width, yw=flt.y_width, p0=flt.p0, p1=flt.p1)
        p = None if proc is None else dict(key=proc.key, burn=proc.burn)
        want = P.apply_image_pipeline(px, 0.25, f, p)
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 2 and (d > 0).mean() <= 0.03, (None if flt is None else flt.type, proc is not None, d.max(), (d > 0).mean())
    assert np.array_equal(img.applyImagePipeline(0.25), P.apply_image_pipeline(px, 0.25)) or np.abs(img.applyImagePipeline(0.25).astype(int) - P.apply_image_pipeline(px, 0.25).astype(int)).max() <= 1
    with pytest.raises(gpu.CtlError):
        img.applyImagePipeline(0.0, api.ctl_reconstruction_filter(9, 1.0, 1.0, 0.0, 0.0), None)


def test_image_pipeline_and_output_files(gpu, tmp_path):
    """applyImagePipeline (no filter / post-process) = toSpectrum(splatScale) -> sRGB curve -> RGBCOL; WriteDisplayImage"""
    import struct, zlib
    sc = scenes.cornell_box(40, 32)
    scene = gpu.Scene(sc.desc)
    tr = gpu.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 4)
    tr.Resize(40, 32); tr.InitializeScene(scene)
    img = gpu.Image(40, 32)
    tr.DoPasses(img, 4, new_trace=True)
    px = img.getPixelData()
    lin = px[..., :3] / np.where(px[..., 6:7] != 0, px[..., 6:7], 1) + px[..., 3:6] * 0.25
    assert np.allclose(img.getRGB(0.25), lin, rtol=1e-6, atol=1e-7)
    srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(np.maximum(lin, 0), 1 / 2.4) - 0.055)
    want = (np.clip(srgb, 0, 1) * 255).astype(np.uint8)
    got = img.applyImagePipeline(0.25)
    assert np.all(got[..., 3] == 255)
    assert np.abs(got[..., :3].astype(int) - want.astype(int)).max() <= 1        # powf vs numpy at a truncation boundary
    for ext in ("png", "hdr", "pfm"):
        img.WriteDisplayImage(str(tmp_path / ("out." + ext)), 0.25)
    raw = open(tmp_path / "out.png", "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (40, 32)
    idat = raw[raw.index(b"IDAT") + 4: raw.index(b"IEND") - 8]
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(32, 1 + 40 * 3)
    assert np.all(rows[:, 0] == 0) and np.array_equal(rows[:, 1:].reshape(32, 40, 3), got[..., :3])
    pfm = open(tmp_path / "out.pfm", "rb").read()
    head = b"PF\n40 32\n-1.0\n"
    assert pfm.startswith(head) and np.allclose(np.frombuffer(pfm[len(head):], "<f4").reshape(32, 40, 3)[::-1], img.getRGB(0.25))
    assert open(tmp_path / "out.hdr", "rb").read().startswith(b"#?RADIANCE")
    with pytest.raises(gpu.CtlError):
        img.WriteDisplayImage(str(tmp_path / "out.jpg"))


def test_environment_map_without_nee(gpu, orc):
    sc = scenes.env_scene(64, 48)
    got, want, _, _ = render_pair(gpu, orc, sc, 64, 48, 2, max_len=4, rr=2, direct=False)
    assert_close(got, want)


def test_no_direct_and_short_paths(gpu, orc):
    sc = scenes.cornell_box(48, 48)
    got, want, _, _ = render_pair(gpu, orc, sc, 48, 48, 2, max_len=3, rr=1, direct=False)
    assert_close(got, want)


def test_ray_count_matches_oracle_within_zero_throughput_shortcut(gpu, orc):
    """rays = primary + continuation + shadow (TraceHelper.cu:176,745); the HIP tracer drops zero-throughput paths early."""
    sc = scenes.cornell_box(64, 64)
    got, want, tr, want_rays = render_pair(gpu, orc, sc, 64, 64, 2)
    rays = tr.stats().rays_total
    assert 0.9 * want_rays <= rays <= want_rays


def test_tile_shards_compose(gpu, orc):
    """image-tile sharding: the sum of the ranks' framebuffers == the single-rank framebuffer (disjoint tiles)."""
    sc = scenes.cornell_box(160, 96)
    d = sc.desc
    tables = orc.sequence_tables(2)
    scene = gpu.Scene(d)

    def run(rank, world):
        tr = gpu.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 6)
        tr.setTileShard(rank, world); tr.Resize(160, 96); tr.InitializeScene(scene)
        img = gpu.Image(160, 96)
        for k in range(2):
            tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
        return img.getPixelData()
    full = run(0, 1)
    parts = [run(r, 3) for r in range(3)]
    s = sum(parts)
    assert np.array_equal(s[..., 6], full[..., 6])
    # a sample may land in a neighbouring tile's pixel (floor(x + jitter)), so compare sums with float tolerance
    assert np.allclose(s[..., :3], full[..., :3], rtol=1e-5, atol=1e-5)


def test_parameters_and_errors(gpu):
    tr = gpu.WavefrontPathTracer()
    p = tr.getParameters()
    assert p.getValue("MaxPathLength") == 50 and p.getValue("RRStartDepth") == 5 and p.getValue("Direct") == 1
    with pytest.raises(gpu.CtlError):
        p.setValue("MaxPathLength", 0)
    with pytest.raises(gpu.CtlError):
        p.setValue("NoSuchKey", 1)
    img = gpu.Image(8, 8)
    with pytest.raises(gpu.CtlError):
        tr.DoPass(img)   # no Resize / InitializeScene yet


def test_flattened_scene_renders_like_two_level(gpu, orc):
    """the flattened re-layout reports the reference's t,u,v bit for bit -> the two-level tolerance applies unchanged"""
    sc = scenes.synthetic_sm(96, 64, n_instances=120, subdiv=2)
    d = sc.desc
    tables = orc.sequence_tables(2)
    want, _ = orc.render(d, 96, 64, n_passes=2, tables=tables, max_path_length=5)
    scene = gpu.Scene(d, flatten=True)
    tr = gpu.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 5)
    tr.Resize(96, 64); tr.InitializeScene(scene)
    img = gpu.Image(96, 64)
    for k in range(2):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData()
    g, w = got[..., :3], want[..., :3]
    ok = (np.abs(g - w) <= 2e-3 * (1 + np.abs(w))).all(axis=2).mean()
    assert ok >= 0.995, ok
    assert abs(g.mean() - w.mean()) <= 1e-3 * w.mean()


def test_pass_batching_is_equivalent(gpu, orc):
    """PassBatch = B renders B passes in one wavefront; every path uses its own pass's tables -> same image as B single passes"""
    sc = scenes.cornell_box(64, 64, glass_sphere=True)
    scene = gpu.Scene(sc.desc)

    def run(batch):
        tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 6); p.setValue("PassBatch", batch)
        tr.Resize(64, 64); tr.InitializeScene(scene)
        img = gpu.Image(64, 64)
        tr.DoPasses(img, 4, new_trace=True)      # the tracer's own generator: passes 1..4 of a fresh XORWOW stream
        return img.getPixelData(), tr.stats().rays_total
    a, ra = run(1)
    b, rb = run(4)
    assert ra == rb
    assert np.array_equal(a[..., 6], b[..., 6])
    assert np.allclose(a[..., :3], b[..., :3], rtol=1e-5, atol=1e-5)   # float atomics accumulate in a different order


def test_native_framebuffer_reduce_single_rank(gpu):
    """ctl_comm_* / ctl_image_reduce (csrc/comm.cpp): RCCL is loaded, a communicator of one rank is created and the PixelData frame goes through
    ncclReduce unchanged.  (Two ranks need two GPUs — RCCL refuses two ranks on one device; the N-rank flow of bench.py is exercised on one GPU
    with CTL_BENCH_SHARE_GPU=1, where the reduce falls back to gloo.)"""
    img = gpu.Image(96, 64)
    a = np.random.RandomState(2).uniform(0, 4, size=(64, 96, 7)).astype(np.float32)
    img.setPixelData(a)
    ident = gpu.Comm.unique_id()
    assert len(ident) == 128
    comm = gpu.Comm(ident, 0, 1)
    comm.reduce(img, 0)
    assert np.array_equal(img.getPixelData(), a)
    with pytest.raises(gpu.CtlError):
        comm.reduce(img, 3)
    comm.reduce(img, 0)     # one rank: the sum is the frame itself, a repeat is harmless (with more ranks a second in-place reduce is refused on every rank, comm.cpp)
    assert np.array_equal(img.getPixelData(), a)
    del comm


def test_native_framebuffer_gather(gpu):
    """ctl_image_gather / _gather_to / ctl_image_pack_tiles / _unpack_tiles (csrc/comm.cpp), north_star's "single RCCL gather of the framebuffer": (1) a communicator of
    one rank: pack -> ncclGather -> unpack gives the frame back bit for bit, in place and out of place, at a size with clipped border tiles; (2) the device pack / unpack
    kernels against their numpy statement (tests/tile_shards.py) for every rank of 2-, 3- and 8-rank shards of frames in which samples were accumulated one pixel inside
    ANOTHER rank's tile (the halo of a slot; corner pixels that lie in the halo of several of a rank's tiles travel once); (3) the root's unpack of all ranks' buffers ->
    the one-rank frame, bit for bit (integer-valued data).  (ncclGather with world > 1 needs > 1 GPU.)"""
    import tile_shards
    from test_distributed_cpu import _spilled_frames
    w, h = 200, 150      # 4 x 3 tiles, right and bottom tiles clipped
    a = np.random.RandomState(3).uniform(-1, 4, size=(h, w, 7)).astype(np.float32)
    img, dst = gpu.Image(w, h), gpu.Image(w, h)
    img.setPixelData(a)
    comm = gpu.Comm(gpu.Comm.unique_id(), 0, 1, timeout_ms=60000)
    comm.gather(img, 0); comm.gather(img, 0)                # one rank: nothing to count twice
    assert np.array_equal(img.getPixelData(), a)
    comm.gather_to(img, dst, 0); comm.gather_to(img, dst, 0)
    assert np.array_equal(dst.getPixelData(), a) and np.array_equal(img.getPixelData(), a)
    comm.reduce(img, 0)                                     # the fallback gives the same frame
    assert np.array_equal(img.getPixelData(), a)
    with pytest.raises(gpu.CtlError):
        comm.gather(img, 2)
    with pytest.raises(gpu.CtlError):
        comm.gather_to(img, img, 0)
    with pytest.raises(gpu.CtlError):
        comm.gather_to(img, gpu.Image(64, 64), 0)
    del comm
    for world in (2, 3, 8):
        full, frames = _spilled_frames(w, h, world, seed=world)
        assert img.packedTileBytes(world) == tile_shards.packed_slots(w, h, world) * 65 * 65 * 28
        packed = []
        for r in range(world):
            img.setPixelData(frames[r])
            p = img.packTiles(r, world)
            assert np.array_equal(p, tile_shards.pack_tiles(frames[r], r, world)), (world, r)
            packed.append(p)
        out = gpu.Image(w, h); out.setPixelData(frames[0])                       # the root unpacks into its own frame (the in-place form) ...
        out.unpackTiles(world, np.stack(packed))
        assert np.array_equal(out.getPixelData(), full), world
        out.setPixelData(np.full_like(full, -7)); out.unpackTiles(world, np.stack(packed))   # ... or into any other image
        assert np.array_equal(out.getPixelData(), full)
    with pytest.raises(gpu.CtlError):
        img.packTiles(3, 3)
    with pytest.raises(ValueError):
        out.unpackTiles(3, np.zeros(5, np.float32))


def test_per_pass_gather_out_of_place(gpu, orc):
    """ctl_image_reduce_to (the per-pass gather of a progressive display; the reference shows the frame after every DoPass, main.cpp:164-172): after each of K passes the
    rank's cumulative frame is gathered into a display image; the source stays untouched, the last gather equals one end-of-render in-place reduce bit for bit.
    (One rank here — RCCL refuses two ranks on one device; tests/test_distributed_cpu.py runs the same contract with two ranks over gloo.)"""
    sc = scenes.cornell_box(64, 64, glass_sphere=True)
    scene = gpu.Scene(sc.desc, flatten=True)
    tr = gpu.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 4)
    tr.Resize(64, 64); tr.InitializeScene(scene)
    img, disp = gpu.Image(64, 64), gpu.Image(64, 64)
    comm = gpu.Comm(gpu.Comm.unique_id(), 0, 1, timeout_ms=60000)
    tables = orc.sequence_tables(3)
    for k in range(3):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
        before = img.getPixelData()
        comm.reduce_to(img, disp, 0)
        assert np.array_equal(img.getPixelData(), before) and np.array_equal(disp.getPixelData(), before)
        assert before[..., 6].sum() == (k + 1) * 64 * 64
    final = disp.getPixelData()
    comm.reduce(img, 0)
    assert np.array_equal(img.getPixelData(), final)
    with pytest.raises(gpu.CtlError):
        comm.reduce_to(img, img, 0)             # source and destination must differ
    small = gpu.Image(32, 32)
    with pytest.raises(gpu.CtlError):
        comm.reduce_to(img, small, 0)
    del comm


def test_tracer_parameter_kinds(gpu):
    """TracerParameterCollection (Kernel/TracerSettings.h:14-350): bool / int intervals, and enumerations addressed by value name or index"""
    tr = gpu.WavefrontPathTracer(); p = tr.getParameters()
    p.setValue("BlockSamplerType", "Variance"); assert p.getValue("BlockSamplerType") == 1
    p.setValue("BlockSamplerType", 3); assert p.getValue("BlockSamplerType") == 3
    p.setValue("BlockSamplerType", "Uniform"); assert p.getValue("BlockSamplerType") == 0
    for bad in (("BlockSamplerType", "NoSuchSampler"), ("BlockSamplerType", 7), ("MaxPathLength", "Uniform"), ("MaxPathLength", 0), ("MaxPathLength", 1.5), ("NoSuchKey", 1)):
        with pytest.raises(gpu.CtlError):
            p.setValue(*bad)
    with pytest.raises(gpu.CtlError):
        p.getFloat("MaxPathLength")


@pytest.mark.parametrize("filter_mode", ["anisotropic", "trilinear"])
def test_first_hit_ray_differentials_and_filtered_textures(gpu, orc, filter_mode):
    """PathTrace computes uv partials at depth 1 (Integrators/PathTracer.cu:60-61, DifferentialGeometry::computePartials) and ImageTexture::Evaluate(dg)
    then filters through the mip pyramid — EWA for the default TEXTURE_Anisotropic, two bilinear taps for TEXTURE_Trilinear (Texture.cu:15-29,
    MIPMap.cu:193-278); the wavefront tracer never computes partials and reads level 0.  A noise-textured ground plane seen at a grazing angle:
    the PathTracer plugin equals the oracle with partials, the wavefront plugin equals the oracle without, and the two differ on the textured first hits."""
    from cudatracerlib_amd import api
    w, h = 96, 64
    sc = api.DynamicScene()
    rs = np.random.RandomState(4)
    tex = rs.uniform(0.05, 0.95, size=(64, 64, 3)).astype(np.float32)
    tex[::2, ::2] *= 0.2                                                  # high-frequency content: minification changes the mean seen through a pixel
    fm = api.FILTER_ANISOTROPIC if filter_mode == "anisotropic" else api.FILTER_TRILINEAR
    img = sc.add_image(api.float3_to_rgbcol(tex), api.TEXEL_RGBCOL, api.WRAP_REPEAT, fm)
    P = np.array([[-40, 0, -40], [-40, 0, 40], [40, 0, 40], [40, 0, -40]], np.float32)
    uv = np.array([[0, 0], [0, 12], [12, 12], [12, 0]], np.float32)
    ground = api.diffuse((1, 1, 1)); ground.tex[0] = api.image_texture(img)
    sc.CreateNode(sc.add_mesh(P, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), normals=np.tile(np.array([0, 1, 0], np.float32), (4, 1)), uvs=uv, materials=[ground]))
    L = np.array([[-6, 12, -6], [6, 12, -6], [6, 12, 6], [-6, 12, 6]], np.float32)
    ln = sc.CreateNode(sc.add_mesh(L, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), normals=np.tile(np.array([0, -1, 0], np.float32), (4, 1)), materials=[api.diffuse((0.5, 0.5, 0.5))]))
    sc.CreateLight(ln, 0, (30.0, 30.0, 30.0))
    sc.setCamera((0, 1.5, -30), (0, 0.5, 0), (0, 1, 0), 50.0, w, h)
    d = sc.UpdateScene()
    tables = orc.sequence_tables(2)
    want_plain, _ = orc.render(d, w, h, n_passes=2, tables=tables, max_path_length=3)
    want_filtered, _ = orc.render(d, w, h, n_passes=2, tables=tables, max_path_length=3, partials=True)
    scene = gpu.Scene(d, flatten=True)
    mega = _render(gpu, gpu.PathTracer, scene, tables, w, h, max_len=3)
    wave = _render(gpu, gpu.WavefrontPathTracer, scene, tables, w, h, max_len=3)

    def frac_close(a, b):
        return (np.abs(a[..., :3] - b[..., :3]) <= 2e-3 * (1 + np.abs(b[..., :3]))).all(axis=2).mean()
    assert np.array_equal(mega[..., 6], want_filtered[..., 6]) and np.array_equal(wave[..., 6], want_plain[..., 6])
    assert frac_close(wave, want_plain) >= 0.995
    assert frac_close(mega, want_filtered) >= 0.99, frac_close(mega, want_filtered)
    assert abs(mega[..., :3].mean() - want_filtered[..., :3].mean()) <= 2e-3 * want_filtered[..., :3].mean()
    assert frac_close(mega, want_plain) < 0.9 and frac_close(want_filtered, want_plain) < 0.9          # the filtering is visible: the distant ground differs


def test_sequence_tables_generated_on_the_gpu_equal_the_host_generator(gpu):
    """k_sequence_fill (256 lanes per pass, two GF(2) jumps + 1440 draws each) against the host generator on the same XORWOW stream
    (SamplingSequenceGeneratorHost<IndependantSamplingSequenceGenerator>, Kernel/Sampler.h:36-85): bit for bit, and the stream goes on correctly"""
    host, dev = gpu.SequenceGenerator(), gpu.SequenceGenerator()
    for n in (1, 3, 5):
        h1, h2 = host.compute_many(n, threads=4)
        d1, d2 = dev.compute_many_device(n)
        assert np.array_equal(h1.view(np.uint32), d1.view(np.uint32))
        assert np.array_equal(h2.view(np.uint32), d2.view(np.uint32))
    a1, a2 = host.compute(); b1, b2 = dev.compute()               # both generators stand at the same place of the stream afterwards
    assert np.array_equal(a1, b1) and np.array_equal(a2, b2)


@pytest.mark.parametrize("direct", [True, False])
def test_path_tracer_regularization(gpu, orc, direct):
    """PathTracer plugin with Regularization = true (PathTraceRegularization, Integrators/PathTracer.cu:115-173): emitters seen only at depth 1 / after delta
    bounces, UniformSampleAllLights on BSDFs without delta lobes, the mollified connection to point / spot / distant emitters on BSDFs with them (cone
    shrinking with the pass number), roulette from RRStartDepth regardless of the bounce type, and the environment term exactly as the reference writes it"""
    sc = scenes.env_scene(96, 64, extra_lights=True)
    tables = orc.sequence_tables(3)
    want, want_rays = orc.render(sc.desc, 96, 64, n_passes=3, tables=tables, direct=direct, max_path_length=6, rr_start=3, regularization=True)
    tr = gpu.PathTracer(); p = tr.getParameters()
    p.setValue("Regularization", True); p.setValue("Direct", direct); p.setValue("MaxPathLength", 6); p.setValue("RRStartDepth", 3)
    tr.Resize(96, 64); tr.InitializeScene(gpu.Scene(sc.desc, flatten=True))
    img = gpu.Image(96, 64)
    for k in range(3):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData()
    assert_close(got, want)
    assert abs(int(tr.stats().rays_total) - int(want_rays)) <= 2e-3 * want_rays
    plain, _ = orc.render(sc.desc, 96, 64, n_passes=3, tables=tables, direct=direct, max_path_length=6, rr_start=3, partials=True)
    assert np.abs(plain[..., :3] - want[..., :3]).mean() > 1e-3       # it really is a different estimator


@pytest.mark.parametrize("kind", ["thinlens", "orthographic", "telecentric", "spherical"])
def test_thinlens_orthographic_and_telecentric_sensors(gpu, orc, kind):
    """ray generation of the other projective sensors (SceneTypes/Sensor.cu; the oracle's restatement is pinned on the reference's own code, tests/golden/sensors.npz):
    the aperture sample is used, orthographic cameras differentiate the ray ORIGIN (computePartials with the differential rays' own origins) — wavefront plugin
    and, with an image texture under first-hit filtering, the megakernel plugin"""
    sc = scenes.area_lights_scene(96, 64, "image")
    s = gpu.api.ctl_sensor.from_buffer_copy(sc.desc.camera)
    s.type = {"thinlens": 3, "orthographic": 4, "telecentric": 5, "spherical": 1}[kind]
    s.aperture_radius, s.focus_distance = (0.25, 9.0) if kind == "thinlens" else (0.05, 6.0)
    s.screen_scale[:] = [2.0, 2.0]
    if kind in ("orthographic", "telecentric"):                       # an orthographic view covers [-1, 1] camera units: look at the boxes from close by
        s.near_depth, s.far_depth = 1e-5, 1e5
    sc.setSensor(s); sc.UpdateScene()
    assert sc.desc.camera.type == s.type
    got, want, tr, rays = render_pair(gpu, orc, sc, 96, 64, 3)
    assert_close(got, want)
    assert want[..., :3].mean() > 0.01
    base = scenes.area_lights_scene(96, 64, "image")
    ref_img, _ = orc.render(base.desc, 96, 64, n_passes=3, tables=orc.sequence_tables(3))
    assert np.abs(ref_img[..., :3] - want[..., :3]).mean() > 1e-2       # a different camera, not the perspective one
    tables = orc.sequence_tables(3)
    want_f, _ = orc.render(sc.desc, 96, 64, n_passes=3, tables=tables, partials=True)
    mk = gpu.PathTracer(); mk.Resize(96, 64); mk.InitializeScene(gpu.Scene(sc.desc, flatten=True))
    img = gpu.Image(96, 64)
    for k in range(3):
        mk.setSamplerTables(*tables[k]); mk.DoPass(img, new_trace=(k == 0))
    assert_close(img.getPixelData(), want_f)


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_share_one_gpu(gpu, tmp_path):
    """`python bench.py --gpus 2` launches its two ranks itself (CTL_BENCH_SHARE_GPU=1: both on device 0, the 1-GPU box's stand-in for two devices): one JSON
    line with n_gpus = 2, all rays accounted for, and the reduced frame equals the frame of the in-process `--gpus 1` run (tiles partition the film,
    the sampler index is the global pixel; float atomics may add in another order)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "2", "--warmup", "1", "--width", "320", "--height", "192", "--instances", "60", "--subdiv", "2", "--no-cpu-baseline", "--no-cache"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    frames, lines = [], []
    for n in (1, 2):
        f = str(tmp_path / ("frame%d.npy" % n))
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--dump-frame", f] + common, env=dict(env, CTL_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        js = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1
        lines.append(json.loads(js[0])); frames.append(np.load(f))
    assert lines[0]["n_gpus"] == 1 and lines[1]["n_gpus"] == 2
    assert "framebuffer_reduce" in lines[1]["config"] and ("ncclReduce" in lines[1]["config"]["framebuffer_reduce"] or "native RCCL unavailable" in lines[1]["config"]["framebuffer_reduce"])
    assert lines[0]["config"]["rays_per_step"] == lines[1]["config"]["rays_per_step"]       # the shards' rays sum to the frame's
    a, b = frames
    assert np.array_equal(a[..., 6], b[..., 6])
    assert np.allclose(a[..., :6], b[..., :6], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("which", ["cornell_glass", "materials", "env", "sm"])
def test_wavefront_path_semantics(gpu, orc, which):
    """PathSemantics = Wavefront: the shade kernels built with pathIterateKernel's own rules (WavefrontPathTracer.cu:51-164) against the oracle's restatement of them
    (ocore.h pathTraceWavefront) on the same tables — Russian roulette before sampling, nothing sampled at the last bounce, sampleEmitterDirect with one 2-D sample,
    16-bit previous normal, the t >= dDist (1 - eps) shadow rule — with and without the 16-bit barycentrics; same ray count, and a different image than the default rules"""
    sc = {"cornell_glass": lambda: scenes.cornell_box(64, 64, glass_sphere=True), "materials": lambda: scenes.cornell_box(64, 64, extra_materials=2),
          "env": lambda: scenes.env_scene(96, 64, extra_lights=True), "sm": lambda: scenes.synthetic_sm(96, 64, n_instances=60, subdiv=2)}[which]()
    w, h = (96, 64) if which in ("env", "sm") else (64, 64)
    d = sc.desc
    tables = orc.sequence_tables(3)
    scene = gpu.Scene(d, flatten=(which == "sm"))
    for u16 in (False, True):
        want, want_rays = orc.render(d, w, h, n_passes=3, tables=tables, max_path_length=6, rr_start=3, wavefront_rules=True, u16_barycentrics=u16)
        tr = gpu.WavefrontPathTracer(); p = tr.getParameters()
        p.setValue("MaxPathLength", 6); p.setValue("RRStartDepth", 3); p.setValue("PathSemantics", "Wavefront"); p.setValue("U16Barycentrics", u16)
        tr.Resize(w, h); tr.InitializeScene(scene)
        img = gpu.Image(w, h); rays = 0
        for k in range(3):
            tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0)); rays += tr.stats().rays_last_pass
        got = img.getPixelData()
        assert_close(got, want)
        assert abs(rays - want_rays) <= 1e-2 * want_rays      # (a flipped discrete decision changes a path's length; a path whose throughput became exactly 0 is dropped at once, DESIGN.md §4 deviation 3)
    default, _ = orc.render(d, w, h, n_passes=3, tables=tables, max_path_length=6, rr_start=3)
    assert abs(default[..., :3].mean() - want[..., :3].mean()) > 2e-3 * want[..., :3].mean()


def test_debug_pixel_trace_single_ray_and_depth_buffer(gpu, orc):
    """The three boundary entries of SURVEY §8(b) beyond DoPass: TracerBase::Debug (Kernel/Tracer.h:119-123 -> PathTracer::DebugInternal, PathTracer.cu:172-180),
    TracerBase::TraceSingleRay (Kernel/Tracer.cu:74-78) and IDepthTracer::setDepthBuffer (Kernel/Tracer.h:16-57, WavefrontPathTracer.cu:76-77)."""
    import ctypes as C
    w, h = 48, 40
    sc = scenes.cornell_box(w, h, glass_sphere=True); d = sc.desc
    scene = gpu.Scene(d, flatten=True)
    fb = gpu.api.FlatBvh(d, gpu.api.FLAT_Q4)
    orc.lib.orc_debug_pixel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    tables = orc.sequence_tables(3)      # the sets a fresh tracer's generator produces, in order
    # ---- Debug: the path of one pixel with the NEXT table set; the pass after it uses the set after that
    tr = gpu.PathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 5)
    tr.Resize(w, h); tr.InitializeScene(scene)
    img = gpu.Image(w, h)
    tr.DoPass(img, new_trace=True)                              # set 1
    for (x, y) in ((10, 12), (24, 20), (40, 35)):
        tr2 = gpu.PathTracer(); tr2.getParameters().setValue("MaxPathLength", 5); tr2.Resize(w, h); tr2.InitializeScene(scene)
        got = tr2.Debug(img, x, y)                              # a fresh tracer: set 1
        want = np.zeros(3, np.float32); dist = C.c_float()
        orc.lib.orc_debug_pixel(C.byref(d), w, h, tables[0][0].ctypes.data, tables[0][1].ctypes.data, x, y, 5, 5, want.ctypes.data, C.byref(dist))
        assert np.allclose(got, want, rtol=2e-3, atol=2e-3), (x, y, got, want)
    before = img.getPixelData().copy()
    tr.Debug(img, 5, 5)                                         # set 2 is drawn and spent; the frame is untouched
    assert np.array_equal(img.getPixelData(), before)
    tr.DoPass(img)                                              # set 3
    want, _ = orc.render(d, w, h, n_passes=2, tables=[tables[0], tables[2]], max_path_length=5, flat=fb.desc, partials=True)
    assert_close(img.getPixelData(), want)
    # the wavefront tracer has no DebugInternal: Debug() only draws the table set
    wt = gpu.WavefrontPathTracer(); wt.getParameters().setValue("MaxPathLength", 5); wt.Resize(w, h); wt.InitializeScene(scene)
    wimg = gpu.Image(w, h)
    assert not wt.Debug(wimg, 3, 3).any()
    wt.DoPass(wimg, new_trace=True)                             # set 2
    want, _ = orc.render(d, w, h, n_passes=1, tables=[tables[1]], max_path_length=5)
    assert_close(wimg.getPixelData(), want)
    with pytest.raises(gpu.CtlError):
        tr.Debug(img, w, 0)
    # ---- TraceSingleRay == the oracle's traceRay == ctl_intersect
    rs = np.random.RandomState(5)
    for _ in range(20):
        o = rs.uniform(-0.9, 0.9, 3).astype(np.float32); dd = rs.normal(size=3); dd = (dd / np.linalg.norm(dd)).astype(np.float32)
        hit = gpu.api.trace_single_ray(scene, o, dd, tmin=d.ray_trace_eps)
        rays = np.zeros((1, 8), np.float32); rays[0, :3] = o; rays[0, 3] = d.ray_trace_eps; rays[0, 4:7] = dd; rays[0, 7] = 3.402823466e+38
        ref = orc.intersect(d, rays)[0]
        assert hit["tri_idx"] == ref["tri_idx"] and hit["node_idx"] == ref["node_idx"] and hit["dist"] == ref["dist"] and hit["u"] == ref["u"] and hit["v"] == ref["v"]
    # ---- depth buffer: NormalizeDepthD3D of the primary hit distance of every pixel
    wt2 = gpu.WavefrontPathTracer(); wt2.getParameters().setValue("MaxPathLength", 2); wt2.Resize(w, h); wt2.InitializeScene(scene)
    wt2.setDepthBuffer(w, h)
    wt2.setSamplerTables(*tables[0]); wt2.DoPass(wimg, new_trace=True)
    depth = wt2.getDepthBuffer()
    near, far = d.camera.near_depth, d.camera.far_depth
    assert ((depth >= 0) & (depth <= 1)).all() and depth.std() > 0
    for (x, y) in ((10, 12), (24, 20), (40, 35), (0, 0), (47, 39)):
        # the pass's primary ray of pixel (x, y): jittered by the first two table values — re-trace it through the oracle's sensor
        t1, t2 = tables[0]
        smp = orc.lib.orc_sampler_float      # (not needed: compare through the hit distance of the un-jittered ray within the pixel's depth range)
        dist = C.c_float(); orc.lib.orc_debug_pixel(C.byref(d), w, h, t1.ctypes.data, t2.ctypes.data, x, y, 1, 5, None, C.byref(dist))
        z = min(max(dist.value, near), far); want_z = (far / (far - near) * z - far * near / (far - near)) / z
        assert abs(depth[y, x] - want_z) < 0.05, (x, y, depth[y, x], want_z)      # (the pass's ray is jittered inside the pixel: same surface, nearly the same depth)
    with pytest.raises(gpu.CtlError):
        tr.setDepthBuffer(w, h)                                  # the megakernel PathTracer is not an IDepthTracer


def test_ordered_accumulation_is_independent_of_the_batching(gpu, orc):
    """OrderedAccumulation (default): a finished path stores its sample per (pass of the batch, pixel) and the batch is added to the frame pass by pass — in the order in which
    Image::AddSample is called by the reference's one-pass-at-a-time loop — instead of four float atomics per path in whatever order the hardware serves them.  So the frame does
    not depend on how the passes are batched (12 passes as 12 x 1, 3 x 4 and 1 x 12: bit-identical) nor on the run, and it equals the oracle's pass-by-pass sum to the bit wherever
    the samples themselves do; with the atomics (OrderedAccumulation = false) the same samples arrive, their sum is equal only to round-off."""
    w, h, n = 64, 64, 12
    sc = scenes.cornell_box(w, h, glass_sphere=True)
    scene = gpu.Scene(sc.desc, flatten=True)

    def render(batch, ordered=True):
        tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 5); p.setValue("PassBatch", batch); p.setValue("OrderedAccumulation", ordered)
        tr.Resize(w, h); tr.InitializeScene(scene)
        img = gpu.Image(w, h)
        tr.DoPasses(img, n, new_trace=True)     # the tracer's own sampling-sequence generator: the same stream for every tracer
        return img.getPixelData(), tr.stats().rays_total

    base, rays = render(1)
    assert (base[..., 6] == n).all()
    for batch in (4, 12, 1):
        got, r = render(batch)
        assert r == rays and np.array_equal(got.view(np.uint32), base.view(np.uint32)), batch
    atom, r = render(12, ordered=False)
    assert r == rays and np.array_equal(atom[..., 6], base[..., 6])
    assert np.allclose(atom[..., :3], base[..., :3], rtol=1e-5, atol=1e-6)
    # the oracle adds pass after pass: with one pass per launch and the oracle's tables the frames agree to the bit in (almost) every pixel
    tables = orc.sequence_tables(3)
    want, _ = orc.render(sc.desc, w, h, n_passes=3, tables=tables, max_path_length=5, flat=None)
    tr = gpu.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 5)
    tr.Resize(w, h); tr.InitializeScene(scene)
    img = gpu.Image(w, h)
    for k in range(3):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData()
    assert (got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)).all(axis=2).mean() > 0.98


def test_image_add_samples_equals_the_references_own_add_sample(gpu):
    """ctl_image_add_samples = Image::AddSample (Engine/Image.cu:22-44) on the device — compaction.h add_sample, the function the shade kernels deposit finished paths with — over
    the samples of tests/golden/image.npz, whose expected frame was made by the reference's own code: which samples are dropped (outside the film after floor, NaN / infinite
    radiance after clampNegative) and where the others land is equal one for one (weightSum exact); the sums are float atomics in hardware order, equal to round-off.
    NaN / infinite POSITIONS are left out: (int)floorf of those is the host's conversion in the fixture and the GPU's here (the reference's own host and CUDA branches differ
    the same way); a film position is pixel + jitter, never one of them."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image.npz"))
    W, H, s = int(g["width"]), int(g["height"]), g["samples"]
    s = s[np.isfinite(s[:, :2]).all(axis=1) & (np.abs(s[:, :2]) < 1e9).all(axis=1)]
    assert len(s) > 5000
    # the fixture's frame without the left-out samples: they were all dropped by the reference (NaN / infinite / beyond-int positions land outside the film)
    want = g["pixels"]
    img = gpu.Image(W, H)
    img.addSamples(s)
    got = img.getPixelData()
    assert np.array_equal(got[..., 6], want[..., 6])
    assert np.allclose(got[..., :3], want[..., :3], rtol=2e-6, atol=0) and (got[..., 3:6] == 0).all()
    img.addSamples(s[:100])
    assert got[..., 6].sum() < img.getPixelData()[..., 6].sum() <= got[..., 6].sum() + 100


def test_refused_batch_leaves_the_tracer_usable(gpu):
    """A batch beyond the 2^31 ray slots of a wavefront is refused BEFORE anything changes (advisor r4: the check used to run after Resize had stored the new size and batch, and every
    later Resize threw again): 4096 x 4096 pixels x 128 passes = 2^31 slots is refused by DoPasses; the same tracer then renders with a smaller batch, and resizes."""
    sc = scenes.cornell_box(64, 64)
    scene = gpu.Scene(sc.desc, flatten=True)
    tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 2); p.setValue("PassBatch", 128)
    tr.Resize(4096, 4096); tr.InitializeScene(scene)
    img = gpu.Image(4096, 4096)
    with pytest.raises(gpu.CtlError, match="2\\^31 ray slots"):
        tr.DoPasses(img, 128, new_trace=True)
    with pytest.raises(gpu.CtlError, match="2\\^31 ray slots"):
        tr.reservePasses(128)
    p.setValue("PassBatch", 2)
    tr.DoPasses(img, 2, new_trace=True)
    assert tr.stats().passes_done == 2 and tr.stats().rays_total > 2 * 4096 * 4096
    tr.Resize(64, 64)
    small = gpu.Image(64, 64); tr.DoPasses(small, 2, new_trace=True)
    assert small.getPixelData()[..., 6].sum() == 2 * 64 * 64


def test_material_index_out_of_range_is_refused(gpu):
    """a triangle that names a material the scene does not have: ctl_scene_create refuses the flattened scene (advisor r4: its traversal key would have been the MISS key and the vertex
    would have been shaded from a record outside the material array)"""
    import ctypes as C
    sc = scenes.cornell_box(64, 64)
    d = sc.desc
    n = d.n_tri_data
    copy = np.frombuffer(C.string_at(d.tri_data, n * 32), np.uint32).copy().reshape(n, 8)
    copy[3, 1] = (int(copy[3, 1]) & 0xff00ffff) | (200 << 16)                              # TriangleData::getMatIndex: bits 16..23 of the second word
    old = d.tri_data
    d.tri_data = copy.ctypes.data
    try:
        with pytest.raises(gpu.CtlError, match="names material"):
            gpu.Scene(d, flatten=True)
    finally:
        d.tri_data = old


@pytest.mark.gpu
def test_bench_rccl_branch_runs_with_one_rank(gpu, tmp_path):
    """bench.py's N-rank code path — gloo group, communicator with a deadline, the gather AND the reduce warmed up before the timed region, the agreed choice, the exchange inside
    the timed region, the per-rank fields — executed with ONE rank against the real librccl.so (CTL_BENCH_COMM_WORLD1=1), so that an 8-GPU node is not the first place where it
    runs: the native gather is chosen; with CTL_BENCH_NO_GATHER=1 the reduce on a fresh communicator is; both frames equal the plain one-rank frame bit for bit."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "192", "--instances", "60", "--subdiv", "2", "--no-cpu-baseline", "--no-cache"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CTL_BENCH_SHARE_GPU")}
    frames, lines = [], []
    for extra in ({}, {"CTL_BENCH_COMM_WORLD1": "1"}, {"CTL_BENCH_COMM_WORLD1": "1", "CTL_BENCH_NO_GATHER": "1"}):
        f = str(tmp_path / ("frame%d.npy" % len(frames)))
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dump-frame", f] + common, env=dict(env, **extra), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        js = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1
        lines.append(json.loads(js[0])); frames.append(np.load(f))
    assert "ncclGather" in lines[1]["config"]["framebuffer_reduce"] and "reduce_ms" in lines[1] and len(lines[1]["rank_ms"]) == 1
    assert "ncclReduce" in lines[2]["config"]["framebuffer_reduce"] and "CTL_BENCH_NO_GATHER" in lines[2]["config"]["framebuffer_reduce"]
    assert lines[0]["n_gpus"] == lines[1]["n_gpus"] == 1 and "rank_ms" not in lines[0]
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)) and np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32))
