"""Self-consistency of the oracle's BSDF restatement (oracle/ocore.h, oracle/obsdf2.h <- SceneTypes/BSDF/BSDF_Simple.cu).

The models themselves are pinned bit for bit on the reference's own BSDF_Simple.cu / BSDF_Complex.cu in tests/test_oracle_golden.py (bsdf.npz) — all but
roughplastic / roughcoating, whose transmittance lookup (Math/Spline.cu) only nvcc compiles.  Here the three entry points of every model, those two included,
are held against each other the way Mitsuba's own chi-square / consistency tests do:
  * sample() returns weight = f(wi, wo) / pdf(wi, wo) and the same pdf that pdf() reports for the sampled direction,
  * delta lobes report f = pdf = 0 through eval,
  * no model creates energy (mean sample weight <= 1 for unit reflectance).
The pieces they are built from (Fresnel terms, warps, the microfacet distribution) are pinned against the reference
itself in test_oracle_golden.py.
"""
import ctypes as C
import numpy as np
import pytest
import oracle
from cudatracerlib_amd import api

EAll = 0x1FF
DELTA = 0x1 | 0x20 | 0x40


@pytest.fixture(scope="module")
def lib():
    o = oracle.load()
    # rough-transmittance tables for the roughplastic probes (synthetic stand-ins for Mitsuba's microfacet/*.dat)
    from cudatracerlib_amd import rough_tables
    keep = []
    tabs = (api.ctl_rough_transmittance * 3)()
    for slot in (0, 1):
        tr, df, er, ar = rough_tables.make_table(slot, n_eta=4, n_alpha=5, n_theta=8, quad=16)
        keep += [tr, df]
        t = tabs[slot]
        t.trans, t.diff_trans = tr.ctypes.data, df.ctypes.data
        t.eta_samples, t.alpha_samples, t.theta_samples = tr.shape[0] // 2, tr.shape[1], tr.shape[2]
        t.eta_min, t.eta_max, t.alpha_min, t.alpha_max = er[0], er[1], ar[0], ar[1]
    o.orc_set_probe_rough_transmittance(C.addressof(tabs))
    o._keep = (keep, tabs)
    return o


def _sample(lib, m, wi, s):
    out = np.zeros(9, np.float32)
    wi = np.asarray(wi, np.float32)
    lib.orc_bsdf_sample(C.addressof(m), wi.ctypes.data, float(s[0]), float(s[1]), out.ctypes.data)
    return out


def _eval(lib, m, wi, wo, mask=EAll):
    out = np.zeros(4, np.float32)
    wi = np.asarray(wi, np.float32); wo = np.asarray(wo, np.float32)
    lib.orc_bsdf_eval(C.addressof(m), wi.ctypes.data, wo.ctypes.data, mask, out.ctypes.data)
    return out


def _wi(theta, phi=0.3):
    return np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], np.float32)


MODELS = {
    "diffuse": lambda: api.diffuse((1, 1, 1)),
    "roughconductor_ggx_vis": lambda: api.roughconductor(alpha=0.2, distribution=1, sample_visible=True),
    "roughconductor_beck": lambda: api.roughconductor(alpha=0.3, distribution=0, sample_visible=False),
    "roughconductor_beck_vis_aniso": lambda: api.roughconductor(alpha=0.25, alpha_v=0.1, distribution=0, sample_visible=True),   # what <string name="distribution" value="beckmann"/> loads as
    "roughconductor_phong": lambda: api.roughconductor(alpha=0.2, alpha_v=0.35, distribution=2, sample_visible=False),
    "roughdielectric_beck_vis": lambda: api.roughdielectric(alpha=0.2, int_ior=1.5, ext_ior=1.0, distribution=0, sample_visible=True),
    "roughdielectric_ggx_vis": lambda: api.roughdielectric(alpha=0.15, int_ior=1.5, ext_ior=1.0, distribution=1, sample_visible=True),
    "roughdielectric_beck": lambda: api.roughdielectric(alpha=0.3, int_ior=1.33, ext_ior=1.0, distribution=0, sample_visible=False),
    "roughdielectric_aniso": lambda: api.roughdielectric(alpha=0.3, alpha_v=0.1, int_ior=1.5, ext_ior=1.0, distribution=1, sample_visible=True),
    "plastic": lambda: api.plastic(diffuse_reflectance=(1, 1, 1), int_ior=1.49),
    "plastic_nonlinear": lambda: api.plastic(diffuse_reflectance=(0.5, 0.4, 0.3), int_ior=1.9, nonlinear=True),
    "phong": lambda: api.phong(diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(0.5, 0.5, 0.5), exponent=25.0),
    "roughdiffuse": lambda: api.roughdiffuse((1, 1, 1), alpha=0.5),
    "roughdiffuse_fast": lambda: api.roughdiffuse((0.8, 0.8, 0.8), alpha=0.3, use_fast_approx=True),
    "ward": lambda: api.ward((0.4, 0.4, 0.4), (0.5, 0.5, 0.5), 0.15, 0.15, variant=2),
    "ward_aniso_duer": lambda: api.ward((0.4, 0.4, 0.4), (0.3, 0.3, 0.3), 0.1, 0.3, variant=1),
    "roughplastic_beckmann": lambda: api.roughplastic((0.5, 0.5, 0.5), alpha=0.2, distribution=0),
    "roughplastic_ggx_nonlinear": lambda: api.roughplastic((0.6, 0.3, 0.2), alpha=0.35, int_ior=1.7, distribution=1, nonlinear=True),
    "thindielectric": lambda: api.thindielectric(int_ior=1.5, ext_ior=1.0),
    "dielectric": lambda: api.dielectric(int_ior=1.5, ext_ior=1.0),
}


@pytest.mark.parametrize("name", list(MODELS))
def test_sample_eval_pdf_are_consistent(lib, name):
    m = MODELS[name]()
    rs = np.random.RandomState(7)
    n_smooth = 0
    thetas = [0.1, 0.7, 1.3] + ([np.pi - 0.4, np.pi - 1.1] if "dielectric" in name else [])
    for theta in thetas:
        wi = _wi(theta)
        for s in rs.rand(200, 2):
            r = _sample(lib, m, wi, s)
            w, pdf, wo, typ = r[:3], r[3], r[4:7], int(r[7])
            if pdf == 0 or not np.any(w):
                continue
            assert np.all(np.isfinite(r))
            assert abs(np.linalg.norm(wo) - 1) < 1e-4
            if typ & DELTA:
                e = _eval(lib, m, wi, wo, typ)      # asking eval for a delta lobe alone: nothing
                assert e[3] == 0 and not np.any(e[:3])
                continue
            e = _eval(lib, m, wi, wo, EAll & ~DELTA if name.startswith("plastic") else EAll)
            if name.startswith("plastic"):
                # sample() chose between the delta coat and the diffuse base; the diffuse branch divides by (1 - probSpecular)
                continue
            n_smooth += 1
            assert e[3] == pytest.approx(pdf, rel=2e-3, abs=1e-6), (theta, s)
            assert e[:3] / e[3] == pytest.approx(w, rel=5e-3, abs=1e-5), (theta, s)
    if name not in ("thindielectric", "dielectric") and not name.startswith("plastic"):
        assert n_smooth > 100


@pytest.mark.parametrize("name", list(MODELS))
def test_no_energy_gain(lib, name):
    if name == "ward":
        # the reference's balanced variant divides by cos^4 of the NORMALISED half vector (BSDF_Simple.cu:1257-1258, marked
        # "POSSIBLE ERROR" there; Mitsuba uses the unnormalised one), which gains up to |wi+wo|^4 = 16x; restated as is
        pytest.skip("reference quirk: the balanced Ward variant is not energy conserving")
    m = MODELS[name]()
    rs = np.random.RandomState(11)
    for theta in (0.2, 1.0, 1.45):
        wi = _wi(theta)
        tot = np.zeros(3)
        S = rs.rand(4000, 2)
        for s in S:
            tot += _sample(lib, m, wi, s)[:3]
        assert np.all(tot / len(S) <= 1.02), (theta, tot / len(S))


def _nesting_models():
    """(name, material array, index of the parent): children are entries of the same array (absolute indices)"""
    mats = (api.ctl_material * 16)()
    children = [api.diffuse((0.8, 0.7, 0.6)), api.roughconductor(alpha=0.2), api.conductor(eta=(0.2, 0.9, 1.1), k=(3.9, 2.4, 2.1)),
                api.plastic((0.5, 0.5, 0.5), int_ior=1.49), api.dielectric(int_ior=1.5, ext_ior=1.0)]
    for i, c in enumerate(children):
        mats[i] = c
    parents = {
        "coating_diffuse": api.coating(0, children[0], int_ior=1.5, ext_ior=1.0, thickness=1.0, sigma_a=(0.1, 0.2, 0.4)),
        "coating_conductor": api.coating(2, children[2], int_ior=1.4, ext_ior=1.0),
        "coating_plastic": api.coating(3, children[3], int_ior=1.6, ext_ior=1.0, thickness=0.5, sigma_a=0.2),
        "roughcoating_diffuse": api.roughcoating(0, children[0], alpha=0.2, int_ior=1.5, ext_ior=1.0, distribution=0),
        "roughcoating_ggx_metal": api.roughcoating(1, children[1], alpha=0.3, int_ior=1.5, ext_ior=1.0, distribution=1, sigma_a=0.1),
        "blend_diffuse_metal": api.blend(0, children[0], 1, children[1], weight=0.3),
        "blend_mirror_diffuse": api.blend(2, children[2], 0, children[0], weight=0.6),
        "blend_glass_diffuse": api.blend(4, children[4], 0, children[0], weight=0.5),
    }
    out = {}
    for k, (name, p) in enumerate(parents.items()):
        mats[8 + k] = p
        out[name] = 8 + k
    return mats, out


@pytest.mark.parametrize("name", ["coating_diffuse", "coating_conductor", "coating_plastic", "roughcoating_diffuse", "roughcoating_ggx_metal",
                                  "blend_diffuse_metal", "blend_mirror_diffuse", "blend_glass_diffuse"])
def test_nesting_models_are_consistent(lib, name):
    """coating / roughcoating / blend: sample() == f / pdf in the measure of the sampled lobe (solid angle or discrete), no energy gain"""
    mats, index = _nesting_models()
    lib.orc_set_probe_materials(C.addressof(mats))
    m = mats[index[name]]
    rs = np.random.RandomState(13)
    n_checked = 0
    for theta in (0.15, 0.8, 1.25):
        wi = _wi(theta)
        tot = np.zeros(3); S = rs.rand(600, 2)
        for s in S:
            r = _sample(lib, m, wi, s)
            w, pdf, wo, typ = r[:3], r[3], r[4:7], int(r[7])
            tot += w
            if pdf == 0 or not np.any(w):
                continue
            assert np.all(np.isfinite(r)) and abs(np.linalg.norm(wo) - 1) < 1e-3
            out = np.zeros(4, np.float32)
            wi32, wo32 = np.asarray(wi, np.float32), np.asarray(wo, np.float32)
            if typ & DELTA:
                lib.orc_bsdf_eval_discrete(C.addressof(m), wi32.ctypes.data, wo32.ctypes.data, EAll, out.ctypes.data)
            else:
                lib.orc_bsdf_eval(C.addressof(m), wi32.ctypes.data, wo32.ctypes.data, EAll, out.ctypes.data)
            if name.startswith("coating") and (typ & DELTA):
                # a delta child reflects into the same direction as the coat itself; f()/pdf() with the discrete measure answer for
                # the coat's own lobe first (BSDF_Complex.cu:89-93,135-138), so the pair cannot be told apart afterwards
                continue
            if name.startswith("coating"):
                # coating::sample scales the nested sample instead of calling f()/pdf(): compare the ratio
                assert out[3] > 0
                assert out[:3] / out[3] == pytest.approx(w, rel=2e-2, abs=2e-4), (theta, s, typ)
            else:
                assert out[3] == pytest.approx(pdf, rel=5e-3, abs=1e-6), (theta, s, typ)
                assert out[:3] / out[3] == pytest.approx(w, rel=1e-2, abs=1e-4), (theta, s, typ)
            n_checked += 1
        assert np.all(tot / len(S) <= 1.03), (theta, tot / len(S))
    assert n_checked > 200 or name == "coating_conductor"   # (all of its lobes are delta)
    lib.orc_set_probe_materials(None)


def test_plastic_branches(lib):
    """plastic: P(specular) = Fi*w / (Fi*w + (1-Fi)(1-w)) (BSDF_Simple.cu plastic::sample); both branches carry 1/prob."""
    m = api.plastic(diffuse_reflectance=(0.5, 0.5, 0.5), int_ior=1.49)
    wi = _wi(0.9)
    rs = np.random.RandomState(3)
    spec = 0
    S = rs.rand(3000, 2)
    for s in S:
        r = _sample(lib, m, wi, s)
        if int(r[7]) & 0x20:
            spec += 1
            assert np.allclose(r[4:7], [-wi[0], -wi[1], wi[2]], atol=1e-6)
        else:
            e = _eval(lib, m, wi, r[4:7], 0x2)
            # pdf() with only the diffuse lobe requested reports the plain cosine pdf; sample() folded (1 - probSpecular) in
            assert e[3] > 0 and r[3] < e[3] * 1.0001
            assert e[:3] / r[3] == pytest.approx(r[:3], rel=5e-3)
    assert 0.0 < spec / len(S) < 0.5


def test_fresnel_diffuse_reflectance_matches_reference_values():
    """api.fresnel_diffuse_reflectance vs FresnelHelper::fresnelDiffuseReflectance(eta, false) run from the reference's own
    source (oracle/_ref, values recorded in tests/golden/generate.py -> math.npz)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "math.npz"))
    for eta, want in zip(z["fdr_eta"], z["fdr_value"]):
        assert api.fresnel_diffuse_reflectance(float(eta)) == pytest.approx(float(want), rel=2e-5)


def test_rough_transmittance_lookup_interpolates_its_table(lib):
    """RoughTransmittance::Evaluate / EvaluateDiffuse (RoughTransmittance.cu:55-119) cannot be pinned on the reference (Spline.cu does not build
    outside nvcc).  What the restatement is held to instead: the cubic interpolant reproduces the table at its knots — the knots sit at
    t^4 of a uniform grid in (cos theta, alpha, eta), the warp the lookup undoes — and stays within the hull of the neighbouring knots' values plus
    the Catmull-Rom overshoot bound between them."""
    keep, tabs = lib._keep
    for slot in (0, 1):
        tr = keep[2 * slot]; df = keep[2 * slot + 1]; t = tabs[slot]
        n_eta, n_alpha, n_theta = t.eta_samples, t.alpha_samples, t.theta_samples
        w = lambda n: (np.linspace(0, 1, n, dtype=np.float64) ** 4)
        etas = t.eta_min + (t.eta_max - t.eta_min) * w(n_eta); alphas = t.alpha_min + (t.alpha_max - t.alpha_min) * w(n_alpha); mus = w(n_theta)
        worst = 0.0
        for i in range(1, n_eta):
            for j in range(1, n_alpha):
                for k in range(1, n_theta):
                    got = lib.orc_rough_transmittance_eval(slot, float(mus[k]), float(alphas[j]), float(etas[i]))
                    worst = max(worst, abs(got - min(1.0, max(0.0, float(tr[i, j, k])))))
                    got_in = lib.orc_rough_transmittance_eval(slot, -float(mus[k]), float(alphas[j]), float(etas[i]))    # from inside: the 1 / eta block
                    worst = max(worst, abs(got_in - min(1.0, max(0.0, float(tr[n_eta + i, j, k])))))
                gd = lib.orc_rough_transmittance_eval_diffuse(slot, float(alphas[j]), float(etas[i]))
                worst = max(worst, abs(gd - min(1.0, max(0.0, float(df[i, j])))))
        assert worst <= 2e-4, worst                                   # the knot positions go through powf(x, 0.25) in fp32
        # between knots: bounded by the surrounding values (+ 12.5 % of their spread per axis for a Catmull-Rom segment)
        rs = np.random.RandomState(5 + slot)
        for _ in range(200):
            i, j, k = rs.randint(1, n_eta - 1), rs.randint(1, n_alpha - 1), rs.randint(1, n_theta - 1)
            f = rs.uniform(0, 1, 3)
            wm = ((k + f[0]) / (n_theta - 1)) ** 4; wa = ((j + f[1]) / (n_alpha - 1)) ** 4; we = ((i + f[2]) / (n_eta - 1)) ** 4
            got = lib.orc_rough_transmittance_eval(slot, float(wm), float(t.alpha_min + (t.alpha_max - t.alpha_min) * wa), float(t.eta_min + (t.eta_max - t.eta_min) * we))
            nb = tr[max(i - 1, 0):i + 3, max(j - 1, 0):j + 3, max(k - 1, 0):k + 3]
            spread = float(nb.max() - nb.min())
            assert nb.min() - 0.4 * spread - 1e-4 <= got <= min(1.0, nb.max() + 0.4 * spread) + 1e-4
