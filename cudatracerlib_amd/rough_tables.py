"""Rough-transmittance tables for roughplastic when Mitsuba's data/microfacet/*.dat files are not at hand.

The reference loads three precomputed tables (Engine/RoughTransmittance.cu:8-45,124-131) that ship with Mitsuba, not with
CudaTracerLib.  ``DynamicScene.loadRoughTransmittance`` reads those files when they exist; this module computes a stand-in
with the same layout and parameterisation by quadrature over the microfacet reflection lobe:

    T(mu, alpha, eta) = 1 - integral F(wi.h) D(h) G(wi, wo, h) / (4 mu) d wo          (transmittance = 1 - reflectance)
    Tdiff(alpha, eta) = 2 integral_0^1 T(mu) mu d mu

Grid (as Mitsuba's rough-transmittance precomputation): mu = t^4, alpha = alphaMin + (alphaMax - alphaMin) t^4,
eta = etaMin + (etaMax - etaMin) t^4 with t uniform on [0, 1]; block 0 holds eta >= 1, block 1 the inverse direction.
It is DATA for tests and synthetic benchmarks; renders with it match the reference only when the reference is given the same
table.
"""
import numpy as np

ETA_RANGE = (1.0001, 4.0)
ALPHA_RANGE = (0.0001, 4.0)


def _fresnel(c, eta):
    c = np.abs(c)
    sin2t = (1 - c * c) / (eta * eta)
    ct = np.sqrt(np.maximum(0.0, 1 - sin2t))
    with np.errstate(divide="ignore", invalid="ignore"):
        rs = (c - eta * ct) / (c + eta * ct); rp = (eta * c - ct) / (eta * c + ct)
    return np.where(sin2t >= 1, 1.0, 0.5 * (rs * rs + rp * rp))


def _g1(v_z, v_dot_m, alpha, ggx):
    ok = (v_dot_m * v_z) > 0
    tan2 = np.maximum(0.0, 1 - v_z * v_z) / np.maximum(v_z * v_z, 1e-12)
    if ggx:
        g = 2.0 / (1.0 + np.sqrt(1.0 + alpha * alpha * tan2))
    else:
        a = 1.0 / np.maximum(alpha * np.sqrt(tan2), 1e-12)
        g = np.where(a >= 1.6, 1.0, (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a))
    return np.where(ok, g, 0.0)


def _reflectance(mu, alpha, eta, ggx, n=48):
    """hemispherical reflectance of the rough interface by sampling h ~ D(h) cos(h) on an n x n stratified grid"""
    u1, u2 = np.meshgrid((np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n, indexing="ij")
    phi = 2 * np.pi * u2
    tan2 = alpha * alpha * u1 / (1 - u1) if ggx else -alpha * alpha * np.log(1 - u1)
    cos_h = 1 / np.sqrt(1 + tan2); sin_h = np.sqrt(np.maximum(0.0, 1 - cos_h * cos_h))
    h = np.stack([sin_h * np.cos(phi), sin_h * np.sin(phi), cos_h], -1)
    wi = np.array([np.sqrt(max(0.0, 1 - mu * mu)), 0.0, mu])
    wih = h @ wi
    wo = 2 * wih[..., None] * h - wi
    ok = (wih > 0) & (wo[..., 2] > 0)
    g = _g1(wi[2], wih, alpha, ggx) * _g1(wo[..., 2], (wo * h).sum(-1), alpha, ggx)
    w = _fresnel(wih, eta) * g * wih / np.maximum(mu * cos_h, 1e-12)
    return float(np.clip(np.where(ok, w, 0.0).mean(), 0.0, 1.0))


def make_table(distribution, n_eta=5, n_alpha=6, n_theta=10, quad=32):
    """returns (trans (2*n_eta, n_alpha, n_theta), diff (2*n_eta, n_alpha), ETA_RANGE, ALPHA_RANGE); distribution 0 = Beckmann, else GGX"""
    ggx = distribution != 0
    t_e, t_a, t_m = (np.linspace(0, 1, n) ** 4 for n in (n_eta, n_alpha, n_theta))
    etas = ETA_RANGE[0] + (ETA_RANGE[1] - ETA_RANGE[0]) * t_e
    alphas = ALPHA_RANGE[0] + (ALPHA_RANGE[1] - ALPHA_RANGE[0]) * t_a
    trans = np.zeros((2 * n_eta, n_alpha, n_theta), np.float32); diff = np.zeros((2 * n_eta, n_alpha), np.float32)
    mu_q = (np.arange(16) + 0.5) / 16
    for blk in range(2):
        for i, e in enumerate(etas):
            eta = e if blk == 0 else 1.0 / e
            for j, a in enumerate(alphas):
                for k, mu in enumerate(t_m):
                    trans[blk * n_eta + i, j, k] = 1.0 - _reflectance(max(float(mu), 1e-3), float(a), float(eta), ggx, quad)
                tq = np.array([1.0 - _reflectance(float(m), float(a), float(eta), ggx, quad // 2) for m in mu_q])
                diff[blk * n_eta + i, j] = float(np.clip(2 * (tq * mu_q).mean(), 0.0, 1.0))
    return trans, diff, ETA_RANGE, ALPHA_RANGE


def write_dat(path, trans, diff, eta_range, alpha_range):
    """the on-disk layout RoughTransmittance::RoughTransmittance reads (RoughTransmittance.cu:8-45)"""
    n_eta2, n_alpha, n_theta = trans.shape
    with open(path, "wb") as f:
        f.write(b"MTS_TRANSMITTANCE")
        f.write(np.array([n_eta2 // 2, n_alpha, n_theta], np.uint64).tobytes())
        f.write(np.array([eta_range[0], eta_range[1], alpha_range[0], alpha_range[1]], np.float32).tobytes())
        for i in range(n_eta2):
            for j in range(n_alpha):
                f.write(trans[i, j].astype(np.float32).tobytes()); f.write(np.float32(diff[i, j]).tobytes())
