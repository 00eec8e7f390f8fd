"""Image-pipeline oracle (oracle/pipeline.py <- Kernel/ImagePipeline/*, Engine/Image.cu:88-168): closed forms."""
import numpy as np
from oracle import pipeline as P
from cudatracerlib_amd import api


def _frame(h=12, w=16, seed=1):
    rs = np.random.RandomState(seed)
    px = np.zeros((h, w, 7), np.float32)
    px[..., 6] = rs.randint(1, 5, (h, w))
    px[..., :3] = rs.rand(h, w, 3).astype(np.float32) * 2.0 * px[..., 6:7]
    px[..., 3:6] = rs.rand(h, w, 3).astype(np.float32) * 0.1
    return px


def test_rgbe_matches_the_texture_codec_and_round_trips():
    c = (np.random.RandomState(2).rand(64, 3).astype(np.float32) * np.float32(50.0)) ** 2
    c[0] = 0; c[1] = 1e-35
    enc = P.to_rgbe(c)
    assert np.array_equal(enc, api.float3_to_rgbe(c[None])[0])      # the same Float3ToRGBE the bitmap loader uses
    dec = P.from_rgbe(enc)
    assert np.all(dec[:2] == 0)
    m = c.max(axis=1)
    assert np.all(np.abs(dec[2:] - c[2:]) <= (m[2:] / 128)[:, None] + 1e-30)   # 8-bit mantissa under a shared exponent


def test_no_filter_no_process_is_gamma_of_to_spectrum():
    px = _frame()
    out = P.apply_image_pipeline(px, 0.5)
    lin = px[..., :3] / px[..., 6:7] + px[..., 3:6] * 0.5
    want = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1 / 2.4) - 0.055)
    assert np.abs(out[..., :3].astype(int) - np.floor(np.clip(want, 0, 1) * 255).astype(int)).max() <= 1
    assert np.all(out[..., 3] == 255)


def test_box_filter_is_the_window_mean_and_constant_images_stay_constant():
    px = _frame()
    flt = dict(type=1, xw=1.0, yw=2.0, p0=0, p1=0)
    got = P.from_rgbe(P.canonical_filter(px, 0.0, flt))
    spec = P.to_spectrum(px, 0.0)
    h, w = spec.shape[:2]
    for (y, x) in ((0, 0), (5, 7), (11, 15), (3, 0)):
        win = spec[max(0, y - 2):min(h, y + 3), max(0, x - 1):min(w, x + 2)].reshape(-1, 3)
        want = win.mean(axis=0)
        assert np.all(np.abs(got[y, x] - want) <= want.max() / 100)
    flat = np.zeros((8, 8, 7), np.float32); flat[..., :3] = (0.25, 0.5, 0.75); flat[..., 6] = 1
    for t, p0, p1 in ((1, 0, 0), (2, 2.0, 0), (3, 1 / 3, 1 / 3), (4, 3.0, 0), (5, 0, 0)):
        f = dict(type=t, xw=2.0, yw=2.0, p0=p0, p1=p1)
        out = P.from_rgbe(P.canonical_filter(flat, 0.0, f))
        assert np.allclose(out, (0.25, 0.5, 0.75), atol=0.75 / 100), t


def test_filter_shapes():
    g = dict(type=2, xw=2.0, yw=2.0, p0=-2.0, p1=0)     # the reference's default alpha is NEGATIVE: the "Gaussian" grows outwards and is clipped at 0
    assert P.filter_eval(g, 0, 0) == 0.0 and P.filter_eval(g, 2, 2) == 0.0
    g = dict(type=2, xw=2.0, yw=2.0, p0=2.0, p1=0)
    assert P.filter_eval(g, 0, 0) > P.filter_eval(g, 1, 0) > P.filter_eval(g, 1, 1) > 0 and P.filter_eval(g, 2, 0) == 0
    t = dict(type=5, xw=2.0, yw=2.0, p0=0, p1=0)
    assert P.filter_eval(t, 0, 0) == 4 and P.filter_eval(t, 1, 1) == 1 and P.filter_eval(t, 2, 0) == 0
    m = dict(type=3, xw=2.0, yw=2.0, p0=1 / 3, p1=1 / 3)
    assert abs(P.filter_eval(m, 0, 0) - (8 / 9) ** 2) < 1e-6 and abs(P.filter_eval(m, 2, 0)) < 1e-6
    l = dict(type=4, xw=3.0, yw=3.0, p0=3.0, p1=0)
    assert P.filter_eval(l, 0, 0) == 1 and abs(P.filter_eval(l, 3, 0)) < 1e-6


def test_reinhard_of_a_grey_image():
    """uniform luminance L: Lp = key, Lwhite = key -> Y = key (1 + key / key^2 ... ) / (1 + key) = key (1 + 1/key) / (1 + key) = 1"""
    grey = np.zeros((6, 6, 7), np.float32); grey[..., :3] = 0.5; grey[..., 6] = 1
    filtered = P.to_rgbe(P.to_spectrum(grey, 0.0))
    mn, mx, avg, log_avg = P.luminance_info(filtered)
    assert abs(mn - 0.5) < 1e-6 and abs(mx - 0.5) < 1e-6 and abs(avg - 0.5) < 1e-6 and abs(log_avg - (0.5 + 2.3e-5)) < 1e-5
    out = P.reinhard(filtered, 0.18, 0.0)
    assert np.all(out[..., :3] >= 253)                             # maps to white: the brightest pixel is the white point
    final = P.apply_image_pipeline(grey, 0.0, None, dict(key=0.18, burn=0.0))
    assert np.all(final[..., :3] >= 253)
    # two luminance levels, half the pixels each: the operator in closed form
    two = grey.copy(); two[:, :3, :3] = 0.05
    out = P.reinhard(P.to_rgbe(P.to_spectrum(two, 0.0)), 0.18, 0.0).astype(int)
    lo = P.from_rgbe(P.to_rgbe(np.float32([0.05] * 3)))[0]       # what RGBE keeps of 0.05
    scale = 0.18 / np.exp(0.5 * (np.log(2.3e-5 + lo) + np.log(2.3e-5 + 0.5)))
    lw = 0.5 * scale
    y = lambda L: (L * scale) * (1 + L * scale / lw ** 2) / (1 + L * scale)
    assert abs(out[0, 0, 0] - int(y(lo) * 255)) <= 1 and out[0, 5, 0] >= 253


def test_filter_functions_against_the_reference():
    """Box / Gaussian / Mitchell / Lanczos-sinc / triangle Evaluate (SceneTypes/Filter.h:28-171) of the numpy restatement against values computed by the
    reference's own header (tests/golden/filters.npz).  Box, Mitchell and triangle are polynomial: bit for bit.  Gaussian and Lanczos go through
    exp / sin of the C library there and of numpy here: 4 ulp of the larger factor."""
    import os
    from oracle import pipeline as P
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "filters.npz"))
    worst = {}
    for cfg, xy, want in zip(g["cfg"], g["xy"], g["value"]):
        t = int(cfg[0])
        got = np.float32(P.filter_eval(dict(type=t, xw=float(cfg[1]), yw=float(cfg[2]), p0=float(cfg[3]), p1=float(cfg[4])), xy[0], xy[1]))
        if t in (1, 3, 5):
            assert got.view(np.uint32) == want.view(np.uint32), (t, cfg, xy, got, want)
        else:
            err = abs(float(got) - float(want)); worst[t] = max(worst.get(t, 0.0), err / max(abs(float(want)), 1e-3))
    assert set(worst) == {2, 4} and max(worst.values()) <= 4 * 1.2e-7 * 8, worst
