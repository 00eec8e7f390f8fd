"""ORACLE (test infrastructure, never imported by the product): numpy restatement of the reference's image pipeline —
applyImagePipeline (Kernel/ImagePipeline/ImagePipeline.cu:54-84), CanonicalFilter (Filter/CanonicalFilter.cu:6-44) over the
reconstruction filters of SceneTypes/Filter.h, ToneMapPostProcess (PostProcess/ToneMapPostProcess.cu:6-42) and
Image::ComputeLuminanceInfo (Engine/Image.cu:88-168).  All arithmetic in float32, loops in the reference's order (rows of the
filter footprint outermost).  Parity unpinned: these files contain kernels and do not compile here; the tests check closed forms
(box filter = window mean, constant image stays constant, Reinhard of a grey image) besides the GPU comparison.
"""
import numpy as np

F = np.float32


def to_spectrum(px, splat_scale):
    """PixelData::toSpectrum (Engine/Image.h:21-28); px = (h, w, 7) float32"""
    w = np.where(px[..., 6] != 0, px[..., 6], F(1))[..., None]
    return (px[..., 0:3] / w + px[..., 3:6] * F(splat_scale)).astype(F)


def to_rgbe(c):
    """SpectrumConverter::Float3ToRGBE (Math/Spectrum.h:534-555) -> uint32 (r | g << 8 | b << 16 | e << 24)"""
    c = np.asarray(c, F)
    m = np.max(c, axis=-1)
    ok = m >= F(1e-32)
    safe = np.where(ok, m, F(1))
    mant, e = np.frexp(safe.astype(np.float64))
    f = (mant.astype(F) * F(256.0) / safe).astype(F)
    q = np.clip((c * f[..., None]).astype(np.int64), 0, 255).astype(np.uint32)   # (unsigned char)(c * f): values are < 256 by construction
    out = q[..., 0] | (q[..., 1] << 8) | (q[..., 2] << 16) | (((e + 128) & 0xff).astype(np.uint32) << 24)
    return np.where(ok, out, np.uint32(0)).astype(np.uint32)


def from_rgbe(v):
    """RGBEToFloat3 (Math/Spectrum.h:557-565)"""
    v = np.asarray(v, np.uint32)
    w = (v >> 24).astype(np.int32)
    e = np.ldexp(F(1.0), w - (128 + 8)).astype(F)
    rgb = np.stack([(v & 0xff), ((v >> 8) & 0xff), ((v >> 16) & 0xff)], axis=-1).astype(F) * e[..., None]
    return np.where((w != 0)[..., None], rgb, F(0)).astype(F)


def to_rgbcol(c):
    """Float3ToCOLORREF (Math/Spectrum.h:521-526) -> (…, 4) uint8"""
    q = (np.clip(np.asarray(c, F), F(0), F(1)) * F(255.0)).astype(np.uint8)
    return np.concatenate([q, np.full(q.shape[:-1] + (1,), 255, np.uint8)], axis=-1)


def from_rgbcol(q):
    return (np.asarray(q)[..., :3].astype(F) / F(255.0)).astype(F)


def srgb(v):
    """toSRGBComponent (Math/Spectrum.cu:229-235)"""
    v = np.asarray(v, F)
    with np.errstate(invalid="ignore"):
        return np.where(v <= F(0.0031308), F(12.92) * v, F(1.055) * np.power(np.maximum(v, F(0)), F(1.0 / 2.4)) - F(0.055)).astype(F)


def gamma_correct(c):
    return to_rgbcol(srgb(c))


def luminance(c):
    c = np.asarray(c, F)
    return (c[..., 0] * F(0.212671) + c[..., 1] * F(0.715160) + c[..., 2] * F(0.072169)).astype(F)


def filter_eval(flt, x, y):
    """Filter::Evaluate(|dx|, |dy|) (SceneTypes/Filter.h); flt = dict(type, xw, yw, p0, p1)"""
    x, y = F(x), F(y)
    t = flt["type"]
    if t == 1:
        return F(1)
    if t == 2:
        a = F(flt["p0"])
        ex, ey = np.exp(-a * F(flt["xw"]) * F(flt["xw"])), np.exp(-a * F(flt["yw"]) * F(flt["yw"]))
        return F(max(F(0), F(np.exp(-a * x * x)) - F(ex))) * F(max(F(0), F(np.exp(-a * y * y)) - F(ey)))
    if t == 3:
        B, Cc = F(flt["p0"]), F(flt["p1"])

        def m1(v):
            v = F(abs(F(2) * v))
            if v > 1:
                return F(((-B - 6 * Cc) * v * v * v + (6 * B + 30 * Cc) * v * v + (-12 * B - 48 * Cc) * v + (8 * B + 24 * Cc)) * F(1.0 / 6.0))
            return F(((12 - 9 * B - 6 * Cc) * v * v * v + (-18 + 12 * B + 6 * Cc) * v * v + (6 - 2 * B)) * F(1.0 / 6.0))
        return F(m1(x * F(1.0 / flt["xw"])) * m1(y * F(1.0 / flt["yw"])))
    if t == 4:
        tau = F(flt["p0"])

        def s1(v):
            v = F(abs(v))
            if v < 1e-5:
                return F(1)
            if v > 1:
                return F(0)
            v = F(v * F(np.pi))
            return F((np.sin(v) / v) * (np.sin(v * tau) / (v * tau)))
        return F(s1(x * F(1.0 / flt["xw"])) * s1(y * F(1.0 / flt["yw"])))
    return F(max(F(0), F(flt["xw"]) - abs(x))) * F(max(F(0), F(flt["yw"]) - abs(y)))


def canonical_filter(px, splat_scale, flt):
    """rtm_Copy / evalFilter (CanonicalFilter.cu:6-36) -> RGBE image (h, w) uint32"""
    h, w = px.shape[:2]
    spec = to_spectrum(px, splat_scale)
    rx, ry = int(np.floor(flt["xw"])), int(np.floor(flt["yw"]))
    acc = np.zeros((h, w, 3), F); accw = np.zeros((h, w), F)
    ys, xs = np.mgrid[0:h, 0:w]
    for dy in range(-ry, ry + 1):          # y0..y1 ascending = dy ascending, then x
        for dx in range(-rx, rx + 1):
            if abs(dx) > flt["xw"] or abs(dy) > flt["yw"]:
                continue
            wt = filter_eval(flt, abs(dx), abs(dy))
            yy, xx = ys + dy, xs + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            src = spec[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
            acc = np.where(ok[..., None], (acc + src * wt).astype(F), acc)
            accw = np.where(ok, (accw + wt).astype(F), accw)
    with np.errstate(invalid="ignore", divide="ignore"):
        return to_rgbe((acc / accw[..., None]).astype(F))


def luminance_info(filtered):
    """Image::ComputeLuminanceInfo: (min, max, avg, exp(mean log(2.3e-5 + Y)))"""
    Y = luminance(from_rgbe(filtered))
    n = F(Y.size)
    return F(Y.min()), F(Y.max()), F(Y.sum(dtype=np.float64) / n), F(np.exp(F(np.log(F(2.3e-5) + Y).sum(dtype=np.float64) / n)))


def reinhard(filtered, key=0.18, burn=0.0):
    """ToneMapPostProcess::Apply + Reinhard05Kernel -> RGBCOL (h, w, 4) BEFORE the final gamma pass"""
    _, max_lum, _, log_avg = luminance_info(filtered)
    scale = F(F(key) / log_avg); lwhite = F(max_lum * scale)
    b = F(min(1.0, max(1e-8, 1.0 - burn)))
    inv_wp2 = F(1) / F(lwhite * lwhite * F(np.power(b, F(4.0))))
    c = from_rgbe(filtered)
    X = c[..., 0] * F(0.412453) + c[..., 1] * F(0.357580) + c[..., 2] * F(0.180423)
    Y0 = c[..., 0] * F(0.212671) + c[..., 1] * F(0.715160) + c[..., 2] * F(0.072169)
    Z = c[..., 0] * F(0.019334) + c[..., 1] * F(0.119193) + c[..., 2] * F(0.950227)
    s = np.clip(X + Y0 + Z, F(0.001), F(100000.0))
    x, y = X / s, Y0 / s
    Lp = scale * Y0
    Y = Lp * (F(1) + Lp * inv_wp2) / (F(1) + Lp)
    yc = np.clip(y, F(0.001), F(100000.0))
    X2, Z2 = Y / yc * x, Y / yc * (F(1) - x - y)
    rgb = np.stack([F(3.240479) * X2 + F(-1.537150) * Y + F(-0.498535) * Z2, F(-0.969256) * X2 + F(1.875991) * Y + F(0.041556) * Z2,
                    F(0.055648) * X2 + F(-0.204043) * Y + F(1.057311) * Z2], axis=-1).astype(F)
    return to_rgbcol(rgb)


def apply_image_pipeline(px, splat_scale, flt=None, process=None):
    """applyImagePipeline (ImagePipeline.cu:54-84) -> (h, w, 4) uint8"""
    if flt is None and process is None:
        return gamma_correct(to_spectrum(px, splat_scale))
    filtered = canonical_filter(px, splat_scale, flt) if flt is not None else to_rgbe(to_spectrum(px, splat_scale))
    if process is None:
        return gamma_correct(from_rgbe(filtered))
    out = reinhard(filtered, process.get("key", 0.18), process.get("burn", 0.0))
    return gamma_correct(from_rgbcol(out))
