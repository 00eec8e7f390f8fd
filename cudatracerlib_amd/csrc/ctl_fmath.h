// ctl_fmath.h — the transcendental functions of the shading code (sin, cos, sincos, tan, acos, atan, atan2, exp, log, log2, pow) as ONE implementation that gives the
// same bits on the host and on the device.
//
// Why: the reference's CPU path (and the oracle that restates it) calls glibc's sinf / cosf / acosf / atan2f / expf / logf / powf; a HIP kernel calls the device
// library's.  Both are good to 1-2 ulp, but not to the SAME ulp, and a last-bit difference in a sampled direction now and then flips a discrete decision further
// down the path (Russian roulette, which light, which lobe, hit or miss of a small sphere) — which is why a depth-8 frame used to need a statistical bar.
// Here every function is evaluated in IEEE double arithmetic — +, -, *, /, sqrt, rint and bit manipulation only, no FMA contraction (-ffp-contract=off on both
// sides), no library call — and rounded to float once: the result is within 1 ulp of the exact value (0.5 ulp + 1e-9 before the final rounding) and is bit-identical
// wherever IEEE-754 double arithmetic is (x86-64 and gfx950 both are; the MI355X runs fp64 at half its fp32 rate, and the shade kernels wait on memory).
//
// Who uses it: the HIP shading code (always), and the oracle when it is built -DORC_SHARED_MATH (oracle/liboracle_sm.so: the checker of the GPU parity tests).  The
// default oracle build keeps glibc — that one is pinned bit for bit on the reference's own code (tests/golden) — and tests/test_fmath.py holds the two together:
// every function within 1 ulp of glibc on dense samples, and the two oracle builds within the render tolerance of each other.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define CTL_FM_HD __host__ __device__ inline
#else
#define CTL_FM_HD inline
#endif

namespace ctl {
namespace fm {

CTL_FM_HD double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
CTL_FM_HD uint64_t to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }
CTL_FM_HD uint32_t fbits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
CTL_FM_HD float ffrom(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
CTL_FM_HD float fnan() { return ffrom(0x7fc00000u); }
CTL_FM_HD float finf() { return ffrom(0x7f800000u); }
CTL_FM_HD bool isnan_(float x) { return x != x; }

constexpr double kPi = 3.14159265358979323846, kPio2 = 1.57079632679489661923, kPio4 = 0.78539816339744830962;

// sin and cos of a finite |x| < 2^30 in double: Cody-Waite reduction by pi/2 (two parts), Taylor polynomials on [-pi/4, pi/4] (truncation < 1e-13)
CTL_FM_HD void sincos_d(double x, double& s, double& c) {
    const double two_over_pi = 0.63661977236758134308, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double kd = __builtin_rint(x * two_over_pi);
    const double r = (x - kd * pio2_hi) - kd * pio2_lo, z = r * r;
    const double sp = r * (1.0 + z * (-1.0 / 6 + z * (1.0 / 120 + z * (-1.0 / 5040 + z * (1.0 / 362880 + z * (-1.0 / 39916800 + z * (1.0 / 6227020800.0)))))));
    const double cp = 1.0 + z * (-0.5 + z * (1.0 / 24 + z * (-1.0 / 720 + z * (1.0 / 40320 + z * (-1.0 / 3628800 + z * (1.0 / 479001600 + z * (-1.0 / 87178291200.0)))))));
    switch ((long long)kd & 3) {
    case 0: s = sp; c = cp; break;
    case 1: s = cp; c = -sp; break;
    case 2: s = -sp; c = -cp; break;
    default: s = -cp; c = sp; break;
    }
}
CTL_FM_HD bool trig_domain(float x) { const float a = x < 0 ? -x : x; return a < 1073741824.0f; }   // finite and below 2^30 (NaN fails the comparison)
CTL_FM_HD float sin(float x) { if (!trig_domain(x)) return fnan(); if (x == 0.0f) return x; double s, c; sincos_d((double)x, s, c); return (float)s; }
CTL_FM_HD float cos(float x) { if (!trig_domain(x)) return fnan(); double s, c; sincos_d((double)x, s, c); return (float)c; }
CTL_FM_HD void sincos(float x, float* sp, float* cp) {
    if (!trig_domain(x)) { *sp = *cp = fnan(); return; }
    double s, c; sincos_d((double)x, s, c);
    *sp = x == 0.0f ? x : (float)s; *cp = (float)c;
}
CTL_FM_HD float tan(float x) { if (!trig_domain(x)) return fnan(); if (x == 0.0f) return x; double s, c; sincos_d((double)x, s, c); return (float)(s / c); }

// atan of t >= 0 (finite or +inf) in double: two argument reductions (tan(3 pi / 8), tan(pi / 8)), odd series on |u| <= tan(pi / 8) (truncation < 2e-12)
CTL_FM_HD double atan_pos_d(double t) {
    double a, u;
    if (t > 2.41421356237309504880) { a = kPio2; u = -1.0 / t; }            // +inf: u = -0
    else if (t > 0.41421356237309504880) { a = kPio4; u = (t - 1.0) / (t + 1.0); }
    else { a = 0.0; u = t; }
    const double z = u * u;
    const double p = u * (1.0 + z * (-1.0 / 3 + z * (1.0 / 5 + z * (-1.0 / 7 + z * (1.0 / 9 + z * (-1.0 / 11 + z * (1.0 / 13 + z * (-1.0 / 15 + z * (1.0 / 17 + z * (-1.0 / 19 + z * (1.0 / 21
                     + z * (-1.0 / 23 + z * (1.0 / 25 + z * (-1.0 / 27))))))))))))));
    return a + p;
}
CTL_FM_HD float atan(float x) {
    if (isnan_(x)) return x;
    if (x == 0.0f) return x;
    const double r = atan_pos_d(x < 0 ? -(double)x : (double)x);
    return (float)(x < 0 ? -r : r);
}
CTL_FM_HD float atan2(float y, float x) {
    if (isnan_(x) || isnan_(y)) return fnan();
    const bool yneg = (fbits(y) >> 31) != 0, xneg = (fbits(x) >> 31) != 0;
    double r;
    if (y == 0.0f) r = xneg ? kPi : 0.0;                                        // atan2(+-0, x)
    else if (x == 0.0f) r = kPio2;
    else {
        const double ay = yneg ? -(double)y : (double)y, ax = xneg ? -(double)x : (double)x;
        const bool yinf = ay > 3.5e38, xinf = ax > 3.5e38;
        double a;
        if (yinf && xinf) a = kPio4; else if (yinf) a = kPio2; else if (xinf) a = 0.0; else a = atan_pos_d(ay / ax);
        r = xneg ? kPi - a : a;
    }
    return (float)(yneg ? -r : r);
}
CTL_FM_HD float acos(float x) {
    if (isnan_(x) || x > 1.0f || x < -1.0f) return fnan();
    if (x == -1.0f) return (float)kPi;
    const double d = (double)x;
    return (float)(2.0 * atan_pos_d(__builtin_sqrt((1.0 - d) / (1.0 + d))));
}

// exp of a double in [-120, 100]: reduction by ln 2 (two parts), Taylor polynomial on [-ln2/2, ln2/2] (truncation < 1e-14), scaling by an exactly constructed 2^k
CTL_FM_HD double exp_d(double x) {
    const double inv_ln2 = 1.44269504088896338700, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double kd = __builtin_rint(x * inv_ln2);
    const double r = (x - kd * ln2_hi) - kd * ln2_lo;
    const double p = 1.0 + r * (1.0 + r * (0.5 + r * (1.0 / 6 + r * (1.0 / 24 + r * (1.0 / 120 + r * (1.0 / 720 + r * (1.0 / 5040 + r * (1.0 / 40320 + r * (1.0 / 362880 + r * (1.0 / 3628800
                     + r * (1.0 / 39916800 + r * (1.0 / 479001600))))))))))));
    return p * from_bits((uint64_t)((long long)kd + 1023) << 52);
}
CTL_FM_HD float exp(float x) {
    if (isnan_(x)) return x;
    if (x > 89.0f) return finf();
    if (x < -104.0f) return 0.0f;
    return (float)exp_d((double)x);
}
// natural logarithm of a finite double > 0 that is normal (every positive float is): x = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m - 1) / (m + 1))
CTL_FM_HD double log_d(double x) {
    const uint64_t b = to_bits(x);
    int e = (int)(b >> 52) - 1023;
    double m = from_bits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.41421356237309504880) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    const double p = 2.0 * s * (1.0 + z * (1.0 / 3 + z * (1.0 / 5 + z * (1.0 / 7 + z * (1.0 / 9 + z * (1.0 / 11 + z * (1.0 / 13 + z * (1.0 / 15 + z * (1.0 / 17 + z * (1.0 / 19))))))))));
    return (double)e * 0.69314718055994530942 + p;
}
CTL_FM_HD float log(float x) {
    if (isnan_(x)) return x;
    if (x < 0.0f) return fnan();
    if (x == 0.0f) return -finf();
    if (x > 3.4e38f) return x;   // +inf
    return (float)log_d((double)x);
}
CTL_FM_HD float log2(float x) {
    if (isnan_(x)) return x;
    if (x < 0.0f) return fnan();
    if (x == 0.0f) return -finf();
    if (x > 3.4e38f) return x;
    return (float)(log_d((double)x) * 1.44269504088896340736);
}
// pow as C's powf defines it (the cases the shading code can reach and the IEEE special values), exp(y log x) in double
CTL_FM_HD float pow(float x, float y) {
    if (y == 0.0f || x == 1.0f) return 1.0f;
    if (isnan_(x) || isnan_(y)) return fnan();
    const float ay = y < 0 ? -y : y;
    const bool y_int = ay >= 8388608.0f || (float)(long long)ay == ay;                     // |y| >= 2^23: every float is an integer
    const bool y_odd = y_int && ay < 16777216.0f && (((long long)ay) & 1);
    if (x == 0.0f) { const bool neg = (fbits(x) >> 31) && y_odd; if (y > 0) return neg ? -0.0f : 0.0f; return neg ? -finf() : finf(); }
    if (ay > 3.4e38f) { const float ax = x < 0 ? -x : x; if (ax == 1.0f) return 1.0f; return ((ax > 1.0f) == (y > 0)) ? finf() : 0.0f; }   // y = +-inf
    float sign = 1.0f; double ax = (double)x;
    if (x < 0.0f) { if (!y_int) return fnan(); ax = -ax; if (y_odd) sign = -1.0f; }
    if (ax > 3.4e38) return y > 0 ? sign * finf() : sign * 0.0f;                            // x = +-inf
    if (y_int && ay <= 6.0f) {                                                              // the shading code's cos^4, (1 - c)^5, x^2 ...: a few exact-enough double products
        double r = ax; const int n = (int)ay;
        for (int k = 1; k < n; k++) r *= ax;
        return sign * (float)(y < 0 ? 1.0 / r : r);
    }
    if (y == 0.25f) return (float)__builtin_sqrt(__builtin_sqrt(ax));   // the warp of the rough-transmittance tables (x >= 0 here): two correctly rounded double roots instead of log + exp
    const double t = (double)y * log_d(ax);
    if (t > 100.0) return sign * finf();
    if (t < -120.0) return sign * 0.0f;
    return sign * (float)exp_d(t);
}

}  // namespace fm
}  // namespace ctl
