// ctl_math.h — fp32 vector / matrix / frame / warp / codec primitives of the MI355X path tracer.
// Host + device (HD): the scene builder runs them on the host, the HIP kernels on gfx950.
// The whole library is compiled with -ffp-contract=off: results must not depend on where the compiler
// finds an FMA; the few places that want one (slab tests) call __builtin_fmaf explicitly.
// Behavioural contract per function = the cited reference lines (paths relative to the reference root).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
#include <cfloat>
#include "ctl_fmath.h"

#define HD __host__ __device__ __forceinline__

namespace ctl {

// ---- transcendental functions: on the device the shared fp32 implementation of ctl_fmath.h (bit-identical to the oracle's -DORC_SHARED_MATH build), on the host libm —
// host code only builds scene data (normal codec of TriangleData, light cosines), which is pinned on the reference's own glibc results (tests/golden)
#ifndef CTL_SHADE_PROBE
#define CTL_SHADE_PROBE 0   // timing probes only (tools/shade_basic_probe.py; results are WRONG in such a build): 1 = the device library's fp32 sin / cos / ... instead of ctl_fmath.h; 2 = the normal codec without trigonometry
#endif
#if defined(__HIP_DEVICE_COMPILE__) && (CTL_SHADE_PROBE & 1)
HD float m_sin(float x) { return __sinf(x); }  HD float m_cos(float x) { return __cosf(x); }  HD float m_tan(float x) { return __tanf(x); }
HD void m_sincos(float x, float* s, float* c) { *s = __sinf(x); *c = __cosf(x); }
HD float m_acos(float x) { return ::acosf(x); }  HD float m_atan(float x) { return ::atanf(x); }  HD float m_atan2(float y, float x) { return ::atan2f(y, x); }
HD float m_exp(float x) { return __expf(x); }  HD float m_log(float x) { return __logf(x); }  HD float m_log2(float x) { return __log2f(x); }  HD float m_pow(float x, float y) { return __powf(x, y); }
#elif defined(__HIP_DEVICE_COMPILE__) && defined(CTL_FMATH_OUTLINE)
// one out-of-line copy of each function per code object instead of one inlined per call site: ctl_fmath.h is 39 % of the code of a model-class shade kernel (293 KB against a 64-KB
// instruction cache); arguments and results travel in registers
namespace fm_ol {
struct sc { float s, c; };
static __device__ __noinline__ float sin_(float x) { return fm::sin(x); }    static __device__ __noinline__ float cos_(float x) { return fm::cos(x); }
static __device__ __noinline__ float tan_(float x) { return fm::tan(x); }    static __device__ __noinline__ sc sincos_(float x) { sc r; fm::sincos(x, &r.s, &r.c); return r; }
static __device__ __noinline__ float acos_(float x) { return fm::acos(x); }  static __device__ __noinline__ float atan_(float x) { return fm::atan(x); }
static __device__ __noinline__ float atan2_(float y, float x) { return fm::atan2(y, x); }
static __device__ __noinline__ float exp_(float x) { return fm::exp(x); }    static __device__ __noinline__ float log_(float x) { return fm::log(x); }
static __device__ __noinline__ float log2_(float x) { return fm::log2(x); }  static __device__ __noinline__ float pow_(float x, float y) { return fm::pow(x, y); }
}
HD float m_sin(float x) { return fm_ol::sin_(x); }  HD float m_cos(float x) { return fm_ol::cos_(x); }  HD float m_tan(float x) { return fm_ol::tan_(x); }
HD void m_sincos(float x, float* s, float* c) { const fm_ol::sc r = fm_ol::sincos_(x); *s = r.s; *c = r.c; }
HD float m_acos(float x) { return fm_ol::acos_(x); }  HD float m_atan(float x) { return fm_ol::atan_(x); }  HD float m_atan2(float y, float x) { return fm_ol::atan2_(y, x); }
HD float m_exp(float x) { return fm_ol::exp_(x); }  HD float m_log(float x) { return fm_ol::log_(x); }  HD float m_log2(float x) { return fm_ol::log2_(x); }  HD float m_pow(float x, float y) { return fm_ol::pow_(x, y); }
#elif defined(__HIP_DEVICE_COMPILE__)
HD float m_sin(float x) { return fm::sin(x); }  HD float m_cos(float x) { return fm::cos(x); }  HD float m_tan(float x) { return fm::tan(x); }
HD void m_sincos(float x, float* s, float* c) { fm::sincos(x, s, c); }
HD float m_acos(float x) { return fm::acos(x); }  HD float m_atan(float x) { return fm::atan(x); }  HD float m_atan2(float y, float x) { return fm::atan2(y, x); }
HD float m_exp(float x) { return fm::exp(x); }  HD float m_log(float x) { return fm::log(x); }  HD float m_log2(float x) { return fm::log2(x); }  HD float m_pow(float x, float y) { return fm::pow(x, y); }
#else
HD float m_sin(float x) { return ::sinf(x); }  HD float m_cos(float x) { return ::cosf(x); }  HD float m_tan(float x) { return ::tanf(x); }
HD void m_sincos(float x, float* s, float* c) { *s = ::sinf(x); *c = ::cosf(x); }
HD float m_acos(float x) { return ::acosf(x); }  HD float m_atan(float x) { return ::atanf(x); }  HD float m_atan2(float y, float x) { return ::atan2f(y, x); }
HD float m_exp(float x) { return ::expf(x); }  HD float m_log(float x) { return ::logf(x); }  HD float m_log2(float x) { return ::log2f(x); }  HD float m_pow(float x, float y) { return ::powf(x, y); }
#endif

static constexpr float kPi = 3.14159265358979f;          // Math/MathFunc.h:12
static constexpr float kInvPi = 1.0f / kPi;
static constexpr float kInvTwoPi = 1.0f / (2.0f * kPi);
static constexpr float kDeltaEpsilon = 1e-3f;            // Math/MathFunc.h:26
static constexpr int kSentinel = 0x76543210;             // Kernel/TraceHelper.cu:20

struct f2 { float x, y; };
struct f3 {
    float x, y, z;
    HD f3() {}
    HD f3(float a) : x(a), y(a), z(a) {}
    HD f3(float a, float b, float c) : x(a), y(b), z(c) {}
};
HD f3 operator+(f3 a, f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
HD f3 operator-(f3 a, f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
HD f3 operator*(f3 a, f3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
HD f3 operator/(f3 a, f3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
HD f3 operator*(f3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
HD f3 operator*(float s, f3 a) { return f3(a.x * s, a.y * s, a.z * s); }
HD f3 operator/(f3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
// Spectrum / scalar of the reference multiplies by the reciprocal (TSpectrum::operator/(Scalar), operator/=(Scalar), Math/Spectrum.h:122-128, :150-155), Vec3f / scalar
// divides (Math/Vector.h:88): every Spectrum-by-scalar division of the path is written sdiv()
HD f3 sdiv(f3 a, float s) { const float r = 1.0f / s; return f3(a.x * r, a.y * r, a.z * r); }
HD f3 operator-(f3 a) { return f3(-a.x, -a.y, -a.z); }
// Math/Vector.h:101 — accumulate from 0 in component order
HD float dot(f3 a, f3 b) { float r = a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }
HD float absdot(f3 a, f3 b) { return fabsf(dot(a, b)); }
HD f3 cross(f3 a, f3 v) { return f3(a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x); }   // Vector.h:329
HD float len_sqr(f3 a) { float r = a.x * a.x; r += a.y * a.y; r += a.z * a.z; return r; }
HD float length(f3 a) { return sqrtf(len_sqr(a)); }
HD f3 normalize(f3 a) { return a * (1.0f / length(a)); }                                                         // Vector.h:369-371
HD float min2(float a, float b) { return (a < b) ? a : b; }                                                      // MathFunc.h:96
HD float max2(float a, float b) { return (a > b) ? a : b; }
HD float max3c(f3 a) { float r = a.x; r = max2(r, a.y); r = max2(r, a.z); return r; }                            // Vector.h:50
HD bool is_zero(f3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
HD float clampf(float v, float lo, float hi) { return min2(max2(v, lo), hi); }
HD float safe_sqrt(float v) { return sqrtf(max2(0.0f, v)); }
HD float fracf(float f) { return f - floorf(f); }                                                                // MathFunc.h:138
HD float copysign_bits(float a, float b) {
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, b) & 0x80000000u) | (__builtin_bit_cast(uint32_t, a) & 0x7fffffffu));
}

// ---- affine 3x4 view of a row-major float4x4 (Math/float4x4.h:383-412): s = 0; s += m[i][k] * v[k] ...
struct m34 { float r[3][4]; };
HD f3 xform_point(const m34& m, f3 p) {   // TransformPoint for an affine map: w == 1 exactly
    f3 o;
    float s;
    s = m.r[0][0] * p.x; s += m.r[0][1] * p.y; s += m.r[0][2] * p.z; s += m.r[0][3]; o.x = s;
    s = m.r[1][0] * p.x; s += m.r[1][1] * p.y; s += m.r[1][2] * p.z; s += m.r[1][3]; o.y = s;
    s = m.r[2][0] * p.x; s += m.r[2][1] * p.y; s += m.r[2][2] * p.z; s += m.r[2][3]; o.z = s;
    return o;
}
// general TransformPoint of a matrix whose last row is (0,0,0,w): the reference divides by w (float4x4.h:402-406)
HD f3 xform_point_w(const m34& m, f3 p, float w) { f3 r = xform_point(m, p); return f3(r.x / w, r.y / w, r.z / w); }
HD f3 xform_dir(const m34& m, f3 p) {
    f3 o;
    float s;
    s = m.r[0][0] * p.x; s += m.r[0][1] * p.y; s += m.r[0][2] * p.z; s += m.r[0][3] * 0.0f; o.x = s;
    s = m.r[1][0] * p.x; s += m.r[1][1] * p.y; s += m.r[1][2] * p.z; s += m.r[1][3] * 0.0f; o.y = s;
    s = m.r[2][0] * p.x; s += m.r[2][1] * p.y; s += m.r[2][2] * p.z; s += m.r[2][3] * 0.0f; o.z = s;
    return o;
}

// ---- local shading frame (Math/Frame.h)
HD void coordinate_system(f3 a, f3& s, f3& t) {   // Frame.h:9-22
    if (fabsf(a.x) > fabsf(a.y)) { float il = 1.0f / sqrtf(a.x * a.x + a.z * a.z); t = f3(a.z * il, 0.0f, -a.x * il); }
    else { float il = 1.0f / sqrtf(a.y * a.y + a.z * a.z); t = f3(0.0f, a.z * il, -a.y * il); }
    s = normalize(cross(t, a));
}
struct frame {
    f3 s, t, n;
    HD f3 to_local(f3 v) const { return f3(dot(v, s), dot(v, t), dot(v, n)); }
    HD f3 to_world(f3 v) const { return s * v.x + t * v.y + n * v.z; }
};
HD float cos_theta(f3 v) { return v.z; }
HD float sin_theta2(f3 v) { return 1.0f - v.z * v.z; }
HD float sin_theta(f3 v) { float t = sin_theta2(v); return t <= 0.0f ? 0.0f : sqrtf(t); }
HD float tan_theta(f3 v) { float t = 1 - v.z * v.z; return t <= 0.0f ? 0.0f : sqrtf(t) / v.z; }
HD f3 reflect_local(f3 wi) { return f3(-wi.x, -wi.y, wi.z); }                                    // Frame.h:134-136
HD f3 refract_local(f3 wi, float cosThetaT, float eta, float invEta) {                            // Frame.h:144-152
    float scale = -(cosThetaT < 0 ? invEta : eta);
    return normalize(f3(scale * wi.x, scale * wi.y, cosThetaT));
}

// ---- warps (Math/Warp.h)
HD f2 square_to_disk_concentric(f2 s) {   // Warp.h:104-127
    float r1 = 2.0f * s.x - 1.0f, r2 = 2.0f * s.y - 1.0f, phi, r;
    if (r1 == 0 && r2 == 0) { r = phi = 0; }
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (kPi / 4.0f) * (r2 / r1); }
    else { r = r2; phi = (kPi / 2.0f) - (r1 / r2) * (kPi / 4.0f); }
    float sp, cp;
#ifdef __HIP_DEVICE_COMPILE__
    m_sincos(phi, &sp, &cp);
#else
    sp = m_sin(phi); cp = m_cos(phi);
#endif
    return f2{ r * cp, r * sp };
}
HD f3 square_to_cosine_hemisphere(f2 s) {   // Warp.h:61-66
    f2 p = square_to_disk_concentric(s);
    return f3(p.x, p.y, sqrtf(1.0f - p.x * p.x - p.y * p.y));
}
HD f2 square_to_uniform_triangle(f2 s) { float a = sqrtf(1.0f - s.x); return f2{ 1 - a, a * s.y }; }   // Warp.h:160-164

HD float power_heuristic(float fPdf, float gPdf) { float f = 1 * fPdf, g = 1 * gPdf; return (f * f) / (f * f + g * g); }   // MonteCarlo.h:29-33

// ---- Fresnel (Math/FresnelHelper.h:27-58, 119-146)
HD float fresnel_dielectric_ext(float cosThetaI_, float& cosThetaT_, float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    float scale = (cosThetaI_ > 0) ? 1.0f / eta : eta, cosThetaTSqr = 1.0f - (1.0f - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = fabsf(cosThetaI_), cosThetaT = safe_sqrt(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
HD float fresnel_conductor_exact1(float cosThetaI, float e, float k) {
    float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    float temp1 = e * e - k * k - sinThetaI2;
    float a2pb2 = safe_sqrt(temp1 * temp1 + k * k * e * e * 4);
    float a = safe_sqrt((a2pb2 + temp1) * 0.5f);
    float term1 = a2pb2 + cosThetaI2, term2 = a * (2 * cosThetaI);
    float Rs2 = (term1 - term2) / (term1 + term2);
    float term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
    float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (Rp2 + Rs2);
}
HD f3 fresnel_conductor_exact(float c, f3 eta, f3 k) {
    return f3(fresnel_conductor_exact1(c, eta.x, k.x), fresnel_conductor_exact1(c, eta.y, k.y), fresnel_conductor_exact1(c, eta.z, k.z));
}

// ---- fp16 (Math/half.h): IEEE binary16, round-to-nearest-even in, exact out (the device behaviour of the reference)
HD uint16_t float_to_half(float f) {
    uint32_t ia = __builtin_bit_cast(uint32_t, f);
    uint16_t ir = (ia >> 16) & 0x8000;
    if ((ia & 0x7f800000) == 0x7f800000) { if ((ia & 0x7fffffff) == 0x7f800000) ir |= 0x7c00; else ir = 0x7fff; }
    else if ((ia & 0x7f800000) >= 0x33000000) {
        int shift = (int)((ia >> 23) & 0xff) - 127;
        if (shift > 15) ir |= 0x7c00;
        else {
            ia = (ia & 0x007fffff) | 0x00800000;
            if (shift < -14) { ir |= ia >> (-1 - shift); ia = ia << (32 - (-1 - shift)); }
            else { ir |= ia >> (24 - 11); ia = ia << (32 - (24 - 11)); ir = ir + ((14 + shift) << 10); }
            if ((ia > 0x80000000) || ((ia == 0x80000000) && (ir & 1))) ir++;
        }
    }
    return ir;
}
HD float half_to_float(uint16_t h) {
#ifdef __HIP_DEVICE_COMPILE__
    return (float)__builtin_bit_cast(_Float16, h);   // v_cvt_f32_f16
#else
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ff, out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else { int e = -1; do { e++; man <<= 1; } while ((man & 0x400) == 0); out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13); }
    } else if (exp == 31) out = sign | 0x7f800000 | (man << 13);
    else out = sign | ((exp + 112) << 23) | (man << 13);
    return __builtin_bit_cast(float, out);
#endif
}

// ---- 8+8 bit spherical normal codec (Math/Compression.h:12-31)
HD uint16_t normal_to_uchar2(f3 v) {
    float theta = (m_acos(v.z) * (255.0f / kPi));
    float phi = (m_atan2(v.y, v.x) * (255.0f / (2.0f * kPi)));
    phi = phi < 0 ? (phi + 255) : phi;
    return (uint16_t)(((unsigned short)theta << 8) | (unsigned short)phi);
}
HD f3 uchar2_to_normal(uint32_t v) {
    const float PI_4 = kPi / 4.0f, PI_2 = kPi / 2.0f;
    uint32_t x = (v >> 8) & 0xff, y = v & 0xff;
    float theta = x == 63 ? PI_4 : (x == 127 ? PI_2 : (x == 191 ? 3 * PI_4 : float(x) * (1.0f / 255.0f) * kPi));
    float phi = y == 63 ? PI_2 : (y == 127 ? kPi : (y == 191 ? 3 * PI_2 : float(y) * (1.0f / 255.0f) * kPi * 2.0f));
    float sp, cp, st, ct;
#if defined(__HIP_DEVICE_COMPILE__) && (CTL_SHADE_PROBE & 2)
    sp = phi * 0.1f; cp = 1.0f - sp; st = theta * 0.2f; ct = 1.0f - st;
#elif defined(__HIP_DEVICE_COMPILE__)
    m_sincos(phi, &sp, &cp); m_sincos(theta, &st, &ct);
#else
    sp = m_sin(phi); cp = m_cos(phi); st = m_sin(theta); ct = m_cos(theta);
#endif
    return f3(st * cp, st * sp, ct);
}
// The same from a table: both angles come from bytes, so there are 256 values of (sin, cos) each — lut[x] for theta, lut[256 + y] for phi, filled on the host with the very
// function the device path above calls (normal_codec_lut, bit-identical by ctl_fmath.h's contract).  Six double-precision sincos per shaded vertex become six 8-byte loads.
HD f3 uchar2_to_normal_lut(uint32_t v, const float2* __restrict__ lut) {
    const float2 t = lut[(v >> 8) & 0xff], p = lut[256 + (v & 0xff)];   // {sin, cos}
    return f3(t.x * p.y, t.x * p.x, t.y);
}
inline void normal_codec_lut(float* out /* 512 x {sin, cos} */) {
    const float PI_4 = kPi / 4.0f, PI_2 = kPi / 2.0f;
    for (uint32_t x = 0; x < 256; x++) {
        const float theta = x == 63 ? PI_4 : (x == 127 ? PI_2 : (x == 191 ? 3 * PI_4 : float(x) * (1.0f / 255.0f) * kPi));
        const float phi = x == 63 ? PI_2 : (x == 127 ? kPi : (x == 191 ? 3 * PI_2 : float(x) * (1.0f / 255.0f) * kPi * 2.0f));
        fm::sincos(theta, &out[2 * x], &out[2 * x + 1]);
        fm::sincos(phi, &out[2 * (256 + x)], &out[2 * (256 + x) + 1]);
    }
}

} // namespace ctl
