// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's arithmetic for the wavefront
// path-tracing hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it;
// the product (cudatracerlib_amd/) never includes, links or calls anything under oracle/.
//
// omath.h — vectors, float4x4, Frame, warps, Fresnel, half / normal codecs.
// Every function cites the reference file:line it restates (paths relative to /root/reference).
// Compile with -ffp-contract=off: the reference's host path is plain IEEE fp32 without FMA contraction.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <algorithm>

#ifdef ORC_SHARED_MATH
#include "../cudatracerlib_amd/csrc/ctl_fmath.h"
#endif

namespace orc {

// ---- transcendental functions of the path: glibc's (the reference's CPU path, what tests/golden pins bit for bit) or, with -DORC_SHARED_MATH (liboracle_sm.so), the product's
// shared fp32 implementation (cudatracerlib_amd/csrc/ctl_fmath.h) that the HIP kernels run — the checker of the GPU parity tests then has no libm-vs-device difference left
#ifdef ORC_SHARED_MATH
inline float msin(float x) { return ctl::fm::sin(x); }  inline float mcos(float x) { return ctl::fm::cos(x); }  inline float mtan(float x) { return ctl::fm::tan(x); }
inline float macos(float x) { return ctl::fm::acos(x); }  inline float matan(float x) { return ctl::fm::atan(x); }  inline float matan2(float y, float x) { return ctl::fm::atan2(y, x); }
inline float mexp(float x) { return ctl::fm::exp(x); }  inline float mlog(float x) { return ctl::fm::log(x); }  inline float mlog2(float x) { return ctl::fm::log2(x); }  inline float mpow(float x, float y) { return ctl::fm::pow(x, y); }
#else
inline float msin(float x) { return sinf(x); }  inline float mcos(float x) { return cosf(x); }  inline float mtan(float x) { return tanf(x); }
inline float macos(float x) { return acosf(x); }  inline float matan(float x) { return atanf(x); }  inline float matan2(float y, float x) { return atan2f(y, x); }
inline float mexp(float x) { return expf(x); }  inline float mlog(float x) { return logf(x); }  inline float mlog2(float x) { return log2f(x); }  inline float mpow(float x, float y) { return powf(x, y); }
#endif


// Math/MathFunc.h:12-26
static constexpr float PI = 3.14159265358979f;
static constexpr float INV_PI = 1.0f / PI;
static constexpr float INV_TWOPI = 1.0f / (2.0f * PI);
static constexpr float INV_FOURPI = 1.0f / (4.0f * PI);
static constexpr float EPSILON = 0.000001f;
static constexpr float DeltaEpsilon = 1e-3f;

inline float fmin2(float a, float b) { return (a < b) ? a : b; }   // MathFunc.h:96 (template min)
inline float fmax2(float a, float b) { return (a > b) ? a : b; }
inline float clampf(float v, float lo, float hi) { return fmin2(fmax2(v, lo), hi); }  // MathFunc.h:170
inline float safe_sqrt(float v) { return std::sqrt(fmax2(0.0f, v)); }                 // MathFunc.h:112
inline float safe_acos(float v) { return macos(fmin2(1.0f, fmax2(-1.0f, v))); }      // MathFunc.h:108
inline float fracf(float f) { return f - floorf(f); }                                 // MathFunc.h:138
inline int floor2int(float v) { return (int)floorf(v); }                              // MathFunc.h:143
inline float int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float copysign_bits(float a, float b) {                                        // MathFunc.h:64-67
    return int_as_float((float_as_int(b) & 0x80000000) | (float_as_int(a) & ~0x80000000));
}


struct V2 { float x, y; };
inline V2 operator*(V2 a, float s) { return V2{ a.x * s, a.y * s }; }
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a) : x(a), y(a), z(a) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return (&x)[i]; }
    float& operator[](int i) { return (&x)[i]; }
};
// Math/Vector.h:84-118 — component-wise, evaluated in index order
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
// Vector.h:101 dot: r = 0; r += a[i]*b[i]
inline float dot(V3 a, V3 b) { float r = 0.0f; r += a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }
inline float absdot(V3 a, V3 b) { return fabsf(dot(a, b)); }
inline V3 cross(V3 a, V3 v) { return V3(a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x); } // Vector.h:329
inline float lenSqr(V3 a) { float r = 0.0f; r += a.x * a.x; r += a.y * a.y; r += a.z * a.z; return r; }              // Vector.h:46
inline float length(V3 a) { return std::sqrt(lenSqr(a)); }
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }                                                       // Vector.h:369-371 (v * rcp(len))
inline float vmax(V3 a) { float r = a.x; r = fmax2(r, a.y); r = fmax2(r, a.z); return r; }                          // Vector.h:50
inline bool isZero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }

// Spectrum == RGB triple (SPECTRUM_SAMPLES 3, Math/Spectrum.h:10); toLinearRGB is the identity (Spectrum.cu:174-184)
typedef V3 Spec;
// TSpectrum::operator/(Scalar) and operator/=(Scalar) (Math/Spectrum.h:122-128, :150-155) multiply by the reciprocal; Vec3f / float (Math/Vector.h:88) divides.
// Spec shares V3's operators, so every Spectrum-by-scalar division of the reference is written sdiv() here (pinned through KernelMIPMap::evalEWA, mipmap.npz).
inline V3 sdiv(V3 s, float f) { const float recip = 1.0f / f; return V3(s.x * recip, s.y * recip, s.z * recip); }

// ---------------------------------------------------------------- float4x4 (Math/float4x4.h)
struct M44 {
    float d[16];
    float operator()(int i, int j) const { return d[i * 4 + j]; }
    float& operator()(int i, int j) { return d[i * 4 + j]; }
    static M44 identity() { M44 m; for (int i = 0; i < 16; i++) m.d[i] = (i % 5 == 0) ? 1.0f : 0.0f; return m; }
};
// float4x4.h:373-381: dot(row(i), col(j)) as a 4-vector dot (r=0; r+=...)
inline M44 mul(const M44& l, const M44& r) {
    M44 o;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        float s = 0.0f; for (int k = 0; k < 4; k++) s += l(i, k) * r(k, j); o(i, j) = s;
    }
    return o;
}
// float4x4.h:383-391,402-412: M * (p,1) then divide by w ; direction uses w = 0
inline V3 transformPoint(const M44& m, V3 p) {
    float r[4];
    for (int i = 0; i < 4; i++) { float s = 0.0f; s += m(i, 0) * p.x; s += m(i, 1) * p.y; s += m(i, 2) * p.z; s += m(i, 3) * 1.0f; r[i] = s; }
    return V3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}
inline V3 transformDir(const M44& m, V3 p) {
    float r[3];
    for (int i = 0; i < 3; i++) { float s = 0.0f; s += m(i, 0) * p.x; s += m(i, 1) * p.y; s += m(i, 2) * p.z; s += m(i, 3) * 0.0f; r[i] = s; }
    return V3(r[0], r[1], r[2]);
}
// float4x4.h:132-193 (cofactor inverse, same expression order)
inline M44 inverse(const M44& Q) {
    float m00 = Q(0, 0), m01 = Q(0, 1), m02 = Q(0, 2), m03 = Q(0, 3);
    float m10 = Q(1, 0), m11 = Q(1, 1), m12 = Q(1, 2), m13 = Q(1, 3);
    float m20 = Q(2, 0), m21 = Q(2, 1), m22 = Q(2, 2), m23 = Q(2, 3);
    float m30 = Q(3, 0), m31 = Q(3, 1), m32 = Q(3, 2), m33 = Q(3, 3);
    float v0 = m20 * m31 - m21 * m30, v1 = m20 * m32 - m22 * m30, v2 = m20 * m33 - m23 * m30;
    float v3 = m21 * m32 - m22 * m31, v4 = m21 * m33 - m23 * m31, v5 = m22 * m33 - m23 * m32;
    float t00 = +(v5 * m11 - v4 * m12 + v3 * m13), t10 = -(v5 * m10 - v2 * m12 + v1 * m13);
    float t20 = +(v4 * m10 - v2 * m11 + v0 * m13), t30 = -(v3 * m10 - v1 * m11 + v0 * m12);
    float invDet = 1 / (t00 * m00 + t10 * m01 + t20 * m02 + t30 * m03);
    float d00 = t00 * invDet, d10 = t10 * invDet, d20 = t20 * invDet, d30 = t30 * invDet;
    float d01 = -(v5 * m01 - v4 * m02 + v3 * m03) * invDet, d11 = +(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d21 = -(v4 * m00 - v2 * m01 + v0 * m03) * invDet, d31 = +(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    v0 = m10 * m31 - m11 * m30; v1 = m10 * m32 - m12 * m30; v2 = m10 * m33 - m13 * m30;
    v3 = m11 * m32 - m12 * m31; v4 = m11 * m33 - m13 * m31; v5 = m12 * m33 - m13 * m32;
    float d02 = +(v5 * m01 - v4 * m02 + v3 * m03) * invDet, d12 = -(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d22 = +(v4 * m00 - v2 * m01 + v0 * m03) * invDet, d32 = -(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    v0 = m21 * m10 - m20 * m11; v1 = m22 * m10 - m20 * m12; v2 = m23 * m10 - m20 * m13;
    v3 = m22 * m11 - m21 * m12; v4 = m23 * m11 - m21 * m13; v5 = m23 * m12 - m22 * m13;
    float d03 = -(v5 * m01 - v4 * m02 + v3 * m03) * invDet, d13 = +(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d23 = -(v4 * m00 - v2 * m01 + v0 * m03) * invDet, d33 = +(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    M44 r;
    float v[16] = { d00, d01, d02, d03, d10, d11, d12, d13, d20, d21, d22, d23, d30, d31, d32, d33 };
    std::memcpy(r.d, v, sizeof(v));
    return r;
}
// float4x4.h:229-244
inline M44 perspective(float fov, float clipNear, float clipFar) {
    float recip = 1.0f / (clipFar - clipNear);
    float cot = 1.0f / tanf(fov / 2.0f);   // set-up time (the product computes the camera matrices on the host, with libm, in both oracle builds)
    M44 m; std::memset(m.d, 0, sizeof(m.d));
    m(0, 0) = cot; m(1, 1) = cot; m(2, 2) = clipFar * recip; m(2, 3) = -clipNear * clipFar * recip; m(3, 2) = 1;
    return m;
}
inline M44 scaleM(V3 s) { M44 m = M44::identity(); m(0, 0) = s.x; m(1, 1) = s.y; m(2, 2) = s.z; return m; }
inline M44 translateM(V3 t) { M44 m = M44::identity(); m(0, 3) = t.x; m(1, 3) = t.y; m(2, 3) = t.z; return m; }

// ---------------------------------------------------------------- Frame (Math/Frame.h)
// Frame.h:9-22
inline void coordinateSystem(V3 a, V3& s, V3& t) {
    if (fabsf(a.x) > fabsf(a.y)) {
        float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
        t = V3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
        t = V3(0.0f, a.z * invLen, -a.y * invLen);
    }
    s = normalize(cross(t, a));
}
struct Frame {
    V3 s, t, n;
    Frame() {}
    Frame(V3 s_, V3 t_, V3 n_) : s(s_), t(t_), n(n_) {}
    explicit Frame(V3 n_) : n(n_) { coordinateSystem(n, s, t); }
    V3 toLocal(V3 v) const { return V3(dot(v, s), dot(v, t), dot(v, n)); }   // Frame.h:37-39
    V3 toWorld(V3 v) const { return s * v.x + t * v.y + n * v.z; }            // Frame.h:40-42
    static float cosTheta(V3 v) { return v.z; }
    static float sinTheta2(V3 v) { return 1.0f - v.z * v.z; }
    static float sinTheta(V3 v) { float t = sinTheta2(v); if (t <= 0.0f) return 0.0f; return std::sqrt(t); }
    static float tanTheta(V3 v) { float t = 1 - v.z * v.z; if (t <= 0.0f) return 0.0f; return std::sqrt(t) / v.z; }
    static float tanTheta2(V3 v) { float t = 1 - v.z * v.z; if (t <= 0.0f) return 0.0f; return t / (v.z * v.z); }
    static float sinPhi(V3 v) { float st = sinTheta(v); if (st == 0.0f) return 1.0f; return clampf(v.y / st, -1.0f, 1.0f); }
    static float cosPhi(V3 v) { float st = sinTheta(v); if (st == 0.0f) return 1.0f; return clampf(v.x / st, -1.0f, 1.0f); }
    static float sinPhi2(V3 v) { return clampf(v.y * v.y / sinTheta2(v), 0.0f, 1.0f); }
    static float cosPhi2(V3 v) { return clampf(v.x * v.x / sinTheta2(v), 0.0f, 1.0f); }
    static V3 reflect(V3 wi) { return V3(-wi.x, -wi.y, wi.z); }                                  // Frame.h:134-136
    static V3 refract(V3 wi, float cosThetaT, float eta, float invEta) {                         // Frame.h:144-152 (normalized overload)
        float scale = -(cosThetaT < 0 ? invEta : eta);
        return normalize(V3(scale * wi.x, scale * wi.y, cosThetaT));
    }
};

// ---------------------------------------------------------------- Warp (Math/Warp.h)
inline V2 squareToUniformDiskConcentric(V2 sample) {   // Warp.h:104-127
    float r1 = 2.0f * sample.x - 1.0f, r2 = 2.0f * sample.y - 1.0f;
    float phi, r;
    if (r1 == 0 && r2 == 0) { r = phi = 0; }
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (PI / 4.0f) * (r2 / r1); }
    else { r = r2; phi = (PI / 2.0f) - (r1 / r2) * (PI / 4.0f); }
    float cosPhi = mcos(phi), sinPhi = msin(phi);
    return V2{ r * cosPhi, r * sinPhi };
}
inline V3 squareToCosineHemisphere(V2 sample) {         // Warp.h:61-66
    V2 p = squareToUniformDiskConcentric(sample);
    float z = std::sqrt(1.0f - p.x * p.x - p.y * p.y);
    return V3(p.x, p.y, z);
}
inline float squareToCosineHemispherePdf(V3 d) { return INV_PI * Frame::cosTheta(d); }   // Warp.h:68-71
inline V2 squareToUniformTriangle(V2 sample) {          // Warp.h:160-164
    float a = std::sqrt(1.0f - sample.x);
    return V2{ 1 - a, a * sample.y };
}
inline V3 squareToUniformSphere(V2 sample) {            // Warp.h:28-35
    float z = 1.0f - 2.0f * sample.y;
    float r = std::sqrt(1.0f - z * z);
    float a = 2.0f * PI * sample.x;
    return V3(r * mcos(a), r * msin(a), z);
}

// ---------------------------------------------------------------- MonteCarlo / Fresnel
inline float powerHeuristic(int nf, float fPdf, int ng, float gPdf) {   // Math/MonteCarlo.h:29-33
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
// Math/FresnelHelper.h:27-58
inline float fresnelDielectricExt(float cosThetaI_, float& cosThetaT_, float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    float scale = (cosThetaI_ > 0) ? 1.0f / eta : eta, cosThetaTSqr = 1.0f - (1.0f - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = fabsf(cosThetaI_);
    float cosThetaT = safe_sqrt(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
inline float fresnelDielectricExt(float cosThetaI, float eta) { float t; return fresnelDielectricExt(cosThetaI, t, eta); }
// Math/FresnelHelper.h:119-146 (Spectrum overload, per channel; safe_sqrt)
inline Spec fresnelConductorExact(float cosThetaI, Spec eta, Spec k) {
    float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    Spec r;
    for (int c = 0; c < 3; c++) {
        float e = eta[c], kk = k[c];
        float temp1 = e * e - kk * kk - sinThetaI2;
        float a2pb2 = safe_sqrt(temp1 * temp1 + kk * kk * e * e * 4);
        float a = safe_sqrt((a2pb2 + temp1) * 0.5f);
        float term1 = a2pb2 + cosThetaI2, term2 = a * (2 * cosThetaI);
        float Rs2 = (term1 - term2) / (term1 + term2);
        float term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
        float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
        r[c] = 0.5f * (Rp2 + Rs2);
    }
    return r;
}

// ---------------------------------------------------------------- half (Math/half.h)
// half.h:20-70 float -> half, IEEE round-to-nearest-even (the host branch; identical to __float2half_rn)
inline uint16_t floatToHalf(float f) {
    uint32_t ia; std::memcpy(&ia, &f, 4);
    uint16_t ir = (ia >> 16) & 0x8000;
    if ((ia & 0x7f800000) == 0x7f800000) {
        if ((ia & 0x7fffffff) == 0x7f800000) ir |= 0x7c00; else ir = 0x7fff;
    } else if ((ia & 0x7f800000) >= 0x33000000) {
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
// half -> float.  host_quirk = false: IEEE (the device branch, __half2float, half.h:74-75).
// host_quirk = true: the reference's host branch (half.h:76-83), which maps zero/denormals to 2^-15-scale values.
inline float halfToFloat(uint16_t val, bool host_quirk = false) {
    if (host_quirk) {
        int fltInt32 = ((val & 0x8000) << 16);
        fltInt32 |= ((val & 0x7fff) << 13) + 0x38000000;
        float r; std::memcpy(&r, &fltInt32, 4); return r;
    }
    uint32_t sign = (uint32_t)(val & 0x8000) << 16, exp = (val >> 10) & 0x1f, man = val & 0x3ff, out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else { int e = -1; do { e++; man <<= 1; } while ((man & 0x400) == 0); out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13); }
    } else if (exp == 31) out = sign | 0x7f800000 | (man << 13);
    else out = sign | ((exp + 112) << 23) | (man << 13);
    float r; std::memcpy(&r, &out, 4); return r;
}

// ---------------------------------------------------------------- Compression (Math/Compression.h)
// Compression.h:12-18
inline uint16_t normalToUchar2(V3 v) {
    float theta = (macos(v.z) * (255.0f / PI));
    float phi = (matan2(v.y, v.x) * (255.0f / (2.0f * PI)));
    phi = phi < 0 ? (phi + 255) : phi;
    return (uint16_t)(((unsigned short)theta << 8) | (unsigned short)phi);
}
// Compression.h:20-31
inline V3 uchar2ToNormal(uint16_t v) {
    const float PI_4 = PI / 4.0f, PI_2 = PI / 2.0f;
    unsigned char x = v >> 8, y = v & 0xff;
    float theta = x == 63 ? PI_4 : (x == 127 ? PI_2 : (x == 191 ? 3 * PI_4 : float(x) * (1.0f / 255.0f) * PI));
    float phi = y == 63 ? PI_2 : (y == 127 ? PI : (y == 191 ? 3 * PI_2 : float(y) * (1.0f / 255.0f) * PI * 2.0f));
    float sinphi = msin(phi), cosphi = mcos(phi), sintheta = msin(theta), costheta = mcos(theta);
    return V3(sintheta * cosphi, sintheta * sinphi, costheta);
}

} // namespace orc
