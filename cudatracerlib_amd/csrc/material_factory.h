// material_factory.h — host-side constructors of ctl_material / ctl_texture descriptors with the parameter derivations the
// reference performs in its BSDF constructors / Update() methods (SceneTypes/BSDF_Simple.h).  Used by the scene loader.
#pragma once
#include "../../include/ctl_amd.h"
#include <cmath>
#include <cstring>

namespace ctl {

inline ctl_texture tex_const(float r, float g, float b) {
    ctl_texture t; std::memset(&t, 0, sizeof(t));
    t.type = CTL_TEX_CONSTANT; t.value[0] = r; t.value[1] = g; t.value[2] = b; t.uv_scale[0] = t.uv_scale[1] = 1.0f; t.image = 0xffffffffu;
    return t;
}
inline ctl_texture tex_const(float v) { return tex_const(v, v, v); }
inline ctl_material mat_base(uint32_t type, uint32_t combined) {
    ctl_material m; std::memset(&m, 0, sizeof(m));
    m.bsdf_type = type; m.combined_type = combined; m.two_sided = 0; m.node_light_index = 0xffffffffu;
    for (int i = 0; i < 4; i++) m.tex[i] = tex_const(0.0f);
    return m;
}
// Texture::Average() of a constant texture and Spectrum::getLuminance (Spectrum.cu:174-177).  Image / checker textures: the
// reference averages the bitmap (ImageTexture::Average samples the coarsest MIP level, Texture.cu:31-37); only level 0 crosses
// this boundary, so the caller passes the average it computed from level 0.
inline float tex_avg_luminance(const ctl_texture& t) {
    if (t.type == CTL_TEX_CHECKER) {
        const float r = (t.value[0] + t.value1[0]) * 0.5f, g = (t.value[1] + t.value1[1]) * 0.5f, b = (t.value[2] + t.value1[2]) * 0.5f;   // CheckerboardTexture::Average (Texture.h:148-151)
        return r * 0.212671f + g * 0.715160f + b * 0.072169f;
    }
    return t.value[0] * 0.212671f + t.value[1] * 0.715160f + t.value[2] * 0.072169f;
}

// FresnelHelper::fresnelDiffuseReflectance(eta, false) (Math/FresnelHelper.cu:13-62): the integral of fresnelDielectricExt(sqrt(xi), eta) over [0, 1] by the
// reference's adaptive Gauss-Lobatto quadrature GaussLobattoIntegrator(1024, 0, 1e-5f) (Math/Integrator.h:28-152), restated in its fp32 expression order —
// the value is a scene parameter (plastic's m_fdrInt / m_fdrExt) and equals the reference's bit for bit (tests/test_oracle_golden.py, against oracle/_ref).
namespace fdr_detail {
inline float fresnel_ext(float cosThetaI_, float eta) {   // FresnelHelper::fresnelDielectricExt (Math/FresnelHelper.h:27-58)
    if (eta == 1) return 0.0f;
    const float scale = (cosThetaI_ > 0) ? 1.0f / eta : eta, cosThetaTSqr = 1.0f - (1.0f - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) return 1.0f;
    const float cosThetaI = std::fabs(cosThetaI_), cosThetaT = std::sqrt(cosThetaTSqr > 0.0f ? cosThetaTSqr : 0.0f);
    const float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT), Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    return 0.5f * (Rs * Rs + Rp * Rp);
}
struct gauss_lobatto {
    float eta; size_t max_evals; float rel_error;
    float alpha() const { return std::sqrt(2.0f / 3.0f); }
    float beta() const { return 1.0f / std::sqrt(5.0f); }
    float f(float xi) const { return fresnel_ext(std::sqrt(xi), eta); }
    float step(float a, float b, float fa, float fb, float acc, size_t& evals) const {   // adaptiveGaussLobattoStep (Integrator.h:68-106)
        const float h = (b - a) / 2, m = (a + b) / 2;
        const float mll = m - alpha() * h, ml = m - beta() * h, mr = m + beta() * h, mrr = m + alpha() * h;
        const float fmll = f(mll), fml = f(ml), fm = f(m), fmr = f(mr), fmrr = f(mrr);
        const float integral2 = (h / 6) * (fa + fb + 5 * (fml + fmr));
        const float integral1 = (h / 1470) * (77 * (fa + fb) + 432 * (fmll + fmrr) + 625 * (fml + fmr) + 672 * fm);
        evals += 5;
        if (evals >= max_evals) return integral1;
        const float dist = acc + (integral1 - integral2);
        if (dist == acc || mll <= a || b <= mrr) return integral1;
        float r = step(a, mll, fa, fmll, acc, evals);   // the six sub-intervals left to right (the evaluation budget is shared)
        r = r + step(mll, ml, fmll, fml, acc, evals);
        r = r + step(ml, m, fml, fm, acc, evals);
        r = r + step(m, mr, fm, fmr, acc, evals);
        r = r + step(mr, mrr, fmr, fmrr, acc, evals);
        r = r + step(mrr, b, fmrr, fb, acc, evals);
        return r;
    }
    float tolerance(float a, float b, size_t& evals) const {   // calculateAbsTolerance (Integrator.h:108-151): absError = 0, useConvergenceEstimate = true
        const float m = (a + b) / 2, h = (b - a) / 2;
        const float x1 = 0.94288241569547971906f, x2 = 0.64185334234578130578f, x3 = 0.23638319966214988028f;
        const float y1 = f(a), y3 = f(m - alpha() * h), y5 = f(m - beta() * h), y7 = f(m), y9 = f(m + beta() * h), y11 = f(m + alpha() * h), y13 = f(b);
        const float acc = h * (0.0158271919734801831f * (y1 + y13) + 0.0942738402188500455f * (f(m - x1 * h) + f(m + x1 * h)) + 0.1550719873365853963f * (y3 + y11)
                               + 0.1888215739601824544f * (f(m - x2 * h) + f(m + x2 * h)) + 0.1997734052268585268f * (y5 + y9)
                               + 0.2249264653333395270f * (f(m - x3 * h) + f(m + x3 * h)) + 0.2426110719014077338f * y7);
        evals += 13;
        float r = 1.0;
        const float integral2 = (h / 6) * (y1 + y13 + 5 * (y5 + y9));
        const float integral1 = (h / 1470) * (77 * (y1 + y13) + 432 * (y3 + y11) + 625 * (y5 + y9) + 672 * y7);
        if (std::fabs(integral2 - acc) != 0.0) r = std::fabs(integral1 - acc) / std::fabs(integral2 - acc);
        if (r == 0.0 || r > 1.0) r = 1.0;
        float result = 3.402823466e+38f;
        if (rel_error != 0 && acc != 0) result = acc * (rel_error > 1.192092896e-07f ? rel_error : 1.192092896e-07f) / (r * 1.192092896e-07f);
        return result;
    }
    float integrate(float a, float b) const {   // Integrator.h:49-66
        size_t evals = 0;
        const float tol = tolerance(a, b, evals);
        evals += 2;
        return 1 * step(a, b, f(a), f(b), tol, evals);
    }
};
}  // namespace fdr_detail
inline float fresnel_diffuse_reflectance(float eta) { const fdr_detail::gauss_lobatto q{ eta, 1024, 1e-5f }; return q.integrate(0.0f, 1.0f); }

inline ctl_material make_diffuse(const ctl_texture& refl) { ctl_material m = mat_base(CTL_BSDF_DIFFUSE, CTL_EDiffuseReflection); m.tex[0] = refl; return m; }
inline ctl_material make_roughdiffuse(const ctl_texture& refl, const ctl_texture& alpha) {
    ctl_material m = mat_base(CTL_BSDF_ROUGHDIFFUSE, CTL_EDiffuseReflection); m.tex[0] = refl; m.tex[1] = alpha; m.u[0] = 0; return m;
}
inline ctl_material make_dielectric(float eta, const ctl_texture& refl, const ctl_texture& trans) {   // dielectric(e, r, t): Cauchy B = e, C = 0
    ctl_material m = mat_base(CTL_BSDF_DIELECTRIC, CTL_EDeltaReflection | CTL_EDeltaTransmission); m.tex[0] = trans; m.tex[1] = refl; m.f[0] = eta; m.f[1] = 0.0f; return m;
}
inline ctl_material make_thindielectric(float eta, const ctl_texture& refl, const ctl_texture& trans) {
    // the reference's constructor (BSDF_Simple.h:102-118) declares these two lobes although sample() reports ENull for the transmitted one
    ctl_material m = mat_base(CTL_BSDF_THINDIELECTRIC, CTL_EDeltaReflection | CTL_EDeltaTransmission); m.tex[0] = trans; m.tex[1] = refl; m.f[0] = eta; return m;
}
inline ctl_material make_roughdielectric(uint32_t dist, float eta, const ctl_texture& aU, const ctl_texture& aV, const ctl_texture& refl, const ctl_texture& trans) {
    ctl_material m = mat_base(CTL_BSDF_ROUGHDIELECTRIC, CTL_EGlossyReflection | CTL_EGlossyTransmission);
    m.tex[0] = trans; m.tex[1] = refl; m.tex[2] = aU; m.tex[3] = aV; m.f[0] = eta; m.f[1] = 1.0f / eta; m.u[0] = dist; m.u[1] = dist == CTL_MF_PHONG ? 0 : 1;
    return m;
}
inline ctl_material make_conductor(const float eta[3], const float k[3], const ctl_texture& refl) {
    ctl_material m = mat_base(CTL_BSDF_CONDUCTOR, CTL_EDeltaReflection); m.tex[0] = refl;
    for (int i = 0; i < 3; i++) { m.f[i] = eta[i]; m.f[3 + i] = k[i]; }
    return m;
}
inline ctl_material make_roughconductor(uint32_t dist, const float eta[3], const float k[3], const ctl_texture& aU, const ctl_texture& aV, const ctl_texture& refl) {
    ctl_material m = mat_base(CTL_BSDF_ROUGHCONDUCTOR, CTL_EGlossyReflection); m.tex[0] = refl; m.tex[1] = aU; m.tex[2] = aV;
    for (int i = 0; i < 3; i++) { m.f[i] = eta[i]; m.f[3 + i] = k[i]; }
    m.u[0] = dist; m.u[1] = dist == CTL_MF_PHONG ? 0 : 1;
    return m;
}
inline ctl_material make_plastic(float eta, const ctl_texture& diff, const ctl_texture& spec, bool nonlinear) {   // plastic::Update (BSDF_Simple.h:255-264)
    ctl_material m = mat_base(CTL_BSDF_PLASTIC, CTL_EDeltaReflection | CTL_EDiffuseReflection); m.tex[0] = diff; m.tex[1] = spec;
    const float dAvg = tex_avg_luminance(diff), sAvg = tex_avg_luminance(spec);
    m.f[0] = fresnel_diffuse_reflectance(1 / eta); m.f[1] = fresnel_diffuse_reflectance(eta); m.f[2] = eta; m.f[3] = 1.0f / (eta * eta); m.f[4] = sAvg / (dAvg + sAvg);
    m.u[0] = nonlinear ? 1 : 0;
    return m;
}
inline ctl_material make_roughplastic(uint32_t dist, float eta, const ctl_texture& alpha, const ctl_texture& diff, const ctl_texture& spec, bool nonlinear) {
    ctl_material m = mat_base(CTL_BSDF_ROUGHPLASTIC, CTL_EGlossyReflection | CTL_EDiffuseReflection); m.tex[0] = diff; m.tex[1] = spec; m.tex[2] = alpha;
    const float dAvg = tex_avg_luminance(diff), sAvg = tex_avg_luminance(spec);
    m.f[0] = eta; m.f[1] = 1.0f / (eta * eta); m.f[2] = sAvg / (dAvg + sAvg);
    m.u[0] = nonlinear ? 1 : 0; m.u[1] = dist == CTL_MF_PHONG ? 0 : 1; m.u[2] = dist;
    return m;
}
inline ctl_material make_phong(const ctl_texture& diff, const ctl_texture& spec, const ctl_texture& exponent) {
    ctl_material m = mat_base(CTL_BSDF_PHONG, CTL_EGlossyReflection | CTL_EDiffuseReflection); m.tex[0] = diff; m.tex[1] = spec; m.tex[2] = exponent;
    const float dAvg = tex_avg_luminance(diff), sAvg = tex_avg_luminance(spec);
    m.f[0] = sAvg / (dAvg + sAvg);
    return m;
}
inline ctl_material make_ward(uint32_t variant, const ctl_texture& diff, const ctl_texture& spec, const ctl_texture& aU, const ctl_texture& aV) {
    ctl_material m = mat_base(CTL_BSDF_WARD, CTL_EGlossyReflection | CTL_EDiffuseReflection); m.tex[0] = diff; m.tex[1] = spec; m.tex[2] = aU; m.tex[3] = aV;
    const float dAvg = tex_avg_luminance(diff), sAvg = tex_avg_luminance(spec);
    m.f[0] = sAvg / (dAvg + sAvg); m.u[0] = variant;
    return m;
}

// coating / roughcoating: m_specularSamplingWeight = 1 / (avg(exp(-2 thickness sigmaA)) + 1) (BSDF_Complex.h:27-29,44-47)
inline float coating_ssw(const ctl_texture& sigmaA, float thickness) {
    const float a = (expf(sigmaA.value[0] * (-2 * thickness)) + expf(sigmaA.value[1] * (-2 * thickness)) + expf(sigmaA.value[2] * (-2 * thickness))) * (1.0f / 3);
    return 1.0f / (a + 1.0f);
}
inline ctl_material make_coating(uint32_t nested_index, uint32_t nested_type, float eta, float thickness, const ctl_texture& sigmaA, const ctl_texture& spec) {
    ctl_material m = mat_base(CTL_BSDF_COATING, CTL_EDeltaReflection | nested_type); m.tex[0] = sigmaA; m.tex[1] = spec;
    m.f[0] = eta; m.f[1] = 1.0f / eta; m.f[2] = thickness; m.f[3] = coating_ssw(sigmaA, thickness); m.u[2] = nested_index;
    return m;
}
inline ctl_material make_roughcoating(uint32_t dist, uint32_t nested_index, uint32_t nested_type, float eta, float thickness, const ctl_texture& sigmaA, const ctl_texture& alpha, const ctl_texture& spec) {
    ctl_material m = mat_base(CTL_BSDF_ROUGHCOATING, CTL_EGlossyReflection | nested_type); m.tex[0] = sigmaA; m.tex[1] = spec; m.tex[2] = alpha;
    m.f[0] = eta; m.f[1] = 1.0f / eta; m.f[2] = thickness; m.f[3] = coating_ssw(sigmaA, thickness); m.u[0] = dist; m.u[1] = dist == CTL_MF_PHONG ? 0 : 1; m.u[2] = nested_index;
    return m;
}
inline ctl_material make_blend(uint32_t n0, uint32_t type0, uint32_t n1, uint32_t type1, const ctl_texture& weight) {
    ctl_material m = mat_base(CTL_BSDF_BLEND, type0 | type1); m.tex[0] = weight; m.u[2] = n0; m.u[3] = n1;
    return m;
}

// BSDF::Update() of the reference (BSDF_Simple.h plastic :255-264, roughplastic :298-304, phong :332-337, ward :371-376; BSDF_Complex.h coating :37-44,
// roughcoating :117-125): recompute the DERIVED fields of a flat material from its primary ones (eta, thickness, the textures' averages).  Models without
// derived fields are left as they are.  Returns false for an unknown bsdf_type.
inline bool material_update(ctl_material& m) {
    switch (m.bsdf_type) {
    case CTL_BSDF_PLASTIC: {
        const float eta = m.f[2], dAvg = tex_avg_luminance(m.tex[0]), sAvg = tex_avg_luminance(m.tex[1]);
        m.f[3] = 1.0f / (eta * eta); m.f[0] = fresnel_diffuse_reflectance(1 / eta); m.f[1] = fresnel_diffuse_reflectance(eta); m.f[4] = sAvg / (dAvg + sAvg); return true; }
    case CTL_BSDF_ROUGHPLASTIC: { const float eta = m.f[0], dAvg = tex_avg_luminance(m.tex[0]), sAvg = tex_avg_luminance(m.tex[1]); m.f[1] = 1.0f / (eta * eta); m.f[2] = sAvg / (dAvg + sAvg); return true; }
    case CTL_BSDF_PHONG: case CTL_BSDF_WARD: { const float dAvg = tex_avg_luminance(m.tex[0]), sAvg = tex_avg_luminance(m.tex[1]); m.f[0] = sAvg / (dAvg + sAvg); return true; }
    case CTL_BSDF_COATING: case CTL_BSDF_ROUGHCOATING: m.f[1] = 1.0f / m.f[0]; m.f[3] = coating_ssw(m.tex[0], m.f[2]); return true;
    case CTL_BSDF_ROUGHDIELECTRIC: m.f[1] = 1.0f / m.f[0]; return true;
    case CTL_BSDF_DIFFUSE: case CTL_BSDF_ROUGHDIFFUSE: case CTL_BSDF_DIELECTRIC: case CTL_BSDF_THINDIELECTRIC: case CTL_BSDF_CONDUCTOR: case CTL_BSDF_ROUGHCONDUCTOR: case CTL_BSDF_BLEND: return true;
    default: return false;
    }
}

} // namespace ctl
