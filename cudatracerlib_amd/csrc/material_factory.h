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

// FresnelHelper::fresnelDiffuseReflectance(eta, false) (Math/FresnelHelper.cu:13-60): integral of F(sqrt(xi), eta) over [0, 1],
// evaluated in double with the substitution xi = c^2 (the reference integrates adaptively to 1e-5)
inline float fresnel_diffuse_reflectance(float eta_f) {
    const double eta = eta_f;
    if (eta == 1.0) return 0.0f;
    const int N = 200000; double acc = 0.0, prev = 0.0;
    for (int i = 0; i <= N; i++) {
        const double c = (double)i / N;
        const double ct2 = 1.0 - (1.0 - c * c) / (eta * eta);
        double F;
        if (ct2 <= 0) F = 1.0;
        else { const double ct = std::sqrt(ct2), rs = (c - eta * ct) / (c + eta * ct), rp = (eta * c - ct) / (eta * c + ct); F = 0.5 * (rs * rs + rp * rp); }
        const double y = F * 2.0 * c;
        if (i > 0) acc += (y + prev) * 0.5 / N;
        prev = y;
    }
    return (float)acc;
}

inline ctl_material make_diffuse(const ctl_texture& refl) { ctl_material m = mat_base(CTL_BSDF_DIFFUSE, CTL_EDiffuseReflection); m.tex[0] = refl; return m; }
inline ctl_material make_roughdiffuse(const ctl_texture& refl, const ctl_texture& alpha) {
    ctl_material m = mat_base(CTL_BSDF_ROUGHDIFFUSE, CTL_EDiffuseReflection); m.tex[0] = refl; m.tex[1] = alpha; m.u[0] = 0; return m;
}
inline ctl_material make_dielectric(float eta, const ctl_texture& refl, const ctl_texture& trans) {   // dielectric(e, r, t): Cauchy B = e, C = 0
    ctl_material m = mat_base(CTL_BSDF_DIELECTRIC, CTL_EDeltaReflection | CTL_EDeltaTransmission); m.tex[0] = trans; m.tex[1] = refl; m.f[0] = eta; m.f[1] = 0.0f; return m;
}
inline ctl_material make_thindielectric(float eta, const ctl_texture& refl, const ctl_texture& trans) {
    ctl_material m = mat_base(CTL_BSDF_THINDIELECTRIC, CTL_EDeltaReflection | CTL_ENull); m.tex[0] = trans; m.tex[1] = refl; m.f[0] = eta; return m;
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

} // namespace ctl
