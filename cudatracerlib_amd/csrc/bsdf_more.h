// bsdf_more.h — thindielectric, roughdielectric, plastic, phong for the shade kernel (included by shading.h).
// Behaviour per function: SceneTypes/BSDF_Simple.cu lines cited at each case.
#pragma once
#include "bsdf_rough.h"

namespace ctl {

__device__ __forceinline__ f3 refract_about(f3 wi, f3 n, float eta, float cosThetaT) {   // FresnelHelper.h:149-155
    if (cosThetaT < 0) eta = 1.0f / eta;
    return n * (dot(wi, n) * eta + cosThetaT) - wi * eta;
}
__device__ __forceinline__ float signum1(float v) { return copysign_bits(1.0f, v); }
__device__ __forceinline__ f3 plastic_diffuse(const ctl_material& M, const diff_geom& dg) {
    const f3 d = tex_eval(M.tex[0], dg);
    return M.u[0] ? d / (f3(1.0f) - d * M.f[0]) : sdiv(d, 1 - M.f[0]);
}
__device__ __forceinline__ microfacet rough_dielectric_distr(const ctl_material& M, const diff_geom& dg, float cos_wi, bool scaled) {
    microfacet d((int)M.u[0], avg3(tex_eval(M.tex[2], dg)), avg3(tex_eval(M.tex[3], dg)), M.u[1] != 0);
    if (scaled && !d.vis) { const float sc = 1.2f - 0.2f * sqrtf(fabsf(cos_wi)); d.scale_alpha(sc); }   // scaleAlpha (MicrofacetDistribution.h:60-66)
    return d;
}

__device__ f3 phong_f(const ctl_material& M, const bsdf_rec& b) {   // BSDF_Simple.cu:1113-1139
    if (cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0) return f3(0.0f);
    f3 result(0.0f);
    if (b.type_mask & CTL_EGlossyReflection) {
        const float alpha = dot(b.wo, reflect_local(b.wi)), e = avg3(tex_eval(M.tex[2], b.dg));
        if (alpha > 0.0f) result = result + tex_eval(M.tex[1], b.dg) * ((e + 2) * kInvTwoPi * m_pow(alpha, e));
    }
    if (b.type_mask & CTL_EDiffuseReflection) result = result + tex_eval(M.tex[0], b.dg) * kInvPi;
    return result * cos_theta(b.wo);
}
__device__ float phong_pdf(const ctl_material& M, const bsdf_rec& b) {   // BSDF_Simple.cu:1141-1171
    if (cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0) return 0.0f;
    const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
    float dp = 0.0f, sp = 0.0f;
    if (hd) dp = kInvPi * cos_theta(b.wo);
    if (hs) { const float alpha = dot(b.wo, reflect_local(b.wi)), e = avg3(tex_eval(M.tex[2], b.dg)); if (alpha > 0) sp = m_pow(alpha, e) * (e + 1.0f) / (2.0f * kPi); }
    if (hd && hs) return M.f[0] * sp + (1 - M.f[0]) * dp;
    return hd ? dp : (hs ? sp : 0.0f);
}

__device__ f3 bsdf_more_sample(const ctl_material& M, bsdf_rec& b, float& pdf, f2 smp) {
    switch (M.bsdf_type) {
    case CTL_BSDF_THINDIELECTRIC: { if (!CTL_HAS_MODEL(CTL_BSDF_THINDIELECTRIC)) return f3(0.0f);   // BSDF_Simple.cu:330-371
        const bool sr = (b.type_mask & CTL_EDeltaReflection) != 0, st = (b.type_mask & CTL_ENull) != 0;
        float ct; float R = fresnel_dielectric_ext(fabsf(cos_theta(b.wi)), ct, M.f[0]); const float T = 1 - R;
        if (R < 1) R += T * T * R / (1 - R * R);
        b.eta = 1.0f;
        if (st && sr) {
            if (smp.x <= R) { b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); pdf = R; return tex_eval(M.tex[1], b.dg); }
            b.sampled_type = CTL_ENull; b.wo = -b.wi; pdf = 1 - R; return tex_eval(M.tex[0], b.dg);
        } else if (sr) { b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); pdf = 1.0f; return tex_eval(M.tex[1], b.dg) * R; }
        else if (st) { b.sampled_type = CTL_ENull; b.wo = -b.wi; pdf = 1.0f; return tex_eval(M.tex[0], b.dg) * (1 - R); }
        return f3(0.0f);
    }
    case CTL_BSDF_ROUGHDIELECTRIC: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHDIELECTRIC)) return f3(0.0f);   // BSDF_Simple.cu:503-615
        const bool hr = (b.type_mask & CTL_EGlossyReflection) != 0, ht = (b.type_mask & CTL_EGlossyTransmission) != 0;
        bool sample_refl = hr;
        if (!hr && !ht) return f3(0.0f);
        const float eta = M.f[0], inv_eta = M.f[1];
        const microfacet distr = rough_dielectric_distr(M, b.dg, 0.0f, false), sdistr = rough_dielectric_distr(M, b.dg, cos_theta(b.wi), true);
        float mpdf;
        const float sign = signum1(cos_theta(b.wi));
        const f3 m = sdistr.sample(sign < 0 ? -b.wi : b.wi, smp, mpdf);
        if (mpdf == 0) return f3(0.0f);
        pdf = mpdf;
        float cosThetaT;
        const float F = fresnel_dielectric_ext(dot(b.wi, m), cosThetaT, eta);
        f3 weight(1.0f);
        const float sample_z = (float)(uint32_t)(int)(smp.x * 10) / 10.0f;   // MonteCarlo::sampleReuse(10, sample.x, slot) (MonteCarlo.cu:16-20)
        if (hr && ht) { if (sample_z > F) { sample_refl = false; pdf *= 1 - F; } else pdf *= F; }
        else weight = weight * (hr ? F : (1 - F));
        float dwh_dwo;
        if (sample_refl) {
            b.wo = reflect_about(b.wi, m); b.eta = 1.0f; b.sampled_type = CTL_EGlossyReflection;
            if (cos_theta(b.wi) * cos_theta(b.wo) <= 0) return f3(0.0f);
            weight = weight * tex_eval(M.tex[1], b.dg);
            dwh_dwo = 1.0f / (4.0f * dot(b.wo, m));
        } else {
            if (cosThetaT == 0) return f3(0.0f);
            b.wo = normalize(refract_about(b.wi, m, eta, cosThetaT));
            b.eta = cosThetaT < 0 ? eta : inv_eta; b.sampled_type = CTL_EGlossyTransmission;
            if (cos_theta(b.wi) * cos_theta(b.wo) >= 0) return f3(0.0f);
            const float factor = (cosThetaT < 0 ? inv_eta : eta);
            weight = weight * (tex_eval(M.tex[0], b.dg) * (factor * factor));
            const float sd = dot(b.wi, m) + b.eta * dot(b.wo, m);
            dwh_dwo = (b.eta * b.eta * dot(b.wo, m)) / (sd * sd);
        }
        if (distr.vis) weight = weight * distr.smith_g1(b.wo, m);
        else weight = weight * fabsf(distr.eval(m) * distr.G(b.wi, b.wo, m) * dot(b.wi, m) / (mpdf * cos_theta(b.wi)));
        pdf *= fabsf(dwh_dwo);
        return weight;
    }
    case CTL_BSDF_PLASTIC: { if (!CTL_HAS_MODEL(CTL_BSDF_PLASTIC)) return f3(0.0f);   // BSDF_Simple.cu:765-826
        const bool hs = (b.type_mask & CTL_EDeltaReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
        if ((!hd && !hs) || cos_theta(b.wi) <= 0) return f3(0.0f);
        float ct; const float Fi = fresnel_dielectric_ext(cos_theta(b.wi), ct, M.f[2]);
        b.eta = 1.0f;
        if (hd && hs) {
            const float ps = (Fi * M.f[4]) / (Fi * M.f[4] + (1 - Fi) * (1 - M.f[4]));
            if (smp.x < ps) { b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); pdf = ps; return sdiv(tex_eval(M.tex[1], b.dg) * Fi, ps); }
            b.sampled_type = CTL_EDiffuseReflection;
            b.wo = square_to_cosine_hemisphere(f2{ (smp.x - ps) / (1 - ps), smp.y });
            const float Fo = fresnel_dielectric_ext(cos_theta(b.wo), ct, M.f[2]);
            pdf = (1 - ps) * (kInvPi * cos_theta(b.wo));
            return plastic_diffuse(M, b.dg) * (M.f[3] * (1 - Fi) * (1 - Fo) / (1 - ps));
        } else if (hs) { b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); pdf = 1; return tex_eval(M.tex[1], b.dg) * Fi; }
        b.sampled_type = CTL_EDiffuseReflection; b.wo = square_to_cosine_hemisphere(smp);
        const float Fo = fresnel_dielectric_ext(cos_theta(b.wo), ct, M.f[2]);
        pdf = kInvPi * cos_theta(b.wo);
        return plastic_diffuse(M, b.dg) * (M.f[3] * (1 - Fi) * (1 - Fo));
    }
    case CTL_BSDF_PHONG: { if (!CTL_HAS_MODEL(CTL_BSDF_PHONG)) return f3(0.0f);   // BSDF_Simple.cu:1059-1111
        const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
        if (!hs && !hd) return f3(0.0f);
        bool spec = hs; const float w = M.f[0];
        if (hd && hs) { if (smp.x <= w) smp.x /= w; else { smp.x = (smp.x - w) / (1 - w); spec = false; } }
        if (spec) {
            const f3 R = reflect_local(b.wi);
            const float e = avg3(tex_eval(M.tex[2], b.dg));
            const float sinA = sqrtf(1 - m_pow(smp.y, 2 / (e + 1))), cosA = m_pow(smp.y, 1 / (e + 1)), phi = (2.0f * kPi) * smp.x;
            frame fr; fr.n = R; coordinate_system(R, fr.s, fr.t);
            b.wo = normalize(fr.to_world(f3(sinA * m_cos(phi), sinA * m_sin(phi), cosA))); b.sampled_type = CTL_EGlossyReflection;
            if (cos_theta(b.wo) <= 0) return f3(0.0f);
        } else { b.wo = square_to_cosine_hemisphere(smp); b.sampled_type = CTL_EDiffuseReflection; }
        b.eta = 1.0f;
        pdf = phong_pdf(M, b);
        if (pdf == 0) return f3(0.0f);
        return sdiv(phong_f(M, b), pdf);
    }
    default: return bsdf_rough_sample(M, b, pdf, smp);
    }
}

__device__ f3 bsdf_more_f(const ctl_material& M, const bsdf_rec& b) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIELECTRIC: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHDIELECTRIC)) return f3(0.0f);   // BSDF_Simple.cu:436-501
        const float eta_ = M.f[0], inv_eta = M.f[1];
        const bool refl = cos_theta(b.wi) * cos_theta(b.wo) > 0;
        f3 H;
        if (refl) { if (!(b.type_mask & CTL_EGlossyReflection)) return f3(0.0f); H = normalize(b.wo + b.wi); }
        else { if (!(b.type_mask & CTL_EGlossyTransmission)) return f3(0.0f); const float e = cos_theta(b.wi) > 0 ? eta_ : inv_eta; H = normalize(b.wi + b.wo * e); }
        H = H * signum1(cos_theta(H));
        const microfacet distr = rough_dielectric_distr(M, b.dg, 0.0f, false);
        const float D = distr.eval(H);
        if (D == 0) return f3(0.0f);
        float ct; const float F = fresnel_dielectric_ext(dot(b.wi, H), ct, eta_);
        const float G = distr.G(b.wi, b.wo, H);
        if (refl) { const float value = F * D * G / (4.0f * fabsf(cos_theta(b.wi))); return tex_eval(M.tex[1], b.dg) * value; }
        const float e = cos_theta(b.wi) > 0.0f ? eta_ : inv_eta;
        const float sd = dot(b.wi, H) + e * dot(b.wo, H);
        const float value = ((1 - F) * D * G * e * e * dot(b.wi, H) * dot(b.wo, H)) / (cos_theta(b.wi) * sd * sd);
        const float factor = (cos_theta(b.wi) > 0 ? inv_eta : eta_);
        return tex_eval(M.tex[0], b.dg) * fabsf(value * factor * factor);
    }
    case CTL_BSDF_PLASTIC: { if (!CTL_HAS_MODEL(CTL_BSDF_PLASTIC)) return f3(0.0f);   // BSDF_Simple.cu:828-858 — solid-angle measure: the diffuse lobe
        if (!(b.type_mask & CTL_EDiffuseReflection) || cos_theta(b.wo) <= 0 || cos_theta(b.wi) <= 0) return f3(0.0f);
        float ct; const float Fi = fresnel_dielectric_ext(cos_theta(b.wi), ct, M.f[2]), Fo = fresnel_dielectric_ext(cos_theta(b.wo), ct, M.f[2]);
        return plastic_diffuse(M, b.dg) * ((kInvPi * cos_theta(b.wo)) * M.f[3] * (1 - Fi) * (1 - Fo));
    }
    case CTL_BSDF_PHONG: return CTL_HAS_MODEL(CTL_BSDF_PHONG) ? phong_f(M, b) : f3(0.0f);
    default: return bsdf_rough_f(M, b);
    }
}

__device__ float bsdf_more_pdf(const ctl_material& M, const bsdf_rec& b) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIELECTRIC: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHDIELECTRIC)) return 0.0f;   // BSDF_Simple.cu:373-434
        const float eta_ = M.f[0], inv_eta = M.f[1];
        const bool hr = (b.type_mask & CTL_EGlossyReflection) != 0, ht = (b.type_mask & CTL_EGlossyTransmission) != 0, refl = cos_theta(b.wi) * cos_theta(b.wo) > 0;
        f3 H; float dwh_dwo;
        if (refl) { if (!hr) return 0.0f; H = normalize(b.wo + b.wi); dwh_dwo = 1.0f / (4.0f * dot(b.wo, H)); }
        else {
            if (!ht) return 0.0f;
            const float e = cos_theta(b.wi) > 0 ? eta_ : inv_eta;
            H = normalize(b.wi + b.wo * e);
            const float sd = dot(b.wi, H) + e * dot(b.wo, H);
            dwh_dwo = (e * e * dot(b.wo, H)) / (sd * sd);
        }
        H = H * signum1(cos_theta(H));
        const microfacet sdistr = rough_dielectric_distr(M, b.dg, cos_theta(b.wi), true);
        const float sign = signum1(cos_theta(b.wi));
        float prob = sdistr.pdf(sign < 0 ? -b.wi : b.wi, H);
        if (ht && hr) { float ct; const float F = fresnel_dielectric_ext(dot(b.wi, H), ct, eta_); prob *= refl ? F : (1 - F); }
        return fabsf(prob * dwh_dwo);
    }
    case CTL_BSDF_PLASTIC: { if (!CTL_HAS_MODEL(CTL_BSDF_PLASTIC)) return 0.0f;   // BSDF_Simple.cu:860-888
        const bool hs = (b.type_mask & CTL_EDeltaReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
        if (cos_theta(b.wo) <= 0 || cos_theta(b.wi) <= 0 || !hd) return 0.0f;
        float ps = hs ? 1.0f : 0.0f;
        if (hs) { float ct; const float Fi = fresnel_dielectric_ext(cos_theta(b.wi), ct, M.f[2]); ps = (Fi * M.f[4]) / (Fi * M.f[4] + (1 - Fi) * (1 - M.f[4])); }
        return (kInvPi * cos_theta(b.wo)) * (1 - ps);
    }
    case CTL_BSDF_PHONG: return CTL_HAS_MODEL(CTL_BSDF_PHONG) ? phong_pdf(M, b) : 0.0f;
    default: return bsdf_rough_pdf(M, b);
    }
}

} // namespace ctl
