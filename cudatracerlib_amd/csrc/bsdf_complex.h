// bsdf_complex.h — the nesting BSDFs coating, roughcoating, blend (SceneTypes/BSDF_Complex.cu:6-385) and the discrete-measure
// f / pdf of the delta models they may wrap (BSDF_Simple.cu dielectric :226-277, thindielectric :279-328, conductor :632-660,
// plastic :828-888).  A nested BSDF is another entry of the scene's material array, named by its absolute index
// (coating / roughcoating: u[2]; blend: u[2], u[3]).  Included by shading.h after the simple dispatch (bsdf_sample / bsdf_f /
// bsdf_pdf); only the full shade-kernel build compiles it in (CTL_SHADE_FEATURES & 16).
#pragma once

namespace ctl {

enum { kMeasSolidAngle = 1, kMeasDiscrete = 4 };
__device__ __forceinline__ int bsdf_measure(uint32_t t) { return (t & kESmooth) ? kMeasSolidAngle : ((t & kEDelta) ? kMeasDiscrete : kMeasSolidAngle); }   // BSDF::getMeasure (BSDF.h:66-80)
__device__ __forceinline__ f3 exp3(f3 s) { return f3(m_exp(s.x), m_exp(s.y), m_exp(s.z)); }

__device__ f3 bsdf_f_discrete(const ctl_material& M, const bsdf_rec& b) {
    const float kDeltaEps = 1e-3f;
    switch (M.bsdf_type) {
    case CTL_BSDF_DIELECTRIC: {
        const bool sr = (b.type_mask & CTL_EDeltaReflection) != 0, st = (b.type_mask & CTL_EDeltaTransmission) != 0;
        float cosThetaT; const float eta = M.f[0] + M.f[1] / (600 / 1e3f), invEta = 1.0f / eta;
        const float F = fresnel_dielectric_ext(cos_theta(b.wi), cosThetaT, eta);
        if (cos_theta(b.wi) * cos_theta(b.wo) >= 0) {
            if (!sr || fabsf(dot(reflect_local(b.wi), b.wo) - 1) > kDeltaEps) return f3(0.0f);
            return tex_eval(M.tex[1], b.dg) * F;
        }
        if (!st || fabsf(dot(refract_local(b.wi, cosThetaT, eta, invEta), b.wo) - 1) > kDeltaEps) return f3(0.0f);
        const float factor = cosThetaT < 0 ? invEta : eta;
        return f3(1.0f) * tex_eval(M.tex[0], b.dg) * factor * factor * (1 - F);
    }
    case CTL_BSDF_THINDIELECTRIC: {
        const bool sr = (b.type_mask & CTL_EDeltaReflection) != 0, st = (b.type_mask & CTL_ENull) != 0;
        float ct; float R = fresnel_dielectric_ext(fabsf(cos_theta(b.wi)), ct, M.f[0]); const float T = 1 - R;
        if (R < 1) R += T * T * R / (1 - R * R);
        if (cos_theta(b.wi) * cos_theta(b.wo) >= 0) {
            if (!sr || fabsf(dot(reflect_local(b.wi), b.wo) - 1) > kDeltaEps) return f3(0.0f);
            return tex_eval(M.tex[1], b.dg) * R;
        }
        if (!st || fabsf(dot(-b.wi, b.wo) - 1) > kDeltaEps) return f3(0.0f);
        return tex_eval(M.tex[0], b.dg) * (1 - R);
    }
    case CTL_BSDF_CONDUCTOR:
        if (!(b.type_mask & CTL_EDeltaReflection) || cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0 || fabsf(dot(reflect_local(b.wi), b.wo) - 1) > kDeltaEps) return f3(0.0f);
        return tex_eval(M.tex[0], b.dg) * fresnel_conductor_exact(cos_theta(b.wi), f3(M.f[0], M.f[1], M.f[2]), f3(M.f[3], M.f[4], M.f[5]));
    case CTL_BSDF_PLASTIC: {
        if (!(b.type_mask & CTL_EDeltaReflection) || cos_theta(b.wo) <= 0 || cos_theta(b.wi) <= 0) return f3(0.0f);
        float ct; const float Fi = fresnel_dielectric_ext(cos_theta(b.wi), ct, M.f[2]);
        if (fabsf(dot(reflect_local(b.wi), b.wo) - 1) < kDeltaEps) return tex_eval(M.tex[1], b.dg) * Fi;
        return f3(0.0f);
    }
    default: return f3(0.0f);
    }
}
__device__ float bsdf_pdf_discrete(const ctl_material& M, const bsdf_rec& b) {
    const float kDeltaEps = 1e-3f;
    switch (M.bsdf_type) {
    case CTL_BSDF_DIELECTRIC: {
        const bool sr = (b.type_mask & CTL_EDeltaReflection) != 0, st = (b.type_mask & CTL_EDeltaTransmission) != 0;
        float cosThetaT; const float eta = M.f[0] + M.f[1] / (600 / 1e3f), invEta = 1.0f / eta;
        const float F = fresnel_dielectric_ext(cos_theta(b.wi), cosThetaT, eta);
        if (cos_theta(b.wi) * cos_theta(b.wo) >= 0) {
            if (!sr || fabsf(dot(reflect_local(b.wi), b.wo) - 1) > kDeltaEps) return 0.0f;
            return st ? 1.0f * F : 1.0f;
        }
        if (!st || fabsf(dot(refract_local(b.wi, cosThetaT, eta, invEta), b.wo) - 1) > kDeltaEps) return 0.0f;
        return sr ? 1 - F : 1.0f * 1.0f;
    }
    case CTL_BSDF_THINDIELECTRIC: {
        const bool sr = (b.type_mask & CTL_EDeltaReflection) != 0, st = (b.type_mask & CTL_ENull) != 0;
        float ct; float R = fresnel_dielectric_ext(fabsf(cos_theta(b.wi)), ct, M.f[0]); const float T = 1 - R;
        if (R < 1) R += T * T * R / (1 - R * R);
        if (cos_theta(b.wi) * cos_theta(b.wo) >= 0) {
            if (!sr || fabsf(dot(reflect_local(b.wi), b.wo) - 1) > kDeltaEps) return 0.0f;
            return st ? R : 1.0f;
        }
        if (!st || fabsf(dot(-b.wi, b.wo) - 1) > kDeltaEps) return 0.0f;
        return sr ? 1 - R : 1.0f;
    }
    case CTL_BSDF_CONDUCTOR:
        if (!(b.type_mask & CTL_EDeltaReflection) || cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0 || fabsf(dot(reflect_local(b.wi), b.wo) - 1) > kDeltaEps) return 0.0f;
        return 1.0f;
    case CTL_BSDF_PLASTIC: {
        const bool hs = (b.type_mask & CTL_EDeltaReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
        if (cos_theta(b.wo) <= 0 || cos_theta(b.wi) <= 0) return 0.0f;
        float ps = hs ? 1.0f : 0.0f;
        if (hs && hd) { float ct; const float Fi = fresnel_dielectric_ext(cos_theta(b.wi), ct, M.f[2]); ps = (Fi * M.f[4]) / (Fi * M.f[4] + (1 - Fi) * (1 - M.f[4])); }
        if (hs && fabsf(dot(reflect_local(b.wi), b.wo) - 1) < kDeltaEps) return ps;
        return 0.0f;
    }
    default: return 0.0f;
    }
}

// calls into the simple dispatch, out of line: each nesting model would otherwise inline every simple model again
__device__ __noinline__ f3 nested_sample(const ctl_material& M, bsdf_rec& b, float& pdf, f2 smp) { return bsdf_sample(M, b, pdf, smp); }
__device__ __noinline__ f3 nested_f(const ctl_material& M, const bsdf_rec& b, int measure) { return measure == kMeasDiscrete ? bsdf_f_discrete(M, b) : bsdf_f(M, b); }
__device__ __noinline__ float nested_pdf(const ctl_material& M, const bsdf_rec& b, int measure) { return measure == kMeasDiscrete ? bsdf_pdf_discrete(M, b) : bsdf_pdf(M, b); }

__device__ __forceinline__ f3 coat_refract_in(const ctl_material& M, f3 wi, float& R) {
    float ct; R = fresnel_dielectric_ext(fabsf(cos_theta(wi)), ct, M.f[0]);
    return normalize(f3(M.f[1] * wi.x, M.f[1] * wi.y, -signum1(cos_theta(wi)) * ct));
}
__device__ __forceinline__ f3 coat_refract_out(const ctl_material& M, f3 wi, float& R) {
    float ct; R = fresnel_dielectric_ext(fabsf(cos_theta(wi)), ct, M.f[1]);
    return normalize(f3(M.f[0] * wi.x, M.f[0] * wi.y, -signum1(cos_theta(wi)) * ct));
}
__device__ __forceinline__ float coat_prob_specular(const ctl_material& M, float R12) { return (R12 * M.f[3]) / (R12 * M.f[3] + (1 - R12) * (1 - M.f[3])); }
__device__ __forceinline__ f3 roughcoat_refract_to(const ctl_material& M, bool interior, f3 wi) {
    const float cosThetaI = cos_theta(wi), invEta = interior ? M.f[1] : M.f[0];
    const float sinThetaTSqr = invEta * invEta * sin_theta2(wi);
    if (sinThetaTSqr >= 1.0f) return f3(0.0f);
    const float cosThetaT = sqrtf(1.0f - sinThetaTSqr);
    return normalize(f3(invEta * wi.x, invEta * wi.y, cosThetaI > 0.0f ? cosThetaT : -cosThetaT));
}
__device__ __forceinline__ microfacet roughcoat_distr(const ctl_material& M, const diff_geom& dg) { const float a = avg3(tex_eval(M.tex[2], dg)); return microfacet((int)M.u[0], a, a, M.u[1] != 0); }
__device__ __forceinline__ float roughcoat_prob_specular(const ctl_material& M, const bsdf_rec& b, const microfacet& distr) {
    const float p = 1 - rough_transmittance(b.dg, M.u[0], fabsf(cos_theta(b.wi)), distr.aU, M.f[0]);
    return (p * M.f[3]) / (p * M.f[3] + (1 - p) * (1 - M.f[3]));
}

__device__ f3 bsdf_complex_f(const ctl_material& M, const bsdf_rec& b, int measure);
__device__ float bsdf_complex_pdf(const ctl_material& M, const bsdf_rec& b, int measure);

__device__ f3 bsdf_complex_sample(const ctl_material& M, bsdf_rec& b, float& pdf, f2 smp) {
    const ctl_material* mats = b.dg.mats;
    switch (M.bsdf_type) {
    case CTL_BSDF_COATING: {   // BSDF_Complex.cu:6-82
        const ctl_material& nested = mats[M.u[2]];
        const bool ss = (b.type_mask & CTL_EDeltaReflection) != 0, sn = (b.type_mask & nested.combined_type & kEAll) != 0;
        if (!ss && !sn) return f3(0.0f);
        float R12; const f3 wiPrime = coat_refract_in(M, b.wi, R12);
        const float ps = coat_prob_specular(M, R12);
        bool spec = ss;
        if (ss && sn) { if (smp.x < ps) smp.x /= ps; else { smp.x = (smp.x - ps) / (1 - ps); spec = false; } }
        if (spec) {
            b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); b.eta = 1.0f;
            pdf = sn ? ps : 1.0f;
            return tex_eval(M.tex[1], b.dg) * (R12 / pdf);
        }
        if (R12 == 1.0f) return f3(0.0f);
        const f3 wiBackup = b.wi; b.wi = wiPrime;
        f3 result = nested_sample(nested, b, pdf, smp);
        b.wi = wiBackup;
        if (is_zero(result)) return f3(0.0f);
        const f3 woPrime = b.wo;
        const f3 sigmaA = tex_eval(M.tex[0], b.dg) * M.f[2];
        if (!is_zero(sigmaA)) result = result * exp3(-sigmaA * (1 / fabsf(cos_theta(wiPrime)) + 1 / fabsf(cos_theta(woPrime))));
        float R21; b.wo = coat_refract_out(M, woPrime, R21);
        if (R21 == 1.0f) return f3(0.0f);
        if (ss) { pdf *= 1.0f - ps; result = sdiv(result, 1.0f - ps); }
        result = result * ((1 - R12) * (1 - R21));
        if (bsdf_measure(b.sampled_type) == kMeasSolidAngle) {
            result = result * (cos_theta(b.wi) / cos_theta(wiPrime));
            pdf *= M.f[1] * M.f[1] * cos_theta(b.wo) / cos_theta(woPrime);
        }
        return result;
    }
    case CTL_BSDF_ROUGHCOATING: {   // BSDF_Complex.cu:159-224
        const ctl_material& nested = mats[M.u[2]];
        const bool hn = (b.type_mask & nested.combined_type & kEAll) != 0, hs = (b.type_mask & CTL_EGlossyReflection) != 0;
        bool spec = hs;
        const microfacet distr = roughcoat_distr(M, b.dg);
        if (hs && hn) { const float ps = roughcoat_prob_specular(M, b, distr); if (smp.y < ps) smp.y /= ps; else { smp.y = (smp.y - ps) / (1 - ps); spec = false; } }
        if (spec) {
            float unused; const f3 m = distr.sample(b.wi, smp, unused);
            b.wo = reflect_about(b.wi, m); b.sampled_type = CTL_EGlossyReflection; b.eta = 1.0f;
            if (cos_theta(b.wo) * cos_theta(b.wi) <= 0) return f3(0.0f);
        } else {
            const f3 wiBackup = b.wi; b.wi = roughcoat_refract_to(M, true, b.wi);
            const f3 result = nested_sample(nested, b, pdf, smp);
            b.wi = wiBackup;
            if (is_zero(result)) return f3(0.0f);
            b.wo = roughcoat_refract_to(M, false, b.wo);
            if (dot(b.wo, b.wo) == 0.0f) return f3(0.0f);
        }
        const int measure = bsdf_measure(b.sampled_type);
        pdf = bsdf_complex_pdf(M, b, measure);
        if (pdf == 0) return f3(0.0f);
        return sdiv(bsdf_complex_f(M, b, measure), pdf);
    }
    case CTL_BSDF_BLEND: {   // BSDF_Complex.cu:344-372
        float w[2]; w[1] = clampf(avg3(tex_eval(M.tex[0], b.dg)), 0.0f, 1.0f); w[0] = 1.0f - w[1];
        uint32_t entry;
        if (smp.x < w[0]) { entry = 0; smp.x /= w[0]; } else { entry = 1; smp.x = (smp.x - w[0]) / w[1]; }
        f3 result = nested_sample(mats[M.u[2 + entry]], b, pdf, smp);
        if (is_zero(result)) return result;
        result = result * (w[entry] * pdf);
        pdf *= w[entry];
        const int measure = bsdf_measure(b.sampled_type);
        const uint32_t other = 1 - entry;
        pdf += nested_pdf(mats[M.u[2 + other]], b, measure) * w[other];
        result = result + nested_f(mats[M.u[2 + other]], b, measure) * w[other];
        return sdiv(result, pdf);
    }
    default: return f3(0.0f);
    }
}

__device__ f3 bsdf_complex_f(const ctl_material& M, const bsdf_rec& b, int measure) {
    const ctl_material* mats = b.dg.mats;
    switch (M.bsdf_type) {
    case CTL_BSDF_COATING: {   // BSDF_Complex.cu:84-122
        const ctl_material& nested = mats[M.u[2]];
        const bool ss = (b.type_mask & CTL_EDeltaReflection) != 0, sn = (b.type_mask & nested.combined_type & kEAll) != 0;
        if (measure == kMeasDiscrete && ss && fabsf(dot(reflect_local(b.wi), b.wo) - 1) < 1e-3f) { float ct; return tex_eval(M.tex[1], b.dg) * fresnel_dielectric_ext(fabsf(cos_theta(b.wi)), ct, M.f[0]); }
        if (sn) {
            float R12, R21; bsdf_rec bi = b;
            bi.wi = coat_refract_in(M, b.wi, R12); bi.wo = coat_refract_in(M, b.wo, R21);
            if (R12 == 1 || R21 == 1) return f3(0.0f);
            f3 result = nested_f(nested, bi, measure) * (1 - R12) * (1 - R21);
            const f3 sigmaA = tex_eval(M.tex[0], b.dg) * M.f[2];
            if (!is_zero(sigmaA)) result = result * exp3(-sigmaA * (1 / fabsf(cos_theta(bi.wi)) + 1 / fabsf(cos_theta(bi.wo))));
            if (measure == kMeasSolidAngle) result = result * (M.f[1] * M.f[1] * cos_theta(b.wi) * cos_theta(b.wo) / (cos_theta(bi.wi) * cos_theta(bi.wo)));
            return result;
        }
        return f3(0.0f);
    }
    case CTL_BSDF_ROUGHCOATING: {   // BSDF_Complex.cu:226-284
        const ctl_material& nested = mats[M.u[2]];
        const bool hn = (b.type_mask & nested.combined_type & kEAll) != 0, hs = (b.type_mask & CTL_EGlossyReflection) != 0 && measure == kMeasSolidAngle;
        const microfacet distr = roughcoat_distr(M, b.dg);
        f3 result(0.0f);
        if (hs && cos_theta(b.wo) * cos_theta(b.wi) > 0) {
            const f3 H = normalize(b.wo + b.wi) * signum1(cos_theta(b.wo));
            const float D = distr.eval(H);
            float ct; const float F = fresnel_dielectric_ext(absdot(b.wi, H), ct, M.f[0]);
            const float G = distr.G(b.wi, b.wo, H);
            const float value = F * D * G / (4.0f * fabsf(cos_theta(b.wi)));
            result = result + tex_eval(M.tex[1], b.dg) * value;
        }
        if (hn) {
            bsdf_rec bi = b;
            bi.wi = roughcoat_refract_to(M, true, b.wi); bi.wo = roughcoat_refract_to(M, true, b.wo);
            f3 nr = nested_f(nested, bi, measure) * rough_transmittance(b.dg, M.u[0], cos_theta(b.wi), distr.aU, M.f[0]) * rough_transmittance(b.dg, M.u[0], cos_theta(b.wo), distr.aU, M.f[0]);
            const f3 sigmaA = tex_eval(M.tex[0], b.dg) * M.f[2];
            if (!is_zero(sigmaA)) nr = nr * exp3(-sigmaA * (1 / fabsf(cos_theta(bi.wi)) + 1 / fabsf(cos_theta(bi.wo))));
            if (measure == kMeasSolidAngle) nr = nr * (M.f[1] * M.f[1] * cos_theta(b.wi) * cos_theta(b.wo) / (cos_theta(bi.wi) * cos_theta(bi.wo)));
            result = result + nr;
        }
        return result;
    }
    case CTL_BSDF_BLEND: {   // BSDF_Complex.cu:374-378
        const float weight = clampf(avg3(tex_eval(M.tex[0], b.dg)), 0.0f, 1.0f);
        return nested_f(mats[M.u[2]], b, measure) * (1 - weight) + nested_f(mats[M.u[3]], b, measure) * weight;
    }
    default: return f3(0.0f);
    }
}

__device__ float bsdf_complex_pdf(const ctl_material& M, const bsdf_rec& b, int measure) {
    const ctl_material* mats = b.dg.mats;
    switch (M.bsdf_type) {
    case CTL_BSDF_COATING: {   // BSDF_Complex.cu:124-157
        const ctl_material& nested = mats[M.u[2]];
        const bool ss = (b.type_mask & CTL_EDeltaReflection) != 0, sn = (b.type_mask & nested.combined_type & kEAll) != 0;
        float R12; const f3 wiPrime = coat_refract_in(M, b.wi, R12);
        const float ps = coat_prob_specular(M, R12);
        if (measure == kMeasDiscrete && ss && fabsf(dot(reflect_local(b.wi), b.wo) - 1) < 1e-3f) return sn ? ps : 1.0f;
        if (sn) {
            float R21; bsdf_rec bi = b;
            bi.wi = wiPrime; bi.wo = coat_refract_in(M, b.wo, R21);
            if (R12 == 1 || R21 == 1) return 0.0f;
            float pdf = nested_pdf(nested, bi, measure);
            if (measure == kMeasSolidAngle) pdf *= M.f[1] * M.f[1] * cos_theta(b.wo) / cos_theta(bi.wo);
            return ss ? (pdf * (1 - ps)) : pdf;
        }
        return 0.0f;
    }
    case CTL_BSDF_ROUGHCOATING: {   // BSDF_Complex.cu:286-342
        const ctl_material& nested = mats[M.u[2]];
        const bool hn = (b.type_mask & nested.combined_type & kEAll) != 0, hs = (b.type_mask & CTL_EGlossyReflection) != 0 && measure == kMeasSolidAngle;
        const f3 H = normalize(b.wo + b.wi) * signum1(cos_theta(b.wo));
        const microfacet distr = roughcoat_distr(M, b.dg);
        float pn, ps;
        if (hs && hn) { ps = roughcoat_prob_specular(M, b, distr); pn = 1 - ps; } else pn = ps = 1.0f;
        float result = 0.0f;
        if (hs && cos_theta(b.wo) * cos_theta(b.wi) > 0) {
            const float dwh_dwo = 1.0f / (4.0f * absdot(b.wo, H));
            const float prob = distr.pdf(b.wi, H);
            result = prob * dwh_dwo * ps;
        }
        if (hn) {
            bsdf_rec bi = b;
            bi.wi = roughcoat_refract_to(M, true, b.wi); bi.wo = roughcoat_refract_to(M, true, b.wo);
            float prob = nested_pdf(nested, bi, measure);
            if (measure == kMeasSolidAngle) prob *= M.f[1] * M.f[1] * cos_theta(b.wo) / cos_theta(bi.wo);
            result += prob * pn;
        }
        return result;
    }
    case CTL_BSDF_BLEND: {   // BSDF_Complex.cu:380-384
        const float weight = clampf(avg3(tex_eval(M.tex[0], b.dg)), 0.0f, 1.0f);
        return nested_pdf(mats[M.u[2]], b, measure) * (1 - weight) + nested_pdf(mats[M.u[3]], b, measure) * weight;
    }
    default: return 0.0f;
    }
}

} // namespace ctl
