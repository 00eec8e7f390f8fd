// bsdf_rough.h — roughdiffuse, ward, roughplastic (SceneTypes/BSDF_Simple.cu:82-172, 1173-1313, 890-1057) with the
// rough-transmittance tables (Engine/RoughTransmittance.cu:55-119) and their cubic spline lookup (Math/Spline.cu:223-453).
// Included by bsdf_more.h; expression order follows the reference (no FMA contraction), see DESIGN.md §4.
#pragma once
#include "spline.h"

namespace ctl {

// Spline::evalCubicInterp2D / 3D over [0,1]^n: spline.h (one statement for the kernels, the host-side table reduction and the test against the reference's Spline.cu)
__device__ CTL_ROUGH_BODY float eval_cubic_interp_2d(float px, float py, const float* __restrict__ values, uint32_t sx, uint32_t sy) { return spline_eval_2d(px, py, values, sx, sy); }
__device__ CTL_ROUGH_BODY float eval_cubic_interp_3d(float px, float py, float pz, const float* __restrict__ values, uint32_t sx, uint32_t sy, uint32_t sz) { return spline_eval_3d(px, py, pz, values, sx, sy, sz); }
// RoughTransmittanceManager::Evaluate / EvaluateDiffuse for the table of slot `type`
__device__ CTL_ROUGH_BODY float rough_transmittance(const diff_geom& dg, uint32_t type, float cosTheta, float alpha, float eta) {
    const ctl_rough_transmittance& T = dg.rough_transmittance[type];
    const float warpedCosTheta = m_pow(fabsf(cosTheta), 0.25f);
    if (cosTheta < 0) { cosTheta = -cosTheta; eta = 1.0f / eta; }
    const float* data = T.trans;
    if (eta < 1) { data += (size_t)T.eta_samples * T.alpha_samples * T.theta_samples; eta = 1.0f / eta; }
    if (eta < T.eta_min) eta = T.eta_min;
    const float warpedAlpha = m_pow((alpha - T.alpha_min) / (T.alpha_max - T.alpha_min), 0.25f);
    const float warpedEta = m_pow((eta - T.eta_min) / (T.eta_max - T.eta_min), 0.25f);
    const float result = eval_cubic_interp_3d(warpedCosTheta, warpedAlpha, warpedEta, data, T.theta_samples, T.alpha_samples, T.eta_samples);
    return min2(1.0f, max2(0.0f, result));
}
__device__ CTL_ROUGH_BODY float rough_transmittance_diffuse(const diff_geom& dg, uint32_t type, float alpha, float eta) {
    const ctl_rough_transmittance& T = dg.rough_transmittance[type];
    const float* data = T.diff_trans;
    if (eta < 1) { data += (size_t)T.eta_samples * T.alpha_samples; eta = 1.0f / eta; }
    if (eta < T.eta_min) eta = T.eta_min;
    const float warpedAlpha = m_pow((alpha - T.alpha_min) / (T.alpha_max - T.alpha_min), 0.25f);
    const float warpedEta = m_pow((eta - T.eta_min) / (T.eta_max - T.eta_min), 0.25f);
    const float result = eval_cubic_interp_2d(warpedAlpha, warpedEta, data, T.alpha_samples, T.eta_samples);
    return min2(1.0f, max2(0.0f, result));
}

__device__ __forceinline__ float sin_phi(f3 v) { const float st = sin_theta(v); if (st == 0.0f) return 1.0f; return clampf(v.y / st, -1.0f, 1.0f); }   // Frame.h
__device__ __forceinline__ float cos_phi(f3 v) { const float st = sin_theta(v); if (st == 0.0f) return 1.0f; return clampf(v.x / st, -1.0f, 1.0f); }
__device__ __forceinline__ float safe_acosf(float v) { return m_acos(min2(1.0f, max2(-1.0f, v))); }
__device__ __forceinline__ float safe_sqrtf(float v) { return sqrtf(max2(0.0f, v)); }

__device__ CTL_ROUGH_BODY f3 roughdiffuse_f(const ctl_material& M, const bsdf_rec& b) {   // BSDF_Simple.cu:82-172
    if (!(b.type_mask & CTL_EGlossyReflection) || cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0) return f3(0.0f);
    const float conversionFactor = 1 / sqrtf((float)2);
    const float sigma = avg3(tex_eval(M.tex[1], b.dg)) * conversionFactor;
    const float sigma2 = sigma * sigma;
    const float sinThetaI = sin_theta(b.wi), sinThetaO = sin_theta(b.wo);
    float cosPhiDiff = 0;
    if (sinThetaI > 0.000001f && sinThetaO > 0.000001f) cosPhiDiff = cos_phi(b.wi) * cos_phi(b.wo) + sin_phi(b.wi) * sin_phi(b.wo);
    if (M.u[0]) {
        const float A = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f), B = 0.45f * sigma2 / (sigma2 + 0.09f);
        float sinAlpha, tanBeta;
        if (cos_theta(b.wi) > cos_theta(b.wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / cos_theta(b.wi); }
        else { sinAlpha = sinThetaI; tanBeta = sinThetaO / cos_theta(b.wo); }
        return tex_eval(M.tex[0], b.dg) * (kInvPi * cos_theta(b.wo) * (A + B * max2(cosPhiDiff, 0.0f) * sinAlpha * tanBeta));
    }
    const float thetaI = safe_acosf(cos_theta(b.wi)), thetaO = safe_acosf(cos_theta(b.wo)), alpha = max2(thetaI, thetaO), beta = min2(thetaI, thetaO);
    float sinAlpha, sinBeta, tanBeta;
    if (cos_theta(b.wi) > cos_theta(b.wo)) { sinAlpha = sinThetaO; sinBeta = sinThetaI; tanBeta = sinThetaI / cos_theta(b.wi); }
    else { sinAlpha = sinThetaI; sinBeta = sinThetaO; tanBeta = sinThetaO / cos_theta(b.wo); }
    const float tmp = sigma2 / (sigma2 + 0.09f), tmp2 = (4 * kInvPi * kInvPi) * alpha * beta, tmp3 = 2 * beta * kInvPi;
    const float C1 = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f), C3 = 0.125f * tmp * tmp2 * tmp2, C4 = 0.17f * sigma2 / (sigma2 + 0.13f);
    float C2 = 0.45f * tmp;
    if (cosPhiDiff > 0) C2 *= sinAlpha; else C2 *= sinAlpha - tmp3 * tmp3 * tmp3;
    const float tanHalf = (sinAlpha + sinBeta) / (safe_sqrtf(1.0f - sinAlpha * sinAlpha) + safe_sqrtf(1.0f - sinBeta * sinBeta));
    const f3 rho = tex_eval(M.tex[0], b.dg);
    const f3 snglScat = rho * (C1 + cosPhiDiff * C2 * tanBeta + (1.0f - fabsf(cosPhiDiff)) * C3 * tanHalf);
    const f3 dblScat = rho * rho * (C4 * (1.0f - cosPhiDiff * tmp3 * tmp3));
    return (snglScat + dblScat) * (kInvPi * cos_theta(b.wo));
}

__device__ CTL_ROUGH_BODY f3 ward_f(const ctl_material& M, const bsdf_rec& b) {   // BSDF_Simple.cu:1232-1276
    if (cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0) return f3(0.0f);
    const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
    f3 result(0.0f);
    if (hs) {
        const f3 H = b.wi + b.wo;
        const float alphaU = avg3(tex_eval(M.tex[2], b.dg)), alphaV = avg3(tex_eval(M.tex[3], b.dg));
        float factor1 = 0.0f;
        switch (M.u[0]) {
        case 0: factor1 = 1.0f / (4.0f * kPi * alphaU * alphaV * sqrtf(cos_theta(b.wi) * cos_theta(b.wo))); break;
        case 1: factor1 = 1.0f / (4.0f * kPi * alphaU * alphaV * cos_theta(b.wi) * cos_theta(b.wo)); break;
        case 2: factor1 = dot(H, H) / (kPi * alphaU * alphaV * m_pow(cos_theta(normalize(H)), 4)); break;
        }
        const float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
        const float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
        const float specRef = factor1 * m_exp(exponent);
        if (specRef > 1e-10f) result = result + tex_eval(M.tex[1], b.dg) * specRef;
    }
    if (hd) result = result + tex_eval(M.tex[0], b.dg) * kInvPi;
    return result * cos_theta(b.wo);
}
__device__ CTL_ROUGH_BODY float ward_pdf(const ctl_material& M, const bsdf_rec& b) {   // BSDF_Simple.cu:1278-1313
    if (cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0) return 0.0f;
    const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
    float diffuseProb = 0.0f, specProb = 0.0f; const float ssw = M.f[0];
    if (hs) {
        const float alphaU = avg3(tex_eval(M.tex[2], b.dg)), alphaV = avg3(tex_eval(M.tex[3], b.dg));
        const f3 H = normalize(b.wi + b.wo);
        const float factor1 = 1.0f / (4.0f * kPi * alphaU * alphaV * dot(H, b.wi) * m_pow(cos_theta(H), 3));
        const float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
        const float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
        specProb = factor1 * m_exp(exponent);
    }
    if (hd) diffuseProb = kInvPi * cos_theta(b.wo);
    if (hd && hs) return ssw * specProb + (1 - ssw) * diffuseProb;
    else if (hd) return diffuseProb;
    else if (hs) return specProb;
    return 0.0f;
}

// Rough plastic with a constant roughness looks the table up at a fixed (alpha, eta): what depends on those two alone is made once per material at scene upload
// (tracer.hip), M.reserved_ = {offset + 1 into dev_scene::rt_reduced, theta samples | kRtRows}.
//  * kRtRows (the default): the SIXTEEN ROWS of the table the 3-D interpolation reads for this (alpha, eta) and their sixteen weight products wy * wz, followed by the constant
//    the 2-D diffuse lookup returns: {wyz[16], rows[16][samples], diffuse}.  rough_transmittance_rows runs the sum of spline_eval_3d over them — the same 64 products and 64
//    additions in the same order, the same skipped zero weights — so the value is the reference's to the bit; what is saved are the two pow() of the warp, two sets of spline
//    weights and the strided addressing of every lookup.
//  * without kRtRows (CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE): the rows summed over alpha / eta beforehand, {table[samples], diffuse}: four taps, equal up to rounding only.
constexpr uint32_t kRtRows = 0x80000000u;
__device__ __forceinline__ float rough_transmittance_1d(const float* __restrict__ table, uint32_t size, float cosTheta) {
    return min2(1.0f, max2(0.0f, spline_eval_1d(m_pow(fabsf(cosTheta), 0.25f), table, size)));
}
__device__ CTL_ROUGH_BODY float rough_transmittance_rows(const float* __restrict__ blk, uint32_t sx, float cosTheta) {   // = rough_transmittance for cosTheta >= 0 at the material's (alpha, eta)
    float wx[4]; uint32_t kx;
    if (!spline_weights(m_pow(fabsf(cosTheta), 0.25f), sx, wx, kx)) return 0.0f;
    // spline_eval_3d skips a tap whose weight product is zero; here every tap is added.  The same value: the sum starts at +0 and, in round-to-nearest, can only ever be +0 or
    // non-zero (+0 + -0 = +0, x + -x = +0), so adding a zero product never changes it; the table is finite, the taps beside a border knot (weight 0) read the neighbouring
    // row's end, inside the block.  Without the data-dependent branch the 64 loads issue together instead of one behind the other.
    const float* __restrict__ rows = blk + 16 + ((int)kx - 1);
    float result = 0.0f;
#pragma unroll
    for (int zy = 0; zy < 16; ++zy) {
        const float wyz = blk[zy];
        const float* __restrict__ r = rows + zy * (int)sx;
#pragma unroll
        for (int x = 0; x < 4; ++x) result += r[x] * (wx[x] * wyz);
    }
    return min2(1.0f, max2(0.0f, result));
}
__device__ __forceinline__ float roughplastic_diffuse_T(const ctl_material& M, const bsdf_rec& b) {   // the material's constant EvaluateDiffuse(alpha, eta), behind its table
    const uint32_t n = M.reserved_[1] & ~kRtRows;
    return b.dg.rt_reduced[M.reserved_[0] - 1 + ((M.reserved_[1] & kRtRows) ? 16u + 16u * n : n)];
}
__device__ __forceinline__ float roughplastic_T(const ctl_material& M, const bsdf_rec& b, float cosTheta, float alpha) {   // cosTheta > 0 on every roughplastic path
    if (M.reserved_[0]) {
        const float* __restrict__ blk = b.dg.rt_reduced + (M.reserved_[0] - 1); const uint32_t n = M.reserved_[1] & ~kRtRows; const bool rows = (M.reserved_[1] & kRtRows) != 0;
#if CTL_SHADE_FEATURES & 2
        // sample, f and pdf of a vertex (and the NEE evaluation behind them) ask for T(cos wi) five times: memoised like the 3-D lookups, keyed by the table's offset
        const uint32_t key = M.reserved_[0] | 0x80000000u;
        if (b.rt_cos == cosTheta && b.rt_type == key) return b.rt_val;
        const float v = rows ? rough_transmittance_rows(blk, n, cosTheta) : rough_transmittance_1d(blk, n, cosTheta);
        if (cosTheta == cos_theta(b.wi)) { b.rt_cos = cosTheta; b.rt_type = key; b.rt_val = v; }   // only the incident direction recurs
        return v;
#else
        return rows ? rough_transmittance_rows(blk, n, cosTheta) : rough_transmittance_1d(blk, n, cosTheta);
#endif
    }
    return rough_transmittance(b.dg, M.u[2], cosTheta, alpha, M.f[0]);
}
__device__ __forceinline__ float rough_transmittance_wi(const bsdf_rec& b, uint32_t type, float cosTheta, float alpha, float eta) {
#if CTL_SHADE_FEATURES & 2
    if (b.rt_cos == cosTheta && b.rt_alpha == alpha && b.rt_eta == eta && b.rt_type == type) return b.rt_val;
    const float v = rough_transmittance(b.dg, type, cosTheta, alpha, eta);
    b.rt_cos = cosTheta; b.rt_alpha = alpha; b.rt_eta = eta; b.rt_type = type; b.rt_val = v;
    return v;
#else
    return rough_transmittance(b.dg, type, cosTheta, alpha, eta);
#endif
}
__device__ __forceinline__ float rough_transmittance_diffuse_memo(const bsdf_rec& b, uint32_t type, float alpha, float eta) {
#if CTL_SHADE_FEATURES & 2
    if (b.rtd_alpha == alpha && b.rtd_eta == eta && b.rtd_type == type) return b.rtd_val;
    const float v = rough_transmittance_diffuse(b.dg, type, alpha, eta);
    b.rtd_alpha = alpha; b.rtd_eta = eta; b.rtd_type = type; b.rtd_val = v;
    return v;
#else
    return rough_transmittance_diffuse(b.dg, type, alpha, eta);
#endif
}
__device__ __forceinline__ microfacet roughplastic_distr(const ctl_material& M, const diff_geom& dg) {
    const float a = avg3(tex_eval(M.tex[2], dg));
    return microfacet((int)M.u[2], a, a, M.u[1] != 0);
}
__device__ __forceinline__ float roughplastic_prob_specular(const ctl_material& M, const bsdf_rec& b, const microfacet& distr) {
    const float ps = 1 - (M.reserved_[0] ? roughplastic_T(M, b, cos_theta(b.wi), distr.aU) : rough_transmittance_wi(b, M.u[2], cos_theta(b.wi), distr.aU, M.f[0]));
    return (ps * M.f[2]) / (ps * M.f[2] + (1 - ps) * (1 - M.f[2]));
}
__device__ CTL_ROUGH_BODY f3 roughplastic_f(const ctl_material& M, const bsdf_rec& b) {   // BSDF_Simple.cu:948-1005
    const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
    if (cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0 || (!hs && !hd)) return f3(0.0f);
    const microfacet distr = roughplastic_distr(M, b.dg);
    f3 result(0.0f);
    if (hs) {
        const f3 H = normalize(b.wo + b.wi);
        const float D = distr.eval(H);
        float ct; const float F = fresnel_dielectric_ext(dot(b.wi, H), ct, M.f[0]);
        const float G = distr.G(b.wi, b.wo, H);
        const float value = F * D * G / (4.0f * cos_theta(b.wi));
        result = result + tex_eval(M.tex[1], b.dg) * value;
    }
    if (hd) {
        f3 diff = tex_eval(M.tex[0], b.dg);
        const float T12 = M.reserved_[0] ? roughplastic_T(M, b, cos_theta(b.wi), distr.aU) : rough_transmittance_wi(b, M.u[2], cos_theta(b.wi), distr.aU, M.f[0]);
        const float T21 = roughplastic_T(M, b, cos_theta(b.wo), distr.aU);
        const float Fdr = 1 - (M.reserved_[0] ? roughplastic_diffuse_T(M, b) : rough_transmittance_diffuse_memo(b, M.u[2], distr.aU, M.f[0]));
        if (M.u[0]) diff = diff / (f3(1.0f) - diff * Fdr);
        else diff = sdiv(diff, 1 - Fdr);
        result = result + diff * (kInvPi * cos_theta(b.wo) * T12 * T21 * M.f[1]);
    }
    return result;
}
__device__ CTL_ROUGH_BODY float roughplastic_pdf(const ctl_material& M, const bsdf_rec& b) {   // BSDF_Simple.cu:1007-1057
    const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
    if (cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0 || (!hs && !hd)) return 0.0f;
    const microfacet distr = roughplastic_distr(M, b.dg);
    const f3 H = normalize(b.wo + b.wi);
    float probDiffuse, probSpecular;
    if (hs && hd) { probSpecular = roughplastic_prob_specular(M, b, distr); probDiffuse = 1 - probSpecular; }
    else probDiffuse = probSpecular = 1.0f;
    float result = 0.0f;
    if (hs) {
        const float dwh_dwo = 1.0f / (4.0f * dot(b.wo, H));
        const float prob = distr.pdf(b.wi, H);
        result = prob * dwh_dwo * probSpecular;
    }
    if (hd) result += probDiffuse * (kInvPi * cos_theta(b.wo));
    return result;
}

__device__ CTL_ROUGH_OUTLINE f3 bsdf_rough_sample(const ctl_material& M, bsdf_rec& b, float& pdf, f2 smp) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIFFUSE: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHDIFFUSE)) return f3(0.0f);   // BSDF_Simple.h:42-49
        b.wo = square_to_cosine_hemisphere(smp); b.eta = 1.0f; b.sampled_type = CTL_EGlossyReflection;
        pdf = kInvPi * cos_theta(b.wo);
        return sdiv(roughdiffuse_f(M, b), pdf);
    }
    case CTL_BSDF_WARD: { if (!CTL_HAS_MODEL(CTL_BSDF_WARD)) return f3(0.0f);   // BSDF_Simple.cu:1173-1230
        const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
        if (!hs && !hd) return f3(0.0f);
        bool spec = hs; const float ssw = M.f[0];
        if (hd && hs) { if (smp.x <= ssw) smp.x /= ssw; else { smp.x = (smp.x - ssw) / (1 - ssw); spec = false; } }
        if (spec) {
            const float alphaU = avg3(tex_eval(M.tex[2], b.dg)), alphaV = avg3(tex_eval(M.tex[3], b.dg));
            float phiH = m_atan(alphaV / alphaU * m_tan(2.0f * kPi * smp.y));
            if (smp.y > 0.5f) phiH += kPi;
            const float cosPhiH = m_cos(phiH);
            const float sinPhiH = safe_sqrtf(1.0f - cosPhiH * cosPhiH);
            const float thetaH = m_atan(safe_sqrtf(-m_log(smp.x) / ((cosPhiH * cosPhiH) / (alphaU * alphaU) + (sinPhiH * sinPhiH) / (alphaV * alphaV))));
            const float sinTheta = m_sin(thetaH), cosTheta = m_cos(thetaH), sinPhi = m_sin(phiH), cosPhi = m_cos(phiH);
            const f3 H(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
            b.wo = reflect_about(b.wi, H);
            b.sampled_type = CTL_EGlossyReflection;
            if (cos_theta(b.wo) <= 0.0f) return f3(0.0f);
        } else { b.wo = square_to_cosine_hemisphere(smp); b.sampled_type = CTL_EDiffuseReflection; }
        b.eta = 1.0f;
        pdf = ward_pdf(M, b);
        if (pdf == 0) return f3(0.0f);
        return sdiv(ward_f(M, b), pdf);
    }
    case CTL_BSDF_ROUGHPLASTIC: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHPLASTIC)) return f3(0.0f);   // BSDF_Simple.cu:890-946
        const bool hs = (b.type_mask & CTL_EGlossyReflection) != 0, hd = (b.type_mask & CTL_EDiffuseReflection) != 0;
        if (cos_theta(b.wi) <= 0 || (!hs && !hd)) return f3(0.0f);
        bool spec = hs;
        const microfacet distr = roughplastic_distr(M, b.dg);
        if (hs && hd) {
            const float ps = roughplastic_prob_specular(M, b, distr);
            if (smp.y < ps) smp.y /= ps; else { smp.y = (smp.y - ps) / (1 - ps); spec = false; }
        }
        if (spec) {
            float unused; const f3 m = distr.sample(b.wi, smp, unused);
            b.wo = reflect_about(b.wi, m);
            b.sampled_type = CTL_EGlossyReflection;
            if (cos_theta(b.wo) <= 0) return f3(0.0f);
        } else { b.sampled_type = CTL_EDiffuseReflection; b.wo = square_to_cosine_hemisphere(smp); }
        b.eta = 1.0f;
        pdf = roughplastic_pdf(M, b);
        if (pdf == 0) return f3(0.0f);
        return sdiv(roughplastic_f(M, b), pdf);
    }
    default: return f3(0.0f);
    }
}
__device__ CTL_ROUGH_OUTLINE f3 bsdf_rough_f(const ctl_material& M, const bsdf_rec& b) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIFFUSE: return CTL_HAS_MODEL(CTL_BSDF_ROUGHDIFFUSE) ? roughdiffuse_f(M, b) : f3(0.0f);
    case CTL_BSDF_WARD: return CTL_HAS_MODEL(CTL_BSDF_WARD) ? ward_f(M, b) : f3(0.0f);
    case CTL_BSDF_ROUGHPLASTIC: return CTL_HAS_MODEL(CTL_BSDF_ROUGHPLASTIC) ? roughplastic_f(M, b) : f3(0.0f);
    default: return f3(0.0f);
    }
}
__device__ CTL_ROUGH_OUTLINE float bsdf_rough_pdf(const ctl_material& M, const bsdf_rec& b) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIFFUSE:   // BSDF_Simple.h:51-59
        if (!(b.type_mask & CTL_EGlossyReflection) || cos_theta(b.wi) <= 0 || cos_theta(b.wo) <= 0) return 0.0f;
        return kInvPi * cos_theta(b.wo);
    case CTL_BSDF_WARD: return CTL_HAS_MODEL(CTL_BSDF_WARD) ? ward_pdf(M, b) : 0.0f;
    case CTL_BSDF_ROUGHPLASTIC: return CTL_HAS_MODEL(CTL_BSDF_ROUGHPLASTIC) ? roughplastic_pdf(M, b) : 0.0f;
    default: return 0.0f;
    }
}

} // namespace ctl
