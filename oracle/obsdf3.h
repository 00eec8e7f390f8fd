// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// obsdf3.h — roughdiffuse, ward, roughplastic (SceneTypes/BSDF_Simple.cu) with the rough-transmittance tables
// (Engine/RoughTransmittance.cu) and the cubic spline interpolation they use (Math/Spline.cu).  Included by ocore.h after
// BRec / Microfacet / texEval are defined.  Pinned on the reference's own code, bit for bit: roughdiffuse / ward against BSDF_Simple.cu (tests/golden/bsdf.npz); the spline against
// Math/Spline.cu (spline.npz); the table lookups against RoughTransmittance::Evaluate / EvaluateDiffuse + the manager's lookup by type, and roughplastic's sample / f / pdf against
// BSDF_Simple.cu:890-1057 run through them (bsdf_rough.npz) — all compiled by `make -C oracle ref`.  Outside: the CONTENT of Mitsuba's data/microfacet/*.dat tables (not part of
// the reference tree; cudatracerlib_amd/rough_tables.py re-derives them).
#pragma once

namespace orc {

// Spline::evalCubicInterp2D / 3D (Math/Spline.cu:223-296, 376-453), extrapolate = false, knots on [0,1]^n
inline bool splineWeights(float p, unsigned size, float* weights, unsigned& knot) {
    if (!(p >= 0.0f && p <= 1.0f)) return false;
    float t = ((p - 0.0f) * (size - 1)) / (1.0f - 0.0f);
    knot = std::min((unsigned)t, size - 2);
    t = t - (float)knot;
    float t2 = t * t, t3 = t2 * t;
    weights[0] = 0.0f; weights[1] = 2 * t3 - 3 * t2 + 1; weights[2] = -2 * t3 + 3 * t2; weights[3] = 0.0f;
    float d0 = t3 - 2 * t2 + t, d1 = t3 - t2;
    if (knot > 0) { weights[2] += 0.5f * d0; weights[0] -= 0.5f * d0; } else { weights[2] += d0; weights[1] -= d0; }
    if (knot + 2 < size) { weights[3] += 0.5f * d1; weights[1] -= 0.5f * d1; } else { weights[2] += d1; weights[1] -= d1; }
    return true;
}
inline float evalCubicInterp2D(float px, float py, const float* values, unsigned sx, unsigned sy) {
    float w[2][4]; unsigned knot[2];
    if (!splineWeights(px, sx, w[0], knot[0]) || !splineWeights(py, sy, w[1], knot[1])) return 0.0f;
    float result = 0.0f;
    for (int y = -1; y <= 2; ++y) {
        float wy = w[1][y + 1];
        for (int x = -1; x <= 2; ++x) {
            float wxy = w[0][x + 1] * wy;
            if (wxy == 0) continue;
            size_t pos = (size_t)(knot[1] + y) * sx + knot[0] + x;
            result += values[pos] * wxy;
        }
    }
    return result;
}
inline float evalCubicInterp3D(float px, float py, float pz, const float* values, unsigned sx, unsigned sy, unsigned sz) {
    float w[3][4]; unsigned knot[3];
    if (!splineWeights(px, sx, w[0], knot[0]) || !splineWeights(py, sy, w[1], knot[1]) || !splineWeights(pz, sz, w[2], knot[2])) return 0.0f;
    float result = 0.0f;
    for (int z = -1; z <= 2; ++z) {
        float wz = w[2][z + 1];
        for (int y = -1; y <= 2; ++y) {
            float wyz = w[1][y + 1] * wz;
            for (int x = -1; x <= 2; ++x) {
                float wxyz = w[0][x + 1] * wyz;
                if (wxyz == 0) continue;
                size_t pos = ((size_t)(knot[2] + z) * sy + (knot[1] + y)) * sx + knot[0] + x;
                result += values[pos] * wxyz;
            }
        }
    }
    return result;
}

// RoughTransmittance::Evaluate / EvaluateDiffuse (Engine/RoughTransmittance.cu:55-119); the table of slot `type`
// (RoughTransmittanceManager indexes its three objects with the distribution type, :139-157)
inline float roughTransmittance(const DG& dg, unsigned type, float cosTheta, float alpha, float eta) {
    if (!dg.rough_transmittance) throw std::runtime_error("oracle: roughplastic needs ctl_scene_desc::rough_transmittance");
    const ctl_rough_transmittance& T = dg.rough_transmittance[type];
    float warpedCosTheta = mpow(fabsf(cosTheta), 0.25f), result;
    if (cosTheta < 0) { cosTheta = -cosTheta; eta = 1.0f / eta; }
    const float* data = T.trans;
    if (eta < 1) { data += (size_t)T.eta_samples * T.alpha_samples * T.theta_samples; eta = 1.0f / eta; }
    if (eta < T.eta_min) eta = T.eta_min;
    float warpedAlpha = mpow((alpha - T.alpha_min) / (T.alpha_max - T.alpha_min), 0.25f);
    float warpedEta = mpow((eta - T.eta_min) / (T.eta_max - T.eta_min), 0.25f);
    result = evalCubicInterp3D(warpedCosTheta, warpedAlpha, warpedEta, data, T.theta_samples, T.alpha_samples, T.eta_samples);
    return fmin2(1.0f, fmax2(0.0f, result));
}
inline float roughTransmittanceDiffuse(const DG& dg, unsigned type, float alpha, float eta) {
    if (!dg.rough_transmittance) throw std::runtime_error("oracle: roughplastic needs ctl_scene_desc::rough_transmittance");
    const ctl_rough_transmittance& T = dg.rough_transmittance[type];
    const float* data = T.diff_trans;
    if (eta < 1) { data += (size_t)T.eta_samples * T.alpha_samples; eta = 1.0f / eta; }
    if (eta < T.eta_min) eta = T.eta_min;
    float warpedAlpha = mpow((alpha - T.alpha_min) / (T.alpha_max - T.alpha_min), 0.25f);
    float warpedEta = mpow((eta - T.eta_min) / (T.eta_max - T.eta_min), 0.25f);
    float result = evalCubicInterp2D(warpedAlpha, warpedEta, data, T.alpha_samples, T.eta_samples);
    return fmin2(1.0f, fmax2(0.0f, result));
}

inline Spec bsdf3F(const ctl_material& M, const BRec& bRec, int measure);
inline float bsdf3Pdf(const ctl_material& M, const BRec& bRec, int measure);

// roughplastic: MicrofacetDistribution(type, alpha, sampleVisible) with isotropic alpha (BSDF_Simple.cu:901-905)
inline Microfacet roughplasticDistr(const ctl_material& M, const BRec& bRec) {
    float a = avg3(texEval(M.tex[2], bRec.dg));
    return Microfacet((int)M.u[2], a, a, M.u[1] != 0);
}
inline float roughplasticProbSpecular(const ctl_material& M, const BRec& bRec, const Microfacet& distr) {
    float probSpecular = 1 - roughTransmittance(bRec.dg, M.u[2], Frame::cosTheta(bRec.wi), distr.alphaU, M.f[0]);
    return (probSpecular * M.f[2]) / (probSpecular * M.f[2] + (1 - probSpecular) * (1 - M.f[2]));
}

inline Spec bsdf3Sample(const ctl_material& M, BRec& bRec, float& pdf, V2 _sample) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIFFUSE: {   // BSDF_Simple.h:42-49
        bRec.wo = squareToCosineHemisphere(_sample);
        bRec.eta = 1.0f;
        bRec.sampledType = CTL_EGlossyReflection;
        pdf = squareToCosineHemispherePdf(bRec.wo);
        return sdiv(bsdf3F(M, bRec, ESolidAngle), pdf);
    }
    case CTL_BSDF_WARD: {   // BSDF_Simple.cu:1173-1230
        V2 sample = _sample;
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if (!hasSpecular && !hasDiffuse) return Spec(0.0f);
        bool choseSpecular = hasSpecular; const float ssw = M.f[0];
        if (hasDiffuse && hasSpecular) {
            if (sample.x <= ssw) sample.x /= ssw;
            else { sample.x = (sample.x - ssw) / (1 - ssw); choseSpecular = false; }
        }
        if (choseSpecular) {
            float alphaU = avg3(texEval(M.tex[2], bRec.dg)), alphaV = avg3(texEval(M.tex[3], bRec.dg));
            float phiH = matan(alphaV / alphaU * mtan(2.0f * PI * sample.y));
            if (sample.y > 0.5f) phiH += PI;
            float cosPhiH = mcos(phiH);
            float sinPhiH = safe_sqrt(1.0f - cosPhiH * cosPhiH);
            float thetaH = matan(safe_sqrt(-mlog(sample.x) / ((cosPhiH * cosPhiH) / (alphaU * alphaU) + (sinPhiH * sinPhiH) / (alphaV * alphaV))));
            float sinTheta = msin(thetaH), cosTheta = mcos(thetaH), sinPhi = msin(phiH), cosPhi = mcos(phiH);   // Warp::SphericalDirection (Warp.h:204-216)
            V3 H(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
            bRec.wo = reflectAbout(bRec.wi, H);
            bRec.sampledType = CTL_EGlossyReflection;
            if (Frame::cosTheta(bRec.wo) <= 0.0f) return Spec(0.0f);
        } else {
            bRec.wo = squareToCosineHemisphere(sample);
            bRec.sampledType = CTL_EDiffuseReflection;
        }
        bRec.eta = 1.0f;
        pdf = bsdf3Pdf(M, bRec, ESolidAngle);
        if (pdf == 0) return Spec(0.0f);
        return sdiv(bsdf3F(M, bRec, ESolidAngle), pdf);
    }
    case CTL_BSDF_ROUGHPLASTIC: {   // BSDF_Simple.cu:890-946
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if (Frame::cosTheta(bRec.wi) <= 0 || (!hasSpecular && !hasDiffuse)) return Spec(0.0f);
        bool choseSpecular = hasSpecular;
        V2 sample = _sample;
        Microfacet distr = roughplasticDistr(M, bRec);
        if (hasSpecular && hasDiffuse) {
            float probSpecular = roughplasticProbSpecular(M, bRec, distr);
            if (sample.y < probSpecular) sample.y /= probSpecular;
            else { sample.y = (sample.y - probSpecular) / (1 - probSpecular); choseSpecular = false; }
        }
        if (choseSpecular) {
            float unused;
            V3 m = distr.sample(bRec.wi, sample, unused);
            bRec.wo = reflectAbout(bRec.wi, m);
            bRec.sampledType = CTL_EGlossyReflection;
            if (Frame::cosTheta(bRec.wo) <= 0) return Spec(0.0f);
        } else {
            bRec.sampledType = CTL_EDiffuseReflection;
            bRec.wo = squareToCosineHemisphere(sample);
        }
        bRec.eta = 1.0f;
        pdf = bsdf3Pdf(M, bRec, ESolidAngle);
        if (pdf == 0) return Spec(0.0f);
        return sdiv(bsdf3F(M, bRec, ESolidAngle), pdf);
    }
    default: throw std::runtime_error("oracle: bsdf type not restated");
    }
}

inline Spec bsdf3F(const ctl_material& M, const BRec& bRec, int measure) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIFFUSE: {   // BSDF_Simple.cu:82-172
        if (!(bRec.typeMask & CTL_EGlossyReflection) || measure != ESolidAngle || Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0) return Spec(0.0f);
        const float conversionFactor = 1 / std::sqrt((float)2);
        float sigma = avg3(texEval(M.tex[1], bRec.dg)) * conversionFactor;
        const float sigma2 = sigma * sigma;
        float sinThetaI = Frame::sinTheta(bRec.wi), sinThetaO = Frame::sinTheta(bRec.wo);
        float cosPhiDiff = 0;
        if (sinThetaI > EPSILON && sinThetaO > EPSILON) {
            float sinPhiI = Frame::sinPhi(bRec.wi), cosPhiI = Frame::cosPhi(bRec.wi), sinPhiO = Frame::sinPhi(bRec.wo), cosPhiO = Frame::cosPhi(bRec.wo);
            cosPhiDiff = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
        }
        if (M.u[0]) {
            float A = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f), B = 0.45f * sigma2 / (sigma2 + 0.09f), sinAlpha, tanBeta;
            if (Frame::cosTheta(bRec.wi) > Frame::cosTheta(bRec.wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / Frame::cosTheta(bRec.wi); }
            else { sinAlpha = sinThetaI; tanBeta = sinThetaO / Frame::cosTheta(bRec.wo); }
            return texEval(M.tex[0], bRec.dg) * (INV_PI * Frame::cosTheta(bRec.wo) * (A + B * fmax2(cosPhiDiff, 0.0f) * sinAlpha * tanBeta));
        } else {
            float thetaI = safe_acos(Frame::cosTheta(bRec.wi)), thetaO = safe_acos(Frame::cosTheta(bRec.wo)), alpha = fmax2(thetaI, thetaO), beta = fmin2(thetaI, thetaO);
            float sinAlpha, sinBeta, tanBeta;
            if (Frame::cosTheta(bRec.wi) > Frame::cosTheta(bRec.wo)) { sinAlpha = sinThetaO; sinBeta = sinThetaI; tanBeta = sinThetaI / Frame::cosTheta(bRec.wi); }
            else { sinAlpha = sinThetaI; sinBeta = sinThetaO; tanBeta = sinThetaO / Frame::cosTheta(bRec.wo); }
            float tmp = sigma2 / (sigma2 + 0.09f), tmp2 = (4 * INV_PI * INV_PI) * alpha * beta, tmp3 = 2 * beta * INV_PI;
            float C1 = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f), C2 = 0.45f * tmp, C3 = 0.125f * tmp * tmp2 * tmp2, C4 = 0.17f * sigma2 / (sigma2 + 0.13f);
            if (cosPhiDiff > 0) C2 *= sinAlpha; else C2 *= sinAlpha - tmp3 * tmp3 * tmp3;
            float tanHalf = (sinAlpha + sinBeta) / (safe_sqrt(1.0f - sinAlpha * sinAlpha) + safe_sqrt(1.0f - sinBeta * sinBeta));
            Spec rho = texEval(M.tex[0], bRec.dg),
                 snglScat = rho * (C1 + cosPhiDiff * C2 * tanBeta + (1.0f - fabsf(cosPhiDiff)) * C3 * tanHalf),
                 dblScat = rho * rho * (C4 * (1.0f - cosPhiDiff * tmp3 * tmp3));
            return (snglScat + dblScat) * (INV_PI * Frame::cosTheta(bRec.wo));
        }
    }
    case CTL_BSDF_WARD: {   // BSDF_Simple.cu:1232-1276
        if (Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || measure != ESolidAngle) return Spec(0.0f);
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        Spec result(0.0f);
        if (hasSpecular) {
            V3 H = bRec.wi + bRec.wo;
            float alphaU = avg3(texEval(M.tex[2], bRec.dg)), alphaV = avg3(texEval(M.tex[3], bRec.dg));
            float factor1 = 0.0f;
            switch (M.u[0]) {
            case 0: factor1 = 1.0f / (4.0f * PI * alphaU * alphaV * std::sqrt(Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo))); break;
            case 1: factor1 = 1.0f / (4.0f * PI * alphaU * alphaV * Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo)); break;
            case 2: factor1 = dot(H, H) / (PI * alphaU * alphaV * mpow(Frame::cosTheta(normalize(H)), 4)); break;
            }
            float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
            float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
            float specRef = factor1 * mexp(exponent);
            if (specRef > 1e-10f) result = result + texEval(M.tex[1], bRec.dg) * specRef;
        }
        if (hasDiffuse) result = result + texEval(M.tex[0], bRec.dg) * INV_PI;
        return result * Frame::cosTheta(bRec.wo);
    }
    case CTL_BSDF_ROUGHPLASTIC: {   // BSDF_Simple.cu:948-1005
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if (measure != ESolidAngle || Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || (!hasSpecular && !hasDiffuse)) return Spec(0.0f);
        Microfacet distr = roughplasticDistr(M, bRec);
        Spec result(0.0f);
        if (hasSpecular) {
            const V3 H = normalize(bRec.wo + bRec.wi);
            const float D = distr.eval(H);
            const float F = fresnelDielectricExt(dot(bRec.wi, H), M.f[0]);
            const float G = distr.G(bRec.wi, bRec.wo, H);
            float value = F * D * G / (4.0f * Frame::cosTheta(bRec.wi));
            result = result + texEval(M.tex[1], bRec.dg) * value;
        }
        if (hasDiffuse) {
            Spec diff = texEval(M.tex[0], bRec.dg);
            float T12 = roughTransmittance(bRec.dg, M.u[2], Frame::cosTheta(bRec.wi), distr.alphaU, M.f[0]);
            float T21 = roughTransmittance(bRec.dg, M.u[2], Frame::cosTheta(bRec.wo), distr.alphaU, M.f[0]);
            float Fdr = 1 - roughTransmittanceDiffuse(bRec.dg, M.u[2], distr.alphaU, M.f[0]);
            if (M.u[0]) diff = diff / (Spec(1.0f) - diff * Fdr);
            else diff = sdiv(diff, 1 - Fdr);
            result = result + diff * (INV_PI * Frame::cosTheta(bRec.wo) * T12 * T21 * M.f[1]);
        }
        return result;
    }
    default: throw std::runtime_error("oracle: bsdf type not restated");
    }
}

inline float bsdf3Pdf(const ctl_material& M, const BRec& bRec, int measure) {
    switch (M.bsdf_type) {
    case CTL_BSDF_ROUGHDIFFUSE:   // BSDF_Simple.h:51-59
        if (!(bRec.typeMask & CTL_EGlossyReflection) || measure != ESolidAngle || Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0) return 0.0f;
        return squareToCosineHemispherePdf(bRec.wo);
    case CTL_BSDF_WARD: {   // BSDF_Simple.cu:1278-1313
        if (Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || measure != ESolidAngle) return 0.0f;
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        float diffuseProb = 0.0f, specProb = 0.0f; const float ssw = M.f[0];
        if (hasSpecular) {
            float alphaU = avg3(texEval(M.tex[2], bRec.dg)), alphaV = avg3(texEval(M.tex[3], bRec.dg));
            V3 H = normalize(bRec.wi + bRec.wo);
            float factor1 = 1.0f / (4.0f * PI * alphaU * alphaV * dot(H, bRec.wi) * mpow(Frame::cosTheta(H), 3));
            float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
            float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
            specProb = factor1 * mexp(exponent);
        }
        if (hasDiffuse) diffuseProb = squareToCosineHemispherePdf(bRec.wo);
        if (hasDiffuse && hasSpecular) return ssw * specProb + (1 - ssw) * diffuseProb;
        else if (hasDiffuse) return diffuseProb;
        else if (hasSpecular) return specProb;
        return 0.0f;
    }
    case CTL_BSDF_ROUGHPLASTIC: {   // BSDF_Simple.cu:1007-1057
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if (measure != ESolidAngle || Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || (!hasSpecular && !hasDiffuse)) return 0.0f;
        Microfacet distr = roughplasticDistr(M, bRec);
        const V3 H = normalize(bRec.wo + bRec.wi);
        float probDiffuse, probSpecular;
        if (hasSpecular && hasDiffuse) { probSpecular = roughplasticProbSpecular(M, bRec, distr); probDiffuse = 1 - probSpecular; }
        else probDiffuse = probSpecular = 1.0f;
        float result = 0.0f;
        if (hasSpecular) {
            const float dwh_dwo = 1.0f / (4.0f * dot(bRec.wo, H));
            const float prob = distr.pdf(bRec.wi, H);
            result = prob * dwh_dwo * probSpecular;
        }
        if (hasDiffuse) result += probDiffuse * squareToCosineHemispherePdf(bRec.wo);
        return result;
    }
    default: throw std::runtime_error("oracle: bsdf type not restated");
    }
}

} // namespace orc
