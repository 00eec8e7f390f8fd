// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// obsdf2.h — thindielectric, roughdielectric, plastic, phong (SceneTypes/BSDF_Simple.cu); other types go on to obsdf3.h.  Included by ocore.h after
// BRec / Microfacet / texEval are defined.  PARITY UNPINNED: BSDF_Simple.cu cannot be built here (curand_kernel.h).
#pragma once

namespace orc {

inline V3 refractAbout(V3 wi, V3 n, float eta, float cosThetaT) {   // FresnelHelper.h:149-155
    if (cosThetaT < 0) eta = 1.0f / eta;
    return n * (dot(wi, n) * eta + cosThetaT) - wi * eta;
}
inline float signum(float v) { return copysign_bits(1.0f, v); }   // MathFunc.h:116-119
inline Spec plasticDiff(const ctl_material& M, const BRec& bRec) {
    Spec diff = texEval(M.tex[0], bRec.dg);
    if (M.u[0]) return diff / (Spec(1.0f) - diff * M.f[0]);
    return sdiv(diff, 1 - M.f[0]);
}

inline Spec bsdf2F(const ctl_material& M, const BRec& bRec, int measure);
inline float bsdf2Pdf(const ctl_material& M, const BRec& bRec, int measure);

inline Spec bsdf2Sample(const ctl_material& M, BRec& bRec, float& pdf, V2 _sample) {
    switch (M.bsdf_type) {
    case CTL_BSDF_THINDIELECTRIC: {   // BSDF_Simple.cu:330-371; transmit(wi) = -wi (BSDF_Simple.h:118-121)
        bool sampleReflection = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleTransmission = (bRec.typeMask & CTL_ENull) != 0;
        float R = fresnelDielectricExt(fabsf(Frame::cosTheta(bRec.wi)), M.f[0]), T = 1 - R;
        if (R < 1) R += T * T * R / (1 - R * R);
        if (sampleTransmission && sampleReflection) {
            if (_sample.x <= R) { bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); bRec.eta = 1.0f; pdf = R; return texEval(M.tex[1], bRec.dg); }
            bRec.sampledType = CTL_ENull; bRec.wo = -bRec.wi; bRec.eta = 1.0f; pdf = 1 - R; return texEval(M.tex[0], bRec.dg);
        } else if (sampleReflection) { bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); bRec.eta = 1.0f; pdf = 1.0f; return texEval(M.tex[1], bRec.dg) * R; }
        else if (sampleTransmission) { bRec.sampledType = CTL_ENull; bRec.wo = -bRec.wi; bRec.eta = 1.0f; pdf = 1.0f; return texEval(M.tex[0], bRec.dg) * (1 - R); }
        return Spec(0.0f);
    }
    case CTL_BSDF_ROUGHDIELECTRIC: {   // BSDF_Simple.cu:503-615
        V2 sample = _sample;
        bool hasReflection = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasTransmission = (bRec.typeMask & CTL_EGlossyTransmission) != 0, sampleReflection = hasReflection;
        if (!hasReflection && !hasTransmission) return Spec(0.0f);
        const float m_eta = M.f[0], m_invEta = M.f[1]; const bool vis = M.u[1] != 0;
        Microfacet distr((int)M.u[0], avg3(texEval(M.tex[2], bRec.dg)), avg3(texEval(M.tex[3], bRec.dg)), vis);
        Microfacet sampleDistr = distr;
        if (!vis) { float sc = 1.2f - 0.2f * std::sqrt(fabsf(Frame::cosTheta(bRec.wi))); sampleDistr.scaleAlpha(sc); }
        float microfacetPDF;
        float sign = signum(Frame::cosTheta(bRec.wi));
        const V3 m = sampleDistr.sample(sign < 0 ? -bRec.wi : bRec.wi, sample, microfacetPDF);
        if (microfacetPDF == 0) return Spec(0.0f);
        pdf = microfacetPDF;
        float cosThetaT;
        float F = fresnelDielectricExt(dot(bRec.wi, m), cosThetaT, m_eta);
        Spec weight(1.0f);
        const unsigned N_REUSE = 10; const unsigned slot = (unsigned)(int)(sample.x * N_REUSE);   // MonteCarlo::sampleReuse(N, pdf, slot) (MonteCarlo.cu:16-20)
        float sample_z = slot / (float)N_REUSE;
        if (hasReflection && hasTransmission) { if (sample_z > F) { sampleReflection = false; pdf *= 1 - F; } else pdf *= F; }
        else weight = weight * (hasReflection ? F : (1 - F));
        float dwh_dwo;
        if (sampleReflection) {
            bRec.wo = reflectAbout(bRec.wi, m); bRec.eta = 1.0f; bRec.sampledType = CTL_EGlossyReflection;
            if (Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) <= 0) return Spec(0.0f);
            weight = weight * texEval(M.tex[1], bRec.dg);
            dwh_dwo = 1.0f / (4.0f * dot(bRec.wo, m));
        } else {
            if (cosThetaT == 0) return Spec(0.0f);
            bRec.wo = normalize(refractAbout(bRec.wi, m, m_eta, cosThetaT));
            bRec.eta = cosThetaT < 0 ? m_eta : m_invEta; bRec.sampledType = CTL_EGlossyTransmission;
            if (Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) >= 0) return Spec(0.0f);
            float factor = (cosThetaT < 0 ? m_invEta : m_eta);
            weight = weight * (texEval(M.tex[0], bRec.dg) * (factor * factor));
            float sqrtDenom = dot(bRec.wi, m) + bRec.eta * dot(bRec.wo, m);
            dwh_dwo = (bRec.eta * bRec.eta * dot(bRec.wo, m)) / (sqrtDenom * sqrtDenom);
        }
        if (vis) weight = weight * distr.smithG1(bRec.wo, m);
        else weight = weight * fabsf(distr.eval(m) * distr.G(bRec.wi, bRec.wo, m) * dot(bRec.wi, m) / (microfacetPDF * Frame::cosTheta(bRec.wi)));
        pdf *= fabsf(dwh_dwo);
        return weight;
    }
    case CTL_BSDF_PLASTIC: {   // BSDF_Simple.cu:765-826
        bool hasSpecular = (bRec.typeMask & CTL_EDeltaReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if ((!hasDiffuse && !hasSpecular) || Frame::cosTheta(bRec.wi) <= 0) return Spec(0.0f);
        const float m_eta = M.f[2], m_invEta2 = M.f[3], ssw = M.f[4];
        float Fi = fresnelDielectricExt(Frame::cosTheta(bRec.wi), m_eta);
        bRec.eta = 1.0f;
        if (hasDiffuse && hasSpecular) {
            float probSpecular = (Fi * ssw) / (Fi * ssw + (1 - Fi) * (1 - ssw));
            if (_sample.x < probSpecular) { bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); pdf = probSpecular; return sdiv(texEval(M.tex[1], bRec.dg) * Fi, probSpecular); }
            bRec.sampledType = CTL_EDiffuseReflection;
            bRec.wo = squareToCosineHemisphere(V2{ (_sample.x - probSpecular) / (1 - probSpecular), _sample.y });
            float Fo = fresnelDielectricExt(Frame::cosTheta(bRec.wo), m_eta);
            Spec diff = plasticDiff(M, bRec);
            pdf = (1 - probSpecular) * squareToCosineHemispherePdf(bRec.wo);
            return diff * (m_invEta2 * (1 - Fi) * (1 - Fo) / (1 - probSpecular));
        } else if (hasSpecular) { bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); pdf = 1; return texEval(M.tex[1], bRec.dg) * Fi; }
        bRec.sampledType = CTL_EDiffuseReflection; bRec.wo = squareToCosineHemisphere(_sample);
        float Fo = fresnelDielectricExt(Frame::cosTheta(bRec.wo), m_eta);
        Spec diff = plasticDiff(M, bRec);
        pdf = squareToCosineHemispherePdf(bRec.wo);
        return diff * (m_invEta2 * (1 - Fi) * (1 - Fo));
    }
    case CTL_BSDF_PHONG: {   // BSDF_Simple.cu:1059-1111
        V2 sample = _sample;
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if (!hasSpecular && !hasDiffuse) return Spec(0.0f);
        bool choseSpecular = hasSpecular; const float ssw = M.f[0];
        if (hasDiffuse && hasSpecular) { if (sample.x <= ssw) sample.x /= ssw; else { sample.x = (sample.x - ssw) / (1 - ssw); choseSpecular = false; } }
        if (choseSpecular) {
            V3 R = Frame::reflect(bRec.wi);
            float exponent = avg3(texEval(M.tex[2], bRec.dg));
            float sinAlpha = std::sqrt(1 - mpow(sample.y, 2 / (exponent + 1))), cosAlpha = mpow(sample.y, 1 / (exponent + 1)), phi = (2.0f * PI) * sample.x;
            V3 localDir(sinAlpha * mcos(phi), sinAlpha * msin(phi), cosAlpha);
            bRec.wo = normalize(Frame(R).toWorld(localDir)); bRec.sampledType = CTL_EGlossyReflection;
            if (Frame::cosTheta(bRec.wo) <= 0) return Spec(0.0f);
        } else { bRec.wo = squareToCosineHemisphere(sample); bRec.sampledType = CTL_EDiffuseReflection; }
        bRec.eta = 1.0f;
        pdf = bsdf2Pdf(M, bRec, ESolidAngle);
        if (pdf == 0) return Spec(0.0f);
        return sdiv(bsdf2F(M, bRec, ESolidAngle), pdf);
    }
    default: return bsdf3Sample(M, bRec, pdf, _sample);
    }
}

inline Spec bsdf2F(const ctl_material& M, const BRec& bRec, int measure) {
    switch (M.bsdf_type) {
    case CTL_BSDF_THINDIELECTRIC: return Spec(0.0f);   // delta lobes only (BSDF_Simple.cu:304-328)
    case CTL_BSDF_ROUGHDIELECTRIC: {   // BSDF_Simple.cu:436-501
        if (measure != ESolidAngle) return Spec(0.0f);
        const float m_eta = M.f[0], m_invEta = M.f[1];
        bool reflect = Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) > 0;
        V3 H;
        if (reflect) { if (!(bRec.typeMask & CTL_EGlossyReflection)) return Spec(0.0f); H = normalize(bRec.wo + bRec.wi); }
        else { if (!(bRec.typeMask & CTL_EGlossyTransmission)) return Spec(0.0f); float eta = Frame::cosTheta(bRec.wi) > 0 ? m_eta : m_invEta; H = normalize(bRec.wi + bRec.wo * eta); }
        H = H * signum(Frame::cosTheta(H));
        Microfacet distr((int)M.u[0], avg3(texEval(M.tex[2], bRec.dg)), avg3(texEval(M.tex[3], bRec.dg)), M.u[1] != 0);
        const float D = distr.eval(H);
        if (D == 0) return Spec(0.0f);
        const float F = fresnelDielectricExt(dot(bRec.wi, H), m_eta);
        const float G = distr.G(bRec.wi, bRec.wo, H);
        if (reflect) { float value = F * D * G / (4.0f * fabsf(Frame::cosTheta(bRec.wi))); return texEval(M.tex[1], bRec.dg) * value; }
        float eta = Frame::cosTheta(bRec.wi) > 0.0f ? m_eta : m_invEta;
        float sqrtDenom = dot(bRec.wi, H) + eta * dot(bRec.wo, H);
        float value = ((1 - F) * D * G * eta * eta * dot(bRec.wi, H) * dot(bRec.wo, H)) / (Frame::cosTheta(bRec.wi) * sqrtDenom * sqrtDenom);
        float factor = (Frame::cosTheta(bRec.wi) > 0 ? m_invEta : m_eta);
        return texEval(M.tex[0], bRec.dg) * fabsf(value * factor * factor);
    }
    case CTL_BSDF_PLASTIC: {   // BSDF_Simple.cu:828-858: with solid-angle measure only the diffuse lobe has a density
        bool hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) && measure == ESolidAngle;
        if (Frame::cosTheta(bRec.wo) <= 0 || Frame::cosTheta(bRec.wi) <= 0) return Spec(0.0f);
        float Fi = fresnelDielectricExt(Frame::cosTheta(bRec.wi), M.f[2]);
        if (hasDiffuse) {
            float Fo = fresnelDielectricExt(Frame::cosTheta(bRec.wo), M.f[2]);
            Spec diff = plasticDiff(M, bRec);
            return diff * (squareToCosineHemispherePdf(bRec.wo) * M.f[3] * (1 - Fi) * (1 - Fo));
        }
        return Spec(0.0f);
    }
    case CTL_BSDF_PHONG: {   // BSDF_Simple.cu:1113-1139
        if (Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || measure != ESolidAngle) return Spec(0.0f);
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        Spec result(0.0f);
        if (hasSpecular) {
            float alpha = dot(bRec.wo, Frame::reflect(bRec.wi)), exponent = avg3(texEval(M.tex[2], bRec.dg));
            if (alpha > 0.0f) result = result + texEval(M.tex[1], bRec.dg) * ((exponent + 2) * INV_TWOPI * mpow(alpha, exponent));
        }
        if (hasDiffuse) result = result + texEval(M.tex[0], bRec.dg) * INV_PI;
        return result * Frame::cosTheta(bRec.wo);
    }
    default: return bsdf3F(M, bRec, measure);
    }
}

inline float bsdf2Pdf(const ctl_material& M, const BRec& bRec, int measure) {
    switch (M.bsdf_type) {
    case CTL_BSDF_THINDIELECTRIC: return 0.0f;
    case CTL_BSDF_ROUGHDIELECTRIC: {   // BSDF_Simple.cu:373-434
        if (measure != ESolidAngle) return 0.0f;
        const float m_eta = M.f[0], m_invEta = M.f[1]; const bool vis = M.u[1] != 0;
        bool hasReflection = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasTransmission = (bRec.typeMask & CTL_EGlossyTransmission) != 0,
             reflect = Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) > 0;
        V3 H; float dwh_dwo;
        if (reflect) { if (!(bRec.typeMask & CTL_EGlossyReflection)) return 0.0f; H = normalize(bRec.wo + bRec.wi); dwh_dwo = 1.0f / (4.0f * dot(bRec.wo, H)); }
        else {
            if (!(bRec.typeMask & CTL_EGlossyTransmission)) return 0.0f;
            float eta = Frame::cosTheta(bRec.wi) > 0 ? m_eta : m_invEta;
            H = normalize(bRec.wi + bRec.wo * eta);
            float sqrtDenom = dot(bRec.wi, H) + eta * dot(bRec.wo, H);
            dwh_dwo = (eta * eta * dot(bRec.wo, H)) / (sqrtDenom * sqrtDenom);
        }
        H = H * signum(Frame::cosTheta(H));
        Microfacet sampleDistr((int)M.u[0], avg3(texEval(M.tex[2], bRec.dg)), avg3(texEval(M.tex[3], bRec.dg)), vis);
        if (!vis) { float sc = 1.2f - 0.2f * std::sqrt(fabsf(Frame::cosTheta(bRec.wi))); sampleDistr.scaleAlpha(sc); }
        float sign = signum(Frame::cosTheta(bRec.wi));
        float prob = sampleDistr.pdf(sign < 0 ? -bRec.wi : bRec.wi, H);
        if (hasTransmission && hasReflection) { float F = fresnelDielectricExt(dot(bRec.wi, H), m_eta); prob *= reflect ? F : (1 - F); }
        return fabsf(prob * dwh_dwo);
    }
    case CTL_BSDF_PLASTIC: {   // BSDF_Simple.cu:860-888
        bool hasSpecular = (bRec.typeMask & CTL_EDeltaReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if (Frame::cosTheta(bRec.wo) <= 0 || Frame::cosTheta(bRec.wi) <= 0) return 0.0f;
        float probSpecular = hasSpecular ? 1.0f : 0.0f;
        if (hasSpecular && hasDiffuse) { float Fi = fresnelDielectricExt(Frame::cosTheta(bRec.wi), M.f[2]); probSpecular = (Fi * M.f[4]) / (Fi * M.f[4] + (1 - Fi) * (1 - M.f[4])); }
        if (hasSpecular && measure == EDiscrete) return 0.0f;   // only reached with the discrete measure, which this path never asks for
        else if (hasDiffuse && measure == ESolidAngle) return squareToCosineHemispherePdf(bRec.wo) * (1 - probSpecular);
        return 0.0f;
    }
    case CTL_BSDF_PHONG: {   // BSDF_Simple.cu:1141-1171
        if (Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || measure != ESolidAngle) return 0.0f;
        bool hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        float diffuseProb = 0.0f, specProb = 0.0f; const float ssw = M.f[0];
        if (hasDiffuse) diffuseProb = squareToCosineHemispherePdf(bRec.wo);
        if (hasSpecular) { float alpha = dot(bRec.wo, Frame::reflect(bRec.wi)), exponent = avg3(texEval(M.tex[2], bRec.dg)); if (alpha > 0) specProb = mpow(alpha, exponent) * (exponent + 1.0f) / (2.0f * PI); }
        if (hasDiffuse && hasSpecular) return ssw * specProb + (1 - ssw) * diffuseProb;
        else if (hasDiffuse) return diffuseProb;
        else if (hasSpecular) return specProb;
        return 0.0f;
    }
    default: return bsdf3Pdf(M, bRec, measure);
    }
}

} // namespace orc
