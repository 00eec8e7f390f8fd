// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// obsdf4.h — the nesting BSDFs coating, roughcoating, blend (SceneTypes/BSDF_Complex.cu) and the discrete-measure f / pdf of the
// delta models they may wrap (dielectric, thindielectric, conductor, plastic; SceneTypes/BSDF_Simple.cu).  Included by ocore.h
// after bsdfSample / bsdfF / bsdfPdf are declared.  A nested BSDF is another entry of the scene's material array, named by its
// absolute index (coating / roughcoating: u[2]; blend: u[2], u[3]); BRec::dg.materials is that array.
// PARITY UNPINNED: BSDF_Complex.cu cannot be built here (curand_kernel.h through BSDF.h).
#pragma once

namespace orc {

inline Spec bsdfSample(const ctl_material& M, BRec& bRec, float& pdf, V2 _sample);
inline Spec bsdfF(const ctl_material& M, const BRec& bRec, int measure);
inline float bsdfPdf(const ctl_material& M, const BRec& bRec, int measure);

inline int bsdfMeasure(unsigned componentType) {   // BSDF::getMeasure (SceneTypes/BSDF.h:66-80)
    if (componentType & ESmooth) return ESolidAngle;
    if (componentType & EDelta) return EDiscrete;
    if (componentType & EDelta1D) return ELength;
    return ESolidAngle;
}
inline const ctl_material& nestedMat(const ctl_material& M, const BRec& bRec, int which) {
    if (!bRec.dg.materials) throw std::runtime_error("oracle: nested BSDF without a material array");
    return bRec.dg.materials[M.u[2 + which]];
}

// ---- f / pdf with the discrete measure (delta lobes)
inline Spec bsdfFDiscrete(const ctl_material& M, const BRec& bRec) {
    switch (M.bsdf_type) {
    case CTL_BSDF_DIELECTRIC: {   // BSDF_Simple.cu:226-252 (dispersion off: f_o = 1, eta = Cauchy B)
        bool sampleReflection = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleTransmission = (bRec.typeMask & CTL_EDeltaTransmission) != 0;
        float eta = M.f[0] + M.f[1] / (600 / 1e3f), invEta = 1.0f / eta, cosThetaT;   // Cauchy B + C / lambda^2 as in bsdfSample
        float F = fresnelDielectricExt(Frame::cosTheta(bRec.wi), cosThetaT, eta);
        if (Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) >= 0) {
            if (!sampleReflection || fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) > DeltaEpsilon) return Spec(0.0f);
            return texEval(M.tex[1], bRec.dg) * F;
        }
        if (!sampleTransmission || fabsf(dot(Frame::refract(bRec.wi, cosThetaT, eta, invEta), bRec.wo) - 1) > DeltaEpsilon) return Spec(0.0f);
        float factor = cosThetaT < 0 ? invEta : eta;   // ERadiance
        return Spec(1.0f) * texEval(M.tex[0], bRec.dg) * factor * factor * (1 - F);
    }
    case CTL_BSDF_THINDIELECTRIC: {   // BSDF_Simple.cu:304-328
        bool sampleReflection = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleTransmission = (bRec.typeMask & CTL_ENull) != 0;
        float R = fresnelDielectricExt(fabsf(Frame::cosTheta(bRec.wi)), M.f[0]), T = 1 - R;
        if (R < 1) R += T * T * R / (1 - R * R);
        if (Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) >= 0) {
            if (!sampleReflection || fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) > DeltaEpsilon) return Spec(0.0f);
            return texEval(M.tex[1], bRec.dg) * R;
        }
        if (!sampleTransmission || fabsf(dot(-bRec.wi, bRec.wo) - 1) > DeltaEpsilon) return Spec(0.0f);
        return texEval(M.tex[0], bRec.dg) * (1 - R);
    }
    case CTL_BSDF_CONDUCTOR: {   // BSDF_Simple.cu:632-646
        if (!(bRec.typeMask & CTL_EDeltaReflection) || Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) > DeltaEpsilon) return Spec(0.0f);
        return texEval(M.tex[0], bRec.dg) * fresnelConductorExact(Frame::cosTheta(bRec.wi), Spec(M.f[0], M.f[1], M.f[2]), Spec(M.f[3], M.f[4], M.f[5]));
    }
    case CTL_BSDF_PLASTIC: {   // BSDF_Simple.cu:828-858, hasSpecular branch
        if (!(bRec.typeMask & CTL_EDeltaReflection) || Frame::cosTheta(bRec.wo) <= 0 || Frame::cosTheta(bRec.wi) <= 0) return Spec(0.0f);
        float Fi = fresnelDielectricExt(Frame::cosTheta(bRec.wi), M.f[2]);
        if (fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) < DeltaEpsilon) return texEval(M.tex[1], bRec.dg) * Fi;
        return Spec(0.0f);
    }
    default: return Spec(0.0f);   // every other simple model returns 0 unless measure == ESolidAngle
    }
}
inline float bsdfPdfDiscrete(const ctl_material& M, const BRec& bRec) {
    switch (M.bsdf_type) {
    case CTL_BSDF_DIELECTRIC: {   // BSDF_Simple.cu:254-277 (eta_pdf = 1 without dispersion)
        bool sampleReflection = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleTransmission = (bRec.typeMask & CTL_EDeltaTransmission) != 0;
        float eta = M.f[0] + M.f[1] / (600 / 1e3f), invEta = 1.0f / eta, cosThetaT;   // Cauchy B + C / lambda^2 as in bsdfSample
        float F = fresnelDielectricExt(Frame::cosTheta(bRec.wi), cosThetaT, eta);
        if (Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) >= 0) {
            if (!sampleReflection || fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) > DeltaEpsilon) return 0.0f;
            return sampleTransmission ? 1.0f * F : 1.0f;
        }
        if (!sampleTransmission || fabsf(dot(Frame::refract(bRec.wi, cosThetaT, eta, invEta), bRec.wo) - 1) > DeltaEpsilon) return 0.0f;
        return sampleReflection ? 1 - F : 1.0f * 1.0f;
    }
    case CTL_BSDF_THINDIELECTRIC: {   // BSDF_Simple.cu:279-302
        bool sampleReflection = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleTransmission = (bRec.typeMask & CTL_ENull) != 0;
        float R = fresnelDielectricExt(fabsf(Frame::cosTheta(bRec.wi)), M.f[0]), T = 1 - R;
        if (R < 1) R += T * T * R / (1 - R * R);
        if (Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) >= 0) {
            if (!sampleReflection || fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) > DeltaEpsilon) return 0.0f;
            return sampleTransmission ? R : 1.0f;
        }
        if (!sampleTransmission || fabsf(dot(-bRec.wi, bRec.wo) - 1) > DeltaEpsilon) return 0.0f;
        return sampleReflection ? 1 - R : 1.0f;
    }
    case CTL_BSDF_CONDUCTOR:   // BSDF_Simple.cu:648-660
        if (!(bRec.typeMask & CTL_EDeltaReflection) || Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0 || fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) > DeltaEpsilon) return 0.0f;
        return 1.0f;
    case CTL_BSDF_PLASTIC: {   // BSDF_Simple.cu:860-888
        bool hasSpecular = (bRec.typeMask & CTL_EDeltaReflection) != 0, hasDiffuse = (bRec.typeMask & CTL_EDiffuseReflection) != 0;
        if (Frame::cosTheta(bRec.wo) <= 0 || Frame::cosTheta(bRec.wi) <= 0) return 0.0f;
        float probSpecular = hasSpecular ? 1.0f : 0.0f;
        if (hasSpecular && hasDiffuse) { float Fi = fresnelDielectricExt(Frame::cosTheta(bRec.wi), M.f[2]); probSpecular = (Fi * M.f[4]) / (Fi * M.f[4] + (1 - Fi) * (1 - M.f[4])); }
        if (hasSpecular && fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) < DeltaEpsilon) return probSpecular;
        return 0.0f;
    }
    default: return 0.0f;
    }
}

// ---- coating (BSDF_Complex.h:9-75, BSDF_Complex.cu:6-157)
inline V3 coatRefractIn(const ctl_material& M, V3 wi, float& R) {
    float cosThetaT; R = fresnelDielectricExt(fabsf(Frame::cosTheta(wi)), cosThetaT, M.f[0]);
    return normalize(V3(M.f[1] * wi.x, M.f[1] * wi.y, -signum(Frame::cosTheta(wi)) * cosThetaT));
}
inline V3 coatRefractOut(const ctl_material& M, V3 wi, float& R) {
    float cosThetaT; R = fresnelDielectricExt(fabsf(Frame::cosTheta(wi)), cosThetaT, M.f[1]);
    return normalize(V3(M.f[0] * wi.x, M.f[0] * wi.y, -signum(Frame::cosTheta(wi)) * cosThetaT));
}
inline Spec specExp(Spec s) { return Spec(mexp(s.x), mexp(s.y), mexp(s.z)); }
inline float coatProbSpecular(const ctl_material& M, float R12) { return (R12 * M.f[3]) / (R12 * M.f[3] + (1 - R12) * (1 - M.f[3])); }

// ---- roughcoating (BSDF_Complex.h:77-147, BSDF_Complex.cu:159-342)
inline V3 roughcoatRefractTo(const ctl_material& M, bool interior, V3 wi) {
    float cosThetaI = Frame::cosTheta(wi);
    float invEta = interior ? M.f[1] : M.f[0];
    bool entering = cosThetaI > 0.0f;
    float sinThetaTSqr = invEta * invEta * Frame::sinTheta2(wi);
    if (sinThetaTSqr >= 1.0f) return V3(0.0f);
    float cosThetaT = std::sqrt(1.0f - sinThetaTSqr);
    return normalize(V3(invEta * wi.x, invEta * wi.y, entering ? cosThetaT : -cosThetaT));
}
inline Microfacet roughcoatDistr(const ctl_material& M, const BRec& bRec) { float a = avg3(texEval(M.tex[2], bRec.dg)); return Microfacet((int)M.u[0], a, a, M.u[1] != 0); }
inline float roughcoatProbSpecular(const ctl_material& M, const BRec& bRec, const Microfacet& distr) {
    float p = 1 - roughTransmittance(bRec.dg, M.u[0], fabsf(Frame::cosTheta(bRec.wi)), distr.alphaU, M.f[0]);
    return (p * M.f[3]) / (p * M.f[3] + (1 - p) * (1 - M.f[3]));
}

inline Spec bsdfComplexF(const ctl_material& M, const BRec& bRec, int measure);
inline float bsdfComplexPdf(const ctl_material& M, const BRec& bRec, int measure);

inline Spec bsdfComplexSample(const ctl_material& M, BRec& bRec, float& pdf, V2 _sample) {
    switch (M.bsdf_type) {
    case CTL_BSDF_COATING: {   // BSDF_Complex.cu:6-82
        const ctl_material& nested = nestedMat(M, bRec, 0);
        bool sampleSpecular = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleNested = (bRec.typeMask & nested.combined_type & EAll) != 0;
        if (!sampleSpecular && !sampleNested) return Spec(0.0f);
        float R12; V3 wiPrime = coatRefractIn(M, bRec.wi, R12);
        float probSpecular = coatProbSpecular(M, R12);
        bool choseSpecular = sampleSpecular;
        V2 sample = _sample;
        if (sampleSpecular && sampleNested) {
            if (sample.x < probSpecular) sample.x /= probSpecular;
            else { sample.x = (sample.x - probSpecular) / (1 - probSpecular); choseSpecular = false; }
        }
        if (choseSpecular) {
            bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); bRec.eta = 1.0f;
            pdf = sampleNested ? probSpecular : 1.0f;
            return texEval(M.tex[1], bRec.dg) * (R12 / pdf);
        }
        if (R12 == 1.0f) return Spec(0.0f);
        V3 wiBackup = bRec.wi; bRec.wi = wiPrime;
        Spec result = bsdfSample(nested, bRec, pdf, sample);
        bRec.wi = wiBackup;
        if (isZero(result)) return Spec(0.0f);
        V3 woPrime = bRec.wo;
        Spec sigmaA = texEval(M.tex[0], bRec.dg) * M.f[2];
        if (!isZero(sigmaA)) result = result * specExp(-sigmaA * (1 / fabsf(Frame::cosTheta(wiPrime)) + 1 / fabsf(Frame::cosTheta(woPrime))));
        float R21; bRec.wo = coatRefractOut(M, woPrime, R21);
        if (R21 == 1.0f) return Spec(0.0f);
        if (sampleSpecular) { pdf *= 1.0f - probSpecular; result = sdiv(result, 1.0f - probSpecular); }
        result = result * ((1 - R12) * (1 - R21));
        if (bsdfMeasure(bRec.sampledType) == ESolidAngle) {
            result = result * (Frame::cosTheta(bRec.wi) / Frame::cosTheta(wiPrime));
            pdf *= M.f[1] * M.f[1] * Frame::cosTheta(bRec.wo) / Frame::cosTheta(woPrime);
        }
        return result;
    }
    case CTL_BSDF_ROUGHCOATING: {   // BSDF_Complex.cu:159-224
        const ctl_material& nested = nestedMat(M, bRec, 0);
        bool hasNested = (bRec.typeMask & nested.combined_type & EAll) != 0, hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0;
        bool choseSpecular = hasSpecular; V2 sample = _sample;
        Microfacet distr = roughcoatDistr(M, bRec);
        if (hasSpecular && hasNested) {
            float probSpecular = roughcoatProbSpecular(M, bRec, distr);
            if (sample.y < probSpecular) sample.y /= probSpecular;
            else { sample.y = (sample.y - probSpecular) / (1 - probSpecular); choseSpecular = false; }
        }
        if (choseSpecular) {
            float unused; V3 m = distr.sample(bRec.wi, sample, unused);
            bRec.wo = reflectAbout(bRec.wi, m); bRec.sampledType = CTL_EGlossyReflection; bRec.eta = 1.0f;
            if (Frame::cosTheta(bRec.wo) * Frame::cosTheta(bRec.wi) <= 0) return Spec(0.0f);
        } else {
            V3 wiBackup = bRec.wi; bRec.wi = roughcoatRefractTo(M, true, bRec.wi);
            Spec result = bsdfSample(nested, bRec, pdf, sample);
            bRec.wi = wiBackup;
            if (isZero(result)) return Spec(0.0f);
            bRec.wo = roughcoatRefractTo(M, false, bRec.wo);
            if (dot(bRec.wo, bRec.wo) == 0.0f) return Spec(0.0f);
        }
        int measure = bsdfMeasure(bRec.sampledType);
        pdf = bsdfComplexPdf(M, bRec, measure);
        if (pdf == 0) return Spec(0.0f);
        return sdiv(bsdfComplexF(M, bRec, measure), pdf);
    }
    case CTL_BSDF_BLEND: {   // BSDF_Complex.cu:344-372
        float weights[2];
        weights[1] = clampf(avg3(texEval(M.tex[0], bRec.dg)), 0.0f, 1.0f); weights[0] = 1.0f - weights[1];
        V2 sample = _sample; unsigned entry;
        if (sample.x < weights[0]) { entry = 0; sample.x /= weights[0]; } else { entry = 1; sample.x = (sample.x - weights[0]) / weights[1]; }
        Spec result = bsdfSample(nestedMat(M, bRec, (int)entry), bRec, pdf, sample);
        if (isZero(result)) return result;
        result = result * (weights[entry] * pdf);
        pdf *= weights[entry];
        int measure = bsdfMeasure(bRec.sampledType);
        for (unsigned i = 0; i < 2; ++i) {
            if (entry == i) continue;
            pdf += bsdfPdf(nestedMat(M, bRec, (int)i), bRec, measure) * weights[i];
            result = result + bsdfF(nestedMat(M, bRec, (int)i), bRec, measure) * weights[i];
        }
        return sdiv(result, pdf);
    }
    default: throw std::runtime_error("oracle: bsdf type not restated");
    }
}

inline Spec bsdfComplexF(const ctl_material& M, const BRec& bRec, int measure) {
    switch (M.bsdf_type) {
    case CTL_BSDF_COATING: {   // BSDF_Complex.cu:84-122
        const ctl_material& nested = nestedMat(M, bRec, 0);
        bool sampleSpecular = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleNested = (bRec.typeMask & nested.combined_type & EAll) != 0;
        if (measure == EDiscrete && sampleSpecular && fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) < DeltaEpsilon)
            return texEval(M.tex[1], bRec.dg) * fresnelDielectricExt(fabsf(Frame::cosTheta(bRec.wi)), M.f[0]);
        if (sampleNested) {
            float R12, R21; BRec bRecInt = bRec;
            bRecInt.wi = coatRefractIn(M, bRec.wi, R12); bRecInt.wo = coatRefractIn(M, bRec.wo, R21);
            if (R12 == 1 || R21 == 1) return Spec(0.0f);
            Spec result = bsdfF(nested, bRecInt, measure) * (1 - R12) * (1 - R21);
            Spec sigmaA = texEval(M.tex[0], bRec.dg) * M.f[2];
            if (!isZero(sigmaA)) result = result * specExp(-sigmaA * (1 / fabsf(Frame::cosTheta(bRecInt.wi)) + 1 / fabsf(Frame::cosTheta(bRecInt.wo))));
            if (measure == ESolidAngle) result = result * (M.f[1] * M.f[1] * Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) / (Frame::cosTheta(bRecInt.wi) * Frame::cosTheta(bRecInt.wo)));
            return result;
        }
        return Spec(0.0f);
    }
    case CTL_BSDF_ROUGHCOATING: {   // BSDF_Complex.cu:226-284
        const ctl_material& nested = nestedMat(M, bRec, 0);
        bool hasNested = (bRec.typeMask & nested.combined_type & EAll) != 0, hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0 && measure == ESolidAngle;
        Microfacet distr = roughcoatDistr(M, bRec);
        Spec result(0.0f);
        if (hasSpecular && Frame::cosTheta(bRec.wo) * Frame::cosTheta(bRec.wi) > 0) {
            const V3 H = normalize(bRec.wo + bRec.wi) * signum(Frame::cosTheta(bRec.wo));
            const float D = distr.eval(H);
            const float F = fresnelDielectricExt(absdot(bRec.wi, H), M.f[0]);
            const float G = distr.G(bRec.wi, bRec.wo, H);
            float value = F * D * G / (4.0f * fabsf(Frame::cosTheta(bRec.wi)));
            result = result + texEval(M.tex[1], bRec.dg) * value;
        }
        if (hasNested) {
            BRec bRecInt = bRec;
            bRecInt.wi = roughcoatRefractTo(M, true, bRec.wi); bRecInt.wo = roughcoatRefractTo(M, true, bRec.wo);
            Spec nestedResult = bsdfF(nested, bRecInt, measure) * roughTransmittance(bRec.dg, M.u[0], Frame::cosTheta(bRec.wi), distr.alphaU, M.f[0])
                                * roughTransmittance(bRec.dg, M.u[0], Frame::cosTheta(bRec.wo), distr.alphaU, M.f[0]);
            Spec sigmaA = texEval(M.tex[0], bRec.dg) * M.f[2];
            if (!isZero(sigmaA)) nestedResult = nestedResult * specExp(-sigmaA * (1 / fabsf(Frame::cosTheta(bRecInt.wi)) + 1 / fabsf(Frame::cosTheta(bRecInt.wo))));
            if (measure == ESolidAngle) nestedResult = nestedResult * (M.f[1] * M.f[1] * Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) / (Frame::cosTheta(bRecInt.wi) * Frame::cosTheta(bRecInt.wo)));
            result = result + nestedResult;
        }
        return result;
    }
    case CTL_BSDF_BLEND: {   // BSDF_Complex.cu:374-378
        float weight = clampf(avg3(texEval(M.tex[0], bRec.dg)), 0.0f, 1.0f);
        return bsdfF(nestedMat(M, bRec, 0), bRec, measure) * (1 - weight) + bsdfF(nestedMat(M, bRec, 1), bRec, measure) * weight;
    }
    default: throw std::runtime_error("oracle: bsdf type not restated");
    }
}

inline float bsdfComplexPdf(const ctl_material& M, const BRec& bRec, int measure) {
    switch (M.bsdf_type) {
    case CTL_BSDF_COATING: {   // BSDF_Complex.cu:124-157
        const ctl_material& nested = nestedMat(M, bRec, 0);
        bool sampleSpecular = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleNested = (bRec.typeMask & nested.combined_type & EAll) != 0;
        float R12; V3 wiPrime = coatRefractIn(M, bRec.wi, R12);
        float probSpecular = coatProbSpecular(M, R12);
        if (measure == EDiscrete && sampleSpecular && fabsf(dot(Frame::reflect(bRec.wi), bRec.wo) - 1) < DeltaEpsilon) return sampleNested ? probSpecular : 1.0f;
        if (sampleNested) {
            float R21; BRec bRecInt = bRec;
            bRecInt.wi = wiPrime; bRecInt.wo = coatRefractIn(M, bRec.wo, R21);
            if (R12 == 1 || R21 == 1) return 0.0f;
            float pdf = bsdfPdf(nested, bRecInt, measure);
            if (measure == ESolidAngle) pdf *= M.f[1] * M.f[1] * Frame::cosTheta(bRec.wo) / Frame::cosTheta(bRecInt.wo);
            return sampleSpecular ? (pdf * (1 - probSpecular)) : pdf;
        }
        return 0.0f;
    }
    case CTL_BSDF_ROUGHCOATING: {   // BSDF_Complex.cu:286-342
        const ctl_material& nested = nestedMat(M, bRec, 0);
        bool hasNested = (bRec.typeMask & nested.combined_type & EAll) != 0, hasSpecular = (bRec.typeMask & CTL_EGlossyReflection) != 0 && measure == ESolidAngle;
        const V3 H = normalize(bRec.wo + bRec.wi) * signum(Frame::cosTheta(bRec.wo));
        Microfacet distr = roughcoatDistr(M, bRec);
        float probNested, probSpecular;
        if (hasSpecular && hasNested) { probSpecular = roughcoatProbSpecular(M, bRec, distr); probNested = 1 - probSpecular; }
        else probNested = probSpecular = 1.0f;
        float result = 0.0f;
        if (hasSpecular && Frame::cosTheta(bRec.wo) * Frame::cosTheta(bRec.wi) > 0) {
            const float dwh_dwo = 1.0f / (4.0f * absdot(bRec.wo, H));
            const float prob = distr.pdf(bRec.wi, H);
            result = prob * dwh_dwo * probSpecular;
        }
        if (hasNested) {
            BRec bRecInt = bRec;
            bRecInt.wi = roughcoatRefractTo(M, true, bRec.wi); bRecInt.wo = roughcoatRefractTo(M, true, bRec.wo);
            float prob = bsdfPdf(nested, bRecInt, measure);
            if (measure == ESolidAngle) prob *= M.f[1] * M.f[1] * Frame::cosTheta(bRec.wo) / Frame::cosTheta(bRecInt.wo);
            result += prob * probNested;
        }
        return result;
    }
    case CTL_BSDF_BLEND: {   // BSDF_Complex.cu:380-384
        float weight = clampf(avg3(texEval(M.tex[0], bRec.dg)), 0.0f, 1.0f);
        return bsdfPdf(nestedMat(M, bRec, 0), bRec, measure) * (1 - weight) + bsdfPdf(nestedMat(M, bRec, 1), bRec, measure) * weight;
    }
    default: throw std::runtime_error("oracle: bsdf type not restated");
    }
}

} // namespace orc
