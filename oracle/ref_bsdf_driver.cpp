// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_bsdf_driver.cpp — extern "C" driver around the reference's own BSDFs: SceneTypes/BSDF_Simple.cu (diffuse, roughdiffuse, dielectric, thindielectric,
// roughdielectric, conductor, roughconductor, plastic, phong, ward) and SceneTypes/BSDF_Complex.cu (coating, blend), compiled by `make ref`:
// BSDF_Complex.cu as it lies; BSDF_Simple.cu through a build-time copy under oracle/_ref/gen/ (git-ignored) that drops line 2, the unused
// `#include <Base/CudaRandom.h>` (curand_kernel.h does not exist in this image; nothing in the file uses it).  This file contains no reference source.
// roughplastic and roughcoating (round 5): their f / sample evaluate RoughTransmittanceManager -> Math/Spline.cu; both now build (oracle/Makefile: Spline.cu behind one `using` line,
// RoughTransmittance.cu:55-123,140-158 extracted) and the manager's three tables are installed through ref_rough_manager_set (ref_rough_driver.cpp) before a query.
//
// A query builds the reference's BSDF object from the product's flat ctl_material (include/ctl_amd.h) the way INTEGRATION.md's converter maps them back,
// lets the reference's constructor / Update() derive what it derives (fdrInt, invEta2, sampling weights, ...), and calls sample / f / pdf in a frame whose
// shading normal is +z.
#include <SceneTypes/BSDF.h>
#include "../include/ctl_amd.h"
#include <cstdint>
#include <cstring>
#include <stdexcept>

using namespace CudaTracerLib;

static Spectrum spec3(const float* v) { return Spectrum(v[0], v[1], v[2]); }
static Texture tex_of(const ctl_texture& t) {
    Texture r;
    if (t.type == CTL_TEX_CHECKER) { CheckerboardTexture c(spec3(t.value), spec3(t.value1), TextureMapping2D(t.uv_scale[0], t.uv_scale[1], t.uv_offset[0], t.uv_offset[1])); r.SetData(c); }
    else if (t.type == CTL_TEX_CONSTANT) { ConstantTexture c(spec3(t.value)); r.SetData(c); }
    else throw std::runtime_error("ref_bsdf_driver: only constant and checkerboard textures");
    return r;
}
static BSDFFirst simple_of(const ctl_material& M) {
    BSDFFirst b;
    switch (M.bsdf_type) {
    case CTL_BSDF_DIFFUSE: { diffuse d(tex_of(M.tex[0])); d.m_combinedType = M.combined_type; b.SetData(d); break; }   // (the loader's `transmission` flag lives in m_combinedType)
    case CTL_BSDF_ROUGHDIFFUSE: { roughdiffuse d(tex_of(M.tex[0]), tex_of(M.tex[1])); d.m_useFastApprox = M.u[0] != 0; b.SetData(d); break; }
    case CTL_BSDF_DIELECTRIC: { dielectric d(M.f[0], tex_of(M.tex[1]), tex_of(M.tex[0])); d.eta_f.SetData(DispersionCauchy(M.f[0], M.f[1])); b.SetData(d); break; }
    case CTL_BSDF_THINDIELECTRIC: { thindielectric d(M.f[0], tex_of(M.tex[1]), tex_of(M.tex[0])); b.SetData(d); break; }
    case CTL_BSDF_ROUGHDIELECTRIC: { roughdielectric d((MicrofacetDistribution::EType)M.u[0], M.f[0], tex_of(M.tex[2]), tex_of(M.tex[3]), tex_of(M.tex[1]), tex_of(M.tex[0])); d.m_sampleVisible = M.u[1] != 0; b.SetData(d); break; }
    case CTL_BSDF_CONDUCTOR: { conductor d(spec3(M.f), spec3(M.f + 3), tex_of(M.tex[0])); b.SetData(d); break; }
    case CTL_BSDF_ROUGHCONDUCTOR: { roughconductor d((MicrofacetDistribution::EType)M.u[0], spec3(M.f), spec3(M.f + 3), tex_of(M.tex[1]), tex_of(M.tex[2]), tex_of(M.tex[0])); d.m_sampleVisible = M.u[1] != 0; b.SetData(d); break; }
    case CTL_BSDF_PLASTIC: { plastic d(M.f[2], tex_of(M.tex[0]), tex_of(M.tex[1]), M.u[0] != 0); b.SetData(d); break; }
    case CTL_BSDF_PHONG: { phong d(tex_of(M.tex[0]), tex_of(M.tex[1]), tex_of(M.tex[2])); b.SetData(d); break; }
    case CTL_BSDF_ROUGHPLASTIC: { Texture a = tex_of(M.tex[2]), d0 = tex_of(M.tex[0]), s0 = tex_of(M.tex[1]); roughplastic d((MicrofacetDistribution::EType)M.u[2], M.f[0], a, d0, s0, M.u[0] != 0); b.SetData(d); break; }
    case CTL_BSDF_WARD: { ward d((ward::EModelVariant)M.u[0], tex_of(M.tex[0]), tex_of(M.tex[1]), tex_of(M.tex[2]), tex_of(M.tex[3])); b.SetData(d); break; }
    default: throw std::runtime_error("ref_bsdf_driver: model not driven");
    }
    b.As()->m_enableTwoSided = M.two_sided != 0;
    return b;
}
static void fill_rec(BSDFSamplingRecord& r, const float* q, unsigned typeMask) {
    std::memset(&r.dg, 0, sizeof(r.dg));
    r.dg.P = Vec3f(0.0f); r.dg.sys = Frame(NormalizedT<Vec3f>(1.0f, 0.0f, 0.0f), NormalizedT<Vec3f>(0.0f, 1.0f, 0.0f), NormalizedT<Vec3f>(0.0f, 0.0f, 1.0f)); r.dg.n = NormalizedT<Vec3f>(0.0f, 0.0f, 1.0f);
    r.dg.uv[0] = Vec2f(q[6], q[7]); r.dg.hasUVPartials = 0;
    r.wi = NormalizedT<Vec3f>(q[0], q[1], q[2]); r.wo = NormalizedT<Vec3f>(0.0f, 0.0f, 0.0f);
    r.eta = 1.0f; r.mode = ERadiance; r.typeMask = typeMask; r.sampledType = 0;
}
template <class B> static void run(const B& bsdf, int mode, unsigned typeMask, int n, const float* q, float* out) {
    for (int i = 0; i < n; i++) {
        const float* a = q + 8 * i; float* o = out + 9 * i;
        BSDFSamplingRecord r; fill_rec(r, a, typeMask);
        if (mode == 0) {
            float pdf = 0.0f; Spectrum w = bsdf.sample(r, pdf, Vec2f(a[3], a[4]));
            float cr, cg, cb; w.toLinearRGB(cr, cg, cb);
            o[0] = cr; o[1] = cg; o[2] = cb; o[3] = pdf; o[4] = r.wo.x; o[5] = r.wo.y; o[6] = r.wo.z; o[7] = (float)r.sampledType; o[8] = r.eta;
        } else {
            r.wo = NormalizedT<Vec3f>(a[3], a[4], a[5]);
            const EMeasure m = mode == 2 ? EDiscrete : ESolidAngle;
            Spectrum f = bsdf.f(r, m); float cr, cg, cb; f.toLinearRGB(cr, cg, cb);
            o[0] = cr; o[1] = cg; o[2] = cb; o[3] = bsdf.pdf(r, m); o[4] = o[5] = o[6] = o[7] = o[8] = 0.0f;
        }
    }
}

extern "C" {

// mats[idx]: the material (its u2 / u3 are indices into mats for coating / blend).  mode 0: sample, 1: f + pdf (solid angle), 2: f + pdf (discrete measure).
// q: 8 floats per query = {wi.xyz, (sample.x, sample.y, -) | wo.xyz, u, v}; out: 9 floats per query = {rgb, pdf, wo.xyz, sampledType, eta} (mode 0) / {rgb, pdf, 0...}.
// Returns 0, or -1 when the model (or a texture kind) is not driven.
int ref_bsdf_query(const ctl_material* mats, uint32_t idx, int mode, uint32_t typeMask, int n, const float* q, float* out) {
    try {
        const ctl_material& M = mats[idx];
        if (M.bsdf_type == CTL_BSDF_COATING) {
            coating c(simple_of(mats[M.u[2]]), M.f[0], M.f[2], tex_of(M.tex[0]), tex_of(M.tex[1]));
            run(c, mode, typeMask, n, q, out);
        } else if (M.bsdf_type == CTL_BSDF_ROUGHCOATING) {
            roughcoating c((MicrofacetDistribution::EType)M.u[0], simple_of(mats[M.u[2]]), M.f[0], M.f[2], tex_of(M.tex[0]), tex_of(M.tex[2]), tex_of(M.tex[1]));
            run(c, mode, typeMask, n, q, out);
        } else if (M.bsdf_type == CTL_BSDF_BLEND) {
            blend b(simple_of(mats[M.u[2]]), simple_of(mats[M.u[3]]), tex_of(M.tex[0]));
            run(b, mode, typeMask, n, q, out);
        } else {
            const BSDFFirst b = simple_of(M);
            run(b, mode, typeMask, n, q, out);
        }
        return 0;
    } catch (const std::exception&) { return -1; }
}
// what the reference's constructors derive: {m_combinedType, fdrInt, fdrExt, invEta2, specularSamplingWeight} of a plastic / {.., weight} of phong, ward, coating
int ref_bsdf_derived(const ctl_material* mats, uint32_t idx, float* out5) {
    try {
        const ctl_material& M = mats[idx];
        std::memset(out5, 0, 20);
        if (M.bsdf_type == CTL_BSDF_COATING) { coating c(simple_of(mats[M.u[2]]), M.f[0], M.f[2], tex_of(M.tex[0]), tex_of(M.tex[1])); out5[0] = (float)c.m_combinedType; out5[4] = c.m_specularSamplingWeight; out5[3] = c.m_invEta; return 0; }
        if (M.bsdf_type == CTL_BSDF_ROUGHCOATING) { roughcoating c((MicrofacetDistribution::EType)M.u[0], simple_of(mats[M.u[2]]), M.f[0], M.f[2], tex_of(M.tex[0]), tex_of(M.tex[2]), tex_of(M.tex[1])); out5[0] = (float)c.m_combinedType; out5[4] = c.m_specularSamplingWeight; out5[3] = c.m_invEta; out5[1] = c.m_sampleVisible ? 1.0f : 0.0f; return 0; }
        if (M.bsdf_type == CTL_BSDF_BLEND) { blend b(simple_of(mats[M.u[2]]), simple_of(mats[M.u[3]]), tex_of(M.tex[0])); out5[0] = (float)b.m_combinedType; return 0; }
        const BSDFFirst b = simple_of(M);
        out5[0] = (float)b.getType();
        if (M.bsdf_type == CTL_BSDF_PLASTIC) { const plastic* p = b.As<plastic>(); out5[1] = p->m_fdrInt; out5[2] = p->m_fdrExt; out5[3] = p->m_invEta2; out5[4] = p->m_specularSamplingWeight; }
        if (M.bsdf_type == CTL_BSDF_ROUGHPLASTIC) { const roughplastic* p = b.As<roughplastic>(); out5[1] = p->m_sampleVisible ? 1.0f : 0.0f; out5[3] = p->m_invEta2; out5[4] = p->m_specularSamplingWeight; }
        if (M.bsdf_type == CTL_BSDF_PHONG) out5[4] = b.As<phong>()->m_specularSamplingWeight;
        if (M.bsdf_type == CTL_BSDF_WARD) out5[4] = b.As<ward>()->m_specularSamplingWeight;
        return 0;
    } catch (const std::exception&) { return -1; }
}

}  // extern "C"
