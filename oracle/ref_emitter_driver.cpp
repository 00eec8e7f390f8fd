// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_emitter_driver.cpp — extern "C" driver around the reference's own emitter SELECTION: KernelDynamicScene::getLight / sampleEmitter / pdfEmitter /
// sampleEmitterDirect (Engine/KernelDynamicScene.cu:8-11, 25-46, 98-117).  Those are plain member functions over `this` — no g_SceneData, no curand — and `make ref`
// compiles exactly these line ranges through a build-time extract under oracle/_ref/gen/ (git-ignored) behind the reference's own headers Engine/KernelDynamicScene.h,
// SceneTypes/Light.h and Base/STL.h; the rest of the file (TraceHelper.h -> curand, the volume, the sensor sampling) is left out.
// What is driven: the light choice by the CDF, the re-scaled sample, emPdf, the pdfEmitter-indexes-the-CDF-by-BUFFER-slot quirk, and
// sampleEmitterDirect over point / spot / distant lights (the emitters whose sampleDirect builds here, ref_light_driver.cpp).  This file contains no reference source.
#include <Engine/KernelDynamicScene.h>
#include <SceneTypes/Light.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace CudaTracerLib;

namespace {
// a KernelDynamicScene whose only live members are the ones the driven functions read: the light buffer, the index list and the CDF.
// Raw zeroed storage: the struct's other members (sensor, volume, BVH views) have no use here and are never touched.
struct scene_box {
    KernelDynamicScene* ks; Light* buf; unsigned n_buf;
    scene_box(const float* cdf, const uint32_t* indices, int n_lights, int n_buf_) : n_buf((unsigned)n_buf_) {
        ks = (KernelDynamicScene*)std::calloc(1, sizeof(KernelDynamicScene));
        buf = (Light*)std::calloc((size_t)n_buf_, sizeof(Light));
        ks->m_sLightBuf.Data = buf; ks->m_sLightBuf.UsedCount = ks->m_sLightBuf.Length = (unsigned)n_buf_;
        ks->m_numLights = (unsigned)n_lights; ks->m_uEnvMapIndex = UINT_MAX;
        for (int i = 0; i < n_lights && i < MAX_NUM_LIGHTS; i++) { ks->m_pLightIndices[i] = indices[i]; ks->m_pLightCDF[i] = cdf[i]; }
    }
    ~scene_box() { std::free(buf); std::free(ks); }
};
}  // namespace

extern "C" {

// sampleEmitter for nq samples (2 floats each): slot_out = index of the chosen light in the BUFFER, pdf_out = emPdf, resampled_out = sample.x afterwards;
// pdf_emitter_out[n_buf] = pdfEmitter(&buffer[i]) for every buffer slot i < MAX_NUM_LIGHTS
int ref_emitter_select(const float* cdf, const uint32_t* indices, int n_lights, int n_buf, int nq, const float* samples, int32_t* slot_out, float* pdf_out, float* resampled_out, float* pdf_emitter_out) {
    if (n_lights < 0 || n_lights > MAX_NUM_LIGHTS || n_buf < 1) return -1;
    scene_box B(cdf, indices, n_lights, n_buf);
    for (int i = 0; i < nq; i++) {
        Vec2f s(samples[2 * i], samples[2 * i + 1]); float emPdf = 0.0f;
        const Light* L = B.ks->sampleEmitter(emPdf, s);
        slot_out[i] = L ? (int32_t)(L - B.buf) : -1; pdf_out[i] = emPdf; resampled_out[i] = s.x;
    }
    for (int i = 0; i < n_buf && i < MAX_NUM_LIGHTS; i++) pdf_emitter_out[i] = B.ks->pdfEmitter(B.buf + i);
    return 0;
}

// sampleEmitterDirect over a buffer of point (type 1) / distant (3) / spot (4) lights, 12 floats of parameters each (as ref_light_sample_direct takes them; unused slots type 0).
// q: 8 floats per query {ref(3), refN(3), sample(2)}; out: 15 floats per query {value rgb, pdf, d(3), dist, p(3), n(3), buffer slot of dRec.object or -1}
int ref_sample_emitter_direct(const float* cdf, const uint32_t* indices, int n_lights, int n_buf, const int32_t* types, const float* params, int nq, const float* q, float* out) {
    if (n_lights < 0 || n_lights > MAX_NUM_LIGHTS || n_buf < 1) return -1;
    scene_box B(cdf, indices, n_lights, n_buf);
    for (int i = 0; i < n_buf; i++) {
        const float* p = params + 12 * i;
        if (types[i] == 1) B.buf[i].SetData(PointLight(Vec3f(p[0], p[1], p[2]), Spectrum(p[3], p[4], p[5])));
        else if (types[i] == 4) B.buf[i].SetData(SpotLight(Vec3f(p[0], p[1], p[2]), Vec3f(p[3], p[4], p[5]), Spectrum(p[6], p[7], p[8]), p[9], p[10]));
        else if (types[i] == 3) B.buf[i].SetData(DistantLight(Spectrum(p[3], p[4], p[5]), Vec3f(p[0], p[1], p[2]).normalized(), p[6]));
    }
    for (int i = 0; i < nq; i++) {
        const float* a = q + 8 * i; float* o = out + 15 * i;
        DirectSamplingRecord d(Vec3f(a[0], a[1], a[2]), NormalizedT<Vec3f>(a[3], a[4], a[5]));
        Spectrum v = B.ks->sampleEmitterDirect(d, Vec2f(a[6], a[7]));
        float r, g, b; v.toLinearRGB(r, g, b);
        o[0] = r; o[1] = g; o[2] = b; o[3] = d.pdf; o[4] = d.d.x; o[5] = d.d.y; o[6] = d.d.z; o[7] = d.dist;
        o[8] = d.p.x; o[9] = d.p.y; o[10] = d.p.z; o[11] = d.n.x; o[12] = d.n.y; o[13] = d.n.z;
        o[14] = d.object ? (float)((const Light*)d.object - B.buf) : -1.0f;
    }
    return 0;
}

}  // extern "C"
