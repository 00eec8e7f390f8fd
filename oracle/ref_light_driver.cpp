// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_light_driver.cpp — extern "C" driver around the reference's own emitters (SceneTypes/Light.cu): PointLight, SpotLight and DistantLight built by the
// reference's constructors from their primary parameters, and their sampleDirect; DiffuseLight::pdfDirect / eval with a constant radiance.  `make ref` compiles
// Light.cu through a build-time copy under oracle/_ref/gen/ (git-ignored) without line 6 (`#include <Kernel/TraceHelper.h>`, which pulls in curand_kernel.h) and
// without lines 420-479 (InfiniteLight::internalSampleDirection / internalPdfDirection: they read the scene's global g_SceneData declared in that header).
// Area-light and environment-map SAMPLING go through that global: ref_scene_light_driver.cpp (round 5).  This file contains no reference source.
#include <SceneTypes/Light.h>
#include <cstdint>
#include <cstring>

using namespace CudaTracerLib;

template <class L> static void sample_direct(const L& light, int n, const float* q, float* out) {
    for (int i = 0; i < n; i++) {
        const float* a = q + 8 * i; float* o = out + 14 * i;
        DirectSamplingRecord d(Vec3f(a[0], a[1], a[2]), NormalizedT<Vec3f>(a[3], a[4], a[5]));
        Spectrum v = light.sampleDirect(d, Vec2f(a[6], a[7]));
        float r, g, b; v.toLinearRGB(r, g, b);
        o[0] = r; o[1] = g; o[2] = b; o[3] = d.pdf; o[4] = d.d.x; o[5] = d.d.y; o[6] = d.d.z; o[7] = d.dist;
        o[8] = d.p.x; o[9] = d.p.y; o[10] = d.p.z; o[11] = d.n.x; o[12] = d.n.y; o[13] = d.n.z;
    }
}

extern "C" {

// type 1 point: p = {position(3), intensity(3)};  4 spot: {position(3), target(3), intensity(3), cutoff angle [deg], beam width [deg]};
// 3 distant: {direction(3), normal irradiance(3), scene radius}.   q: 8 floats per query = {ref(3), refN(3), sample(2)};  out: 14 floats per query.
int ref_light_sample_direct(int type, const float* p, int n, const float* q, float* out) {
    if (type == 1) { PointLight l(Vec3f(p[0], p[1], p[2]), Spectrum(p[3], p[4], p[5])); sample_direct(l, n, q, out); return 0; }
    if (type == 4) { SpotLight l(Vec3f(p[0], p[1], p[2]), Vec3f(p[3], p[4], p[5]), Spectrum(p[6], p[7], p[8]), p[9], p[10]); sample_direct(l, n, q, out); return 0; }
    if (type == 3) { DistantLight l(Spectrum(p[3], p[4], p[5]), Vec3f(p[0], p[1], p[2]).normalized(), p[6]); sample_direct(l, n, q, out); return 0; }
    return -1;
}

}  // extern "C"
