// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_driver.cpp — thin extern "C" driver around the REFERENCE'S OWN host-compilable code, compiled from
// /root/reference where it lies (oracle/Makefile target `ref`; output only into oracle/_ref/, git-ignored).
// It contains no copy of reference source: it includes the reference headers and calls their functions.
// Used by tests/ to pin the restatement in ocore.h/omath.h and to generate tests/golden/ fixtures.
//
// Header-only reference code driven from here as well: Warp.h, Frame.h, half.h, Compression.h, float4x4.h, MonteCarlo.h, Filter.h, and the traversal
// template of BVHTraversal.h with the slab arithmetic of MathFunc.h (ref_trace_two_level).
//
// Further reference code is driven by ref_mipmap_driver.cpp (KernelMIPMap), ref_bsdf_driver.cpp (BSDF_Simple.cu, BSDF_Complex.cu) and ref_light_driver.cpp
// (Light.cu) — files whose host-compilable part `make ref` extracts at build time.
// What could NOT be built from the reference here (needs curand_kernel.h from the CUDA toolkit, nvcc's ::min / ::max, the global g_SceneData declared behind
// curand, or un-vendored boost / pugixml / FreeImage): Math/Spline.cu (rough transmittance), Engine/ShapeSet.cu (area-light sampling), the two environment-map
// sampling functions of Light.cu, KernelDynamicScene.cu, TraceHelper.cu, TraceAlgorithms.cu, PathTracer.cu, Sampler / CudaRandom, DynamicScene, the Mitsuba
// loader.  See DESIGN.md "Oracle".
#include <Engine/TriIntersectorData.h>
#include <Engine/TriangleData.h>
#include <Engine/DifferentialGeometry.h>
#include <Engine/MicrofacetDistribution.h>
#include <Engine/MeshLoader/BVHBuilderHelper.h>
#include <Math/Compression.h>
#include <Math/half.h>
#include <Math/Warp.h>
#include <Math/Frame.h>
#include <Math/FresnelHelper.h>
#include <Math/MonteCarlo.h>
#include <Math/Ray.h>
#include <SceneTypes/Sensor.h>
#include <SceneTypes/Filter.h>
#include <SceneTypes/Texture.h>
#include <Engine/MIPMap_device.h>
#include <Engine/SpatialStructures/BVH/BVHTraversal.h>
#include <Math/float4x4.h>
#include <Math/AlgebraHelper.h>
#include <cstdint>
#include <cstring>

using namespace CudaTracerLib;

template <class T> static void sensor_rays(T& s, const float* to_world, float nearD, float farD, float px, float py, float ax, float ay, float* out18, float* out6b) {
    s.SetNearFarDepth(nearD, farD);
    NormalizedT<OrthogonalAffineMap> m; std::memcpy(m.data, to_world, 64);
    s.SetToWorld(m);   // -> Update()
    NormalizedT<Ray> r, r2, rx, ry;
    s.sampleRay(r, Vec2f(px, py), Vec2f(ax, ay));
    s.sampleRayDifferential(r2, rx, ry, Vec2f(px, py), Vec2f(ax, ay));
    const Vec3f v[6] = { r.ori(), r.dir(), rx.ori(), rx.dir(), ry.ori(), ry.dir() };
    for (int k = 0; k < 6; k++) { out18[3 * k] = v[k].x; out18[3 * k + 1] = v[k].y; out18[3 * k + 2] = v[k].z; }
    out6b[0] = r2.ori().x; out6b[1] = r2.ori().y; out6b[2] = r2.ori().z; out6b[3] = r2.dir().x; out6b[4] = r2.dir().y; out6b[5] = r2.dir().z;
}

extern "C" {

void ref_woop_set_data(const float* v0, const float* v1, const float* v2, float* out12) {
    TriIntersectorData t; t.setData(Vec3f(v0[0], v0[1], v0[2]), Vec3f(v1[0], v1[1], v1[2]), Vec3f(v2[0], v2[1], v2[2]));
    std::memcpy(out12, &t, 48);
}
void ref_woop_get_data(const float* w12, float* v0, float* v1, float* v2) {
    TriIntersectorData t; std::memcpy(&t, w12, 48);
    Vec3f a, b, c; t.getData(a, b, c);
    v0[0] = a.x; v0[1] = a.y; v0[2] = a.z; v1[0] = b.x; v1[1] = b.y; v1[2] = b.z; v2[0] = c.x; v2[1] = c.y; v2[2] = c.z;
}
// TriIntersectorData::Intersect uses the fixed tmin 1e-4 (TriIntersectorData.cu:40)
int ref_woop_intersect(const float* w12, const float* o, const float* d, float tmax, float* tuv) {
    TriIntersectorData t; std::memcpy(&t, w12, 48);
    float dist = tmax; Vec2f bary(0.0f);   // left untouched on a miss
    bool h = t.Intersect(Ray(Vec3f(o[0], o[1], o[2]), Vec3f(d[0], d[1], d[2])), &dist, &bary);
    tuv[0] = dist; tuv[1] = bary.x; tuv[2] = bary.y;
    return h ? 1 : 0;
}
uint16_t ref_float_to_half(float f) { return (uint16_t)half(f).bits(); }
float ref_half_to_float(uint16_t h) { return half((unsigned short)h).ToFloat(); }   // host branch (half.h:76-83)
uint16_t ref_normal_encode(const float* n) { return NormalizedFloat3ToUchar2(NormalizedT<Vec3f>(Vec3f(n[0], n[1], n[2]))); }
void ref_normal_decode(uint16_t v, float* n) { auto r = Uchar2ToNormalizedFloat3(v); n[0] = r.x; n[1] = r.y; n[2] = r.z; }
void ref_matrix_inverse(const float* m, float* out) { float4x4 a; std::memcpy(a.data, m, 64); float4x4 r = a.inverse(); std::memcpy(out, r.data, 64); }

void ref_triangle_data_pack(const float* P, const float* N, const float* T, uint32_t mat_index, uint32_t* out8) {
    Vec3f p[3] = { Vec3f(P[0], P[1], P[2]), Vec3f(P[3], P[4], P[5]), Vec3f(P[6], P[7], P[8]) };
    NormalizedT<Vec3f> n[3] = { NormalizedT<Vec3f>(Vec3f(N[0], N[1], N[2])), NormalizedT<Vec3f>(Vec3f(N[3], N[4], N[5])), NormalizedT<Vec3f>(Vec3f(N[6], N[7], N[8])) };
    Vec2f t[3] = { Vec2f(T[0], T[1]), Vec2f(T[2], T[3]), Vec2f(T[4], T[5]) };
    TriangleData td; std::memset(&td, 0, sizeof(td));
    td = TriangleData(p, (unsigned char)mat_index, t, n);
    std::memcpy(out8, &td, 32);
}
void ref_triangle_fill_dg(const uint32_t* td8, const float* local_to_world, float u, float v, float* out) {
    TriangleData td; std::memcpy(&td, td8, 32);
    float4x4 m; std::memcpy(m.data, local_to_world, 64);
    DifferentialGeometry dg; dg.bary = Vec2f(u, v);
    td.fillDG(m, dg);
    const Vec3f vs[6] = { dg.sys.s, dg.sys.t, dg.sys.n, dg.n, dg.dpdu, dg.dpdv };
    for (int i = 0; i < 6; i++) { out[i * 3] = vs[i].x; out[i * 3 + 1] = vs[i].y; out[i * 3 + 2] = vs[i].z; }
    out[18] = dg.uv[0].x; out[19] = dg.uv[0].y; out[20] = (float)dg.extraData;
}

void ref_square_to_cosine_hemisphere(float x, float y, float* out) { auto r = Warp::squareToCosineHemisphere(Vec2f(x, y)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_square_to_uniform_triangle(float x, float y, float* out) { auto r = Warp::squareToUniformTriangle(Vec2f(x, y)); out[0] = r.x; out[1] = r.y; }
void ref_square_to_uniform_disk_concentric(float x, float y, float* out) { auto r = Warp::squareToUniformDiskConcentric(Vec2f(x, y)); out[0] = r.x; out[1] = r.y; }
float ref_fresnel_dielectric_ext(float cosThetaI, float eta, float* cosThetaT) { return FresnelHelper::fresnelDielectricExt(cosThetaI, *cosThetaT, eta); }
void ref_fresnel_conductor_exact(float cosThetaI, const float* eta, const float* k, float* out) {
    // the scalar overload (FresnelHelper.h:92-117): the Spectrum overload (:119-146) is the same expression per channel, but
    // Spectrum's RGB constructor lives in Math/Spectrum.cu, which only nvcc can compile (cudaMemcpyToSymbol of a struct, :729)
    for (int c = 0; c < 3; c++) out[c] = FresnelHelper::fresnelConductorExact(cosThetaI, eta[c], k[c]);
}
void ref_coordinate_system(const float* a, float* s, float* t) {
    NormalizedT<Vec3f> S, T; coordinateSystem(NormalizedT<Vec3f>(Vec3f(a[0], a[1], a[2])), S, T);
    s[0] = S.x; s[1] = S.y; s[2] = S.z; t[0] = T.x; t[1] = T.y; t[2] = T.z;
}
float ref_power_heuristic(float a, float b) { return MonteCarlo::PowerHeuristic(1, a, 1, b); }
float ref_fresnel_diffuse_reflectance(float eta, int fast) { return FresnelHelper::fresnelDiffuseReflectance(eta, fast != 0); }

void ref_microfacet_eval(int type, float aU, float aV, int sampleVisible, const float* wi, const float* m, float* out) {
    MicrofacetDistribution d((MicrofacetDistribution::EType)type, aU, aV, sampleVisible != 0);
    NormalizedT<Vec3f> WI(Vec3f(wi[0], wi[1], wi[2])), M(Vec3f(m[0], m[1], m[2]));
    out[0] = d.eval(M); out[1] = d.smithG1(WI, M); out[2] = d.pdf(WI, M);
}
void ref_microfacet_sample(int type, float aU, float aV, int sampleVisible, const float* wi, float sx, float sy, float* out) {
    MicrofacetDistribution d((MicrofacetDistribution::EType)type, aU, aV, sampleVisible != 0);
    float pdf; auto m = d.sample(NormalizedT<Vec3f>(Vec3f(wi[0], wi[1], wi[2])), Vec2f(sx, sy), pdf);
    out[0] = m.x; out[1] = m.y; out[2] = m.z; out[3] = pdf;
}

// PerspectiveSensor::sampleRay (SceneTypes/Sensor.cu:116-128). to_world row-major, fov in radians.
void ref_sensor_sample_ray(const float* to_world, float fov_rad, float nearD, float farD, int w, int h, float px, float py, float* o, float* d) {
    PerspectiveSensor s(w, h, 90.0f);
    s.SetNearFarDepth(nearD, farD);
    s.fov = fov_rad;
    NormalizedT<OrthogonalAffineMap> m; std::memcpy(m.data, to_world, 64);
    s.SetToWorld(m);   // -> Update()
    NormalizedT<Ray> r;
    s.sampleRay(r, Vec2f(px, py), Vec2f(0.0f));
    o[0] = r.ori().x; o[1] = r.ori().y; o[2] = r.ori().z; d[0] = r.dir().x; d[1] = r.dir().y; d[2] = r.dir().z;
}

// DifferentialGeometry::computePartials (Engine/DifferentialGeometry.cu:9-90).  out = dudx, dudy, dvdx, dvdy
void ref_compute_partials(const float* P, const float* n, const float* dpdu, const float* dpdv, const float* ro, const float* rd, const float* rxd, const float* ryd, float* out4) {
    DifferentialGeometry dg;
    dg.P = Vec3f(P[0], P[1], P[2]); dg.n = NormalizedT<Vec3f>(Vec3f(n[0], n[1], n[2])); dg.dpdu = Vec3f(dpdu[0], dpdu[1], dpdu[2]); dg.dpdv = Vec3f(dpdv[0], dpdv[1], dpdv[2]);
    const Vec3f o(ro[0], ro[1], ro[2]);
    dg.computePartials(Ray(o, Vec3f(rd[0], rd[1], rd[2])), Ray(o, Vec3f(rxd[0], rxd[1], rxd[2])), Ray(o, Vec3f(ryd[0], ryd[1], ryd[2])));
    out4[0] = dg.dudx; out4[1] = dg.dudy; out4[2] = dg.dvdx; out4[3] = dg.dvdy;
}
// PerspectiveSensor::sampleRayDifferential (SceneTypes/Sensor.cu:130-144)
void ref_sensor_sample_ray_differential(const float* to_world, float fov_rad, float nearD, float farD, int w, int h, float px, float py, float* o, float* d, float* dX, float* dY) {
    PerspectiveSensor s(w, h, 90.0f);
    s.SetNearFarDepth(nearD, farD);
    s.fov = fov_rad;
    NormalizedT<OrthogonalAffineMap> m; std::memcpy(m.data, to_world, 64);
    s.SetToWorld(m);
    NormalizedT<Ray> r, rx, ry;
    s.sampleRayDifferential(r, rx, ry, Vec2f(px, py), Vec2f(0.0f));
    o[0] = r.ori().x; o[1] = r.ori().y; o[2] = r.ori().z; d[0] = r.dir().x; d[1] = r.dir().y; d[2] = r.dir().z;
    dX[0] = rx.dir().x; dX[1] = rx.dir().y; dX[2] = rx.dir().z; dY[0] = ry.dir().x; dY[1] = ry.dir().y; dY[2] = ry.dir().z;
}

// The four projective sensors (SceneTypes/Sensor.cu): type = TYPE_FUNC id (2 perspective, 3 thin lens, 4 orthographic, 5 telecentric).
// out18 = sampleRay (o, d), then the x ray (o, d) and the y ray (o, d) of sampleRayDifferential; out6b = sampleRayDifferential's own ray
void ref_sensor_rays(int type, const float* to_world, float fov_rad, float nearD, float farD, int w, int h, float aperture, float focus, float screen_scale,
                     float px, float py, float ax, float ay, float* out18, float* out6b) {
    if (type == 1) { SphericalSensor s(w, h); NormalizedT<OrthogonalAffineMap> m; std::memcpy(m.data, to_world, 64); s.SetToWorld(m); NormalizedT<Ray> r; s.sampleRay(r, Vec2f(px, py), Vec2f(ax, ay));
        const Vec3f v[2] = { r.ori(), r.dir() }; for (int k = 0; k < 6; k++) { out18[3 * k] = v[k & 1].x; out18[3 * k + 1] = v[k & 1].y; out18[3 * k + 2] = v[k & 1].z; } for (int k = 0; k < 6; k++) out6b[k] = out18[k]; }
    else if (type == 2) { PerspectiveSensor s(w, h, 90.0f); s.fov = fov_rad; sensor_rays(s, to_world, nearD, farD, px, py, ax, ay, out18, out6b); }
    else if (type == 3) { ThinLensSensor s(w, h, 90.0f, aperture, focus); s.fov = fov_rad; sensor_rays(s, to_world, nearD, farD, px, py, ax, ay, out18, out6b); }
    else if (type == 4) { OrthographicSensor s(w, h, screen_scale, screen_scale); sensor_rays(s, to_world, nearD, farD, px, py, ax, ay, out18, out6b); }
    else { TelecentricSensor s(w, h, aperture, focus, screen_scale, screen_scale); sensor_rays(s, to_world, nearD, farD, px, py, ax, ay, out18, out6b); }
}
// computePartials with the differential rays' own origins (orthographic / telecentric sensors)
void ref_compute_partials_origins(const float* P, const float* n, const float* dpdu, const float* dpdv, const float* ro, const float* rd, const float* rox, const float* rxd, const float* roy, const float* ryd, float* out4) {
    DifferentialGeometry dg;
    dg.P = Vec3f(P[0], P[1], P[2]); dg.n = NormalizedT<Vec3f>(Vec3f(n[0], n[1], n[2])); dg.dpdu = Vec3f(dpdu[0], dpdu[1], dpdu[2]); dg.dpdv = Vec3f(dpdv[0], dpdv[1], dpdv[2]);
    dg.computePartials(Ray(Vec3f(ro[0], ro[1], ro[2]), Vec3f(rd[0], rd[1], rd[2])), Ray(Vec3f(rox[0], rox[1], rox[2]), Vec3f(rxd[0], rxd[1], rxd[2])), Ray(Vec3f(roy[0], roy[1], roy[2]), Vec3f(ryd[0], ryd[1], ryd[2])));
    out4[0] = dg.dudx; out4[1] = dg.dudy; out4[2] = dg.dvdx; out4[3] = dg.dvdy;
}

// Texel / frame codecs (Math/Spectrum.h:521-565): SpectrumConverter::Float3ToRGBE / RGBEToFloat3 / Float3ToCOLORREF / COLORREFToFloat3
uint32_t ref_float3_to_rgbe(float r, float g, float b) { RGBE v = SpectrumConverter::Float3ToRGBE(Vec3f(r, g, b)); return (uint32_t)v.x | ((uint32_t)v.y << 8) | ((uint32_t)v.z << 16) | ((uint32_t)v.w << 24); }
void ref_rgbe_to_float3(uint32_t q, float* out) { RGBE v; v.x = q & 255; v.y = (q >> 8) & 255; v.z = (q >> 16) & 255; v.w = q >> 24; Vec3f c = SpectrumConverter::RGBEToFloat3(v); out[0] = c.x; out[1] = c.y; out[2] = c.z; }
uint32_t ref_float3_to_rgbcol(float r, float g, float b) { RGBCOL v = SpectrumConverter::Float3ToCOLORREF(Vec3f(r, g, b)); return (uint32_t)v.x | ((uint32_t)v.y << 8) | ((uint32_t)v.z << 16) | ((uint32_t)v.w << 24); }
void ref_rgbcol_to_float3(uint32_t q, float* out) { RGBCOL v; v.x = q & 255; v.y = (q >> 8) & 255; v.z = (q >> 16) & 255; v.w = q >> 24; Vec3f c = SpectrumConverter::COLORREFToFloat3(v); out[0] = c.x; out[1] = c.y; out[2] = c.z; }

// WrapCoordinates (Engine/MIPMap_device.h:33-55): texture addressing of the four wrap modes; returns 0 when the lookup is black
int ref_wrap_coordinates(float u, float v, float w, float h, int mode, float* loc) {
    Vec2f l(0.0f); const bool ok = WrapCoordinates(Vec2f(u, v), Vec2f(w, h), (ImageWrap)mode, &l); loc[0] = l.x; loc[1] = l.y; return ok ? 1 : 0;
}
// CheckerboardTexture::Evaluate over TextureMapping2D(su, sv, du, dv) (SceneTypes/Texture.h:127-146, :10-41): 0 = val0, 1 = val1
int ref_checkerboard_select(float u, float v, float su, float sv, float du, float dv) {
    CheckerboardTexture t(Spectrum(1.0f), Spectrum(0.0f), TextureMapping2D(su, sv, du, dv));
    Spectrum s = t.Evaluate(Vec2f(u, v)); return s[0] > 0.5f ? 0 : 1;
}

// Reconstruction filters of the image pipeline (SceneTypes/Filter.h:28-171): type = TYPE_FUNC id (1 box, 2 Gaussian, 3 Mitchell, 4 Lanczos-sinc, 5 triangle)
float ref_filter_evaluate(int type, float xw, float yw, float p0, float p1, float x, float y) {
    if (type == 1) return BoxFilter(xw, yw).Evaluate(x, y);
    if (type == 2) return GaussianFilter(xw, yw, p0).Evaluate(x, y);
    if (type == 3) return MitchellFilter(p0, p1, xw, yw).Evaluate(x, y);
    if (type == 4) return LanczosSincFilter(xw, yw, p0).Evaluate(x, y);
    return TriangleFilter(xw, yw).Evaluate(x, y);
}

// ConstructBVH (Engine/MeshLoader/BVHBuilderHelper.cpp:116-127): SBVH with max leaf size 8.
// Two-call protocol: first with NULL outputs to get the counts, then with buffers.
static BVH_Construction_Result g_last;
void ref_construct_bvh(const float* vertices, const uint32_t* indices, uint32_t vCount, uint32_t iCount, uint32_t* n_nodes, uint32_t* n_tris) {
    g_last = BVH_Construction_Result();
    ConstructBVH((const Vec3f*)vertices, indices, vCount, iCount, g_last);
    *n_nodes = (uint32_t)g_last.nodes.size(); *n_tris = (uint32_t)g_last.tris.size();
}
void ref_construct_bvh_fetch(void* nodes, void* tris, void* tris2) {
    std::memcpy(nodes, g_last.nodes.data(), g_last.nodes.size() * sizeof(BVHNodeData));
    std::memcpy(tris, g_last.tris.data(), g_last.tris.size() * sizeof(TriIntersectorData));
    std::memcpy(tris2, g_last.tris2.data(), g_last.tris2.size() * sizeof(TriIntersectorData2));
}


// Two-level single-ray traversal with the REFERENCE'S OWN traversal template at both levels (TracerayTemplate, Engine/SpatialStructures/BVH/BVHTraversal.h:122-232:
// its slab arithmetic through kepler_math::spanBeginKepler / spanEndKepler, its child ordering, its postponed leaf), the reference's float4x4::TransformDirection /
// TransformPoint for the instance (Math/float4x4.h) and the reference's TriIntersectorData::Intersect per leaf entry (fixed tmin 1e-4: the rays must carry that tmin).
// The two callbacks have the shape of __traceRay_internal__ (Kernel/TraceHelper.cu:88-170 — that file itself needs curand through TraceHelper.h and does not build here):
// scene-BVH leaf -> node -> mesh BVH at the mesh's node offset; mesh-BVH leaf -> entries up to the one whose index word has bit 0 set.
// Inputs are plain arrays in the reference's layouts: BVHNodeData (64 B), TriIntersectorData (48 B), index words, ctl_node (6 words, [0] = mesh index),
// ctl_kernel_mesh (5 words: tri_offset, bvh_node_offset, bvh_tri_offset, bvh_index_offset, -), one row-major 4x4 inverse transform per node.
void ref_trace_two_level(const void* scene_nodes, int scene_start_node, const void* mesh_nodes, const void* woop, const uint32_t* woop_index,
                         const uint32_t* nodes6, const uint32_t* meshes5, const float* node_inv, uint32_t n_rays, const float* rays8,
                         float* out_tuv, int32_t* out_tri, int32_t* out_node) {
    const BVHNodeData* SN = (const BVHNodeData*)scene_nodes; const BVHNodeData* MN = (const BVHNodeData*)mesh_nodes;
    const TriIntersectorData* W = (const TriIntersectorData*)woop;
    for (uint32_t i = 0; i < n_rays; i++) {
        const float* r = rays8 + 8 * (size_t)i;
        const Vec3f ori(r[0], r[1], r[2]), dir(r[4], r[5], r[6]);
        float dist = r[7]; Vec2f bary(0.0f); int tri = -1, node = -1;
        TracerayTemplate(Ray(ori, dir), dist, [&](int nodeIdx) {
            const uint32_t* m = meshes5 + 5 * (size_t)nodes6[6 * (size_t)nodeIdx];
            float4x4 modl; std::memcpy(modl.data, node_inv + 16 * (size_t)nodeIdx, 64);
            const Vec3f d = modl.TransformDirection(dir), o = modl.TransformPoint(ori);
            return TracerayTemplate(Ray(o, d), dist, [&](int triIdx) {
                bool found = false;
                for (int triAddr = triIdx;; triAddr++) {
                    const uint32_t index = woop_index[m[3] + triAddr];
                    if (W[m[2] / 3 + triAddr].Intersect(Ray(o, d), &dist, &bary)) { node = nodeIdx; tri = (int)((index >> 1) + m[0]); found = true; }
                    if (index & 1) break;
                }
                return found;
            }, MN, MN, (int)m[1], 0);
        }, SN, SN, 0, scene_start_node);
        out_tuv[3 * (size_t)i] = dist; out_tuv[3 * (size_t)i + 1] = bary.x; out_tuv[3 * (size_t)i + 2] = bary.y;
        out_tri[i] = tri; out_node[i] = node;
    }
}

// second batch of small functions of the path (Math/Warp.h:13-27, :29-36, :68-71, :180-186; Math/AlgebraHelper.h:46-59)
float ref_interval_to_tent(float s) { return Warp::squareToTent(Vec2f(s, s)).x; }   // intervalToTent itself is private: squareToTent applies it per component
void ref_square_to_tent(float x, float y, float* out) { auto r = Warp::squareToTent(Vec2f(x, y)); out[0] = r.x; out[1] = r.y; }
float ref_cosine_hemisphere_pdf(const float* d) { return Warp::squareToCosineHemispherePdf(NormalizedT<Vec3f>(Vec3f(d[0], d[1], d[2]))); }
void ref_square_to_uniform_sphere(float x, float y, float* out) { auto r = Warp::squareToUniformSphere(Vec2f(x, y)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
int ref_barycentric(const float* p, const float* a, const float* b, const float* c, float* uv) {
    float u = 0, v = 0;
    const bool in = AlgebraHelper::Barycentric(Vec3f(p[0], p[1], p[2]), Vec3f(a[0], a[1], a[2]), Vec3f(b[0], b[1], b[2]), Vec3f(c[0], c[1], c[2]), u, v);
    uv[0] = u; uv[1] = v; return in ? 1 : 0;
}

// FresnelHelper::reflect / refract about a normal (Math/FresnelHelper.h:144-155): the microfacet BSDFs' outgoing directions
void ref_reflect_about(const float* wi, const float* n, float* out) {
    auto r = FresnelHelper::reflect(NormalizedT<Vec3f>(Vec3f(wi[0], wi[1], wi[2])), NormalizedT<Vec3f>(Vec3f(n[0], n[1], n[2]))); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_refract_about(const float* wi, const float* n, float eta, float cosThetaT, float* out) {
    Vec3f r = FresnelHelper::refract(Vec3f(wi[0], wi[1], wi[2]), Vec3f(n[0], n[1], n[2]), eta, cosThetaT); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
} // extern "C"
