// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).  C entry points for tests/ and bench.py's cpu_baseline.
#include "ocore.h"
#include "orng.h"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <vector>

using namespace orc;

extern "C" {

// ---- primitives -----------------------------------------------------------------------------------------------
void orc_woop_set_data(const float* v0, const float* v1, const float* v2, ctl_woop_tri* out) {
    woopSetData(*out, V3(v0[0], v0[1], v0[2]), V3(v1[0], v1[1], v1[2]), V3(v2[0], v2[1], v2[2]));
}
void orc_woop_get_data(const ctl_woop_tri* w, float* v0, float* v1, float* v2) {
    V3 a, b, c; woopGetData(*w, a, b, c);
    v0[0] = a.x; v0[1] = a.y; v0[2] = a.z; v1[0] = b.x; v1[1] = b.y; v1[2] = b.z; v2[0] = c.x; v2[1] = c.y; v2[2] = c.z;
}
int orc_woop_intersect(const ctl_woop_tri* w, const float* o, const float* d, float tmin, float tmax, float* tuv) {
    return woopIntersect(*w, V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmin, tmax, tuv[0], tuv[1], tuv[2]) ? 1 : 0;
}
uint16_t orc_float_to_half(float f) { return floatToHalf(f); }
float orc_half_to_float(uint16_t h, int host_quirk) { return halfToFloat(h, host_quirk != 0); }
uint16_t orc_normal_encode(const float* n) { return normalToUchar2(V3(n[0], n[1], n[2])); }
void orc_normal_decode(uint16_t v, float* n) { V3 r = uchar2ToNormal(v); n[0] = r.x; n[1] = r.y; n[2] = r.z; }
// batch variants (the exhaustive codec tests)
void orc_half_to_float_n(const uint16_t* h, uint32_t n, int host_quirk, float* out) { for (uint32_t i = 0; i < n; i++) out[i] = halfToFloat(h[i], host_quirk != 0); }
void orc_float_to_half_n(const float* f, uint32_t n, uint16_t* out) { for (uint32_t i = 0; i < n; i++) out[i] = floatToHalf(f[i]); }
void orc_normal_decode_n(const uint16_t* v, uint32_t n, float* out) { for (uint32_t i = 0; i < n; i++) { V3 r = uchar2ToNormal(v[i]); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; } }
void orc_normal_encode_n(const float* nrm, uint32_t n, uint16_t* out) { for (uint32_t i = 0; i < n; i++) out[i] = normalToUchar2(V3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2])); }
void orc_matrix_inverse(const float* m, float* out) { M44 a; std::memcpy(a.d, m, 64); M44 r = inverse(a); std::memcpy(out, r.d, 64); }

// TriangleData(P, matIndex, T, N) (Engine/TriangleData.cu:11-16)
void orc_triangle_data_pack(const float* P, const float* N, const float* T, uint32_t mat_index, int quirk, ctl_triangle_data* out) {
    std::memset(out, 0, sizeof(*out));
    out->nor_mat_extra[1] = (mat_index & 0xff) << 16;   // m_sHostData.MatIndex is byte 6
    triDataSetUV(*out, V2{ T[0], T[1] }, V2{ T[2], T[3] }, V2{ T[4], T[5] });
    triDataSetData(*out, V3(P[0], P[1], P[2]), V3(P[3], P[4], P[5]), V3(P[6], P[7], P[8]), V3(N[0], N[1], N[2]), V3(N[3], N[4], N[5]), V3(N[6], N[7], N[8]), quirk != 0);
}
// out[0..2]=P is not touched; out = sys.s(3) sys.t(3) sys.n(3) n(3) dpdu(3) dpdv(3) uv(2) extra(1) = 21 floats
void orc_triangle_fill_dg(const ctl_triangle_data* T, const float* local_to_world, float u, float v, int quirk, float* out) {
    M44 m; std::memcpy(m.d, local_to_world, 64);
    DG dg; dg.bary = V2{ u, v };
    triDataFillDG(*T, m, dg, quirk != 0);
    const V3* vs[6] = { &dg.sys.s, &dg.sys.t, &dg.sys.n, &dg.n, &dg.dpdu, &dg.dpdv };
    for (int i = 0; i < 6; i++) { out[i * 3] = vs[i]->x; out[i * 3 + 1] = vs[i]->y; out[i * 3 + 2] = vs[i]->z; }
    out[18] = dg.uv.x; out[19] = dg.uv.y; out[20] = (float)dg.extraData;
}

// ---- warps / fresnel ------------------------------------------------------------------------------------------
void orc_square_to_cosine_hemisphere(float x, float y, float* out) { V3 r = squareToCosineHemisphere(V2{ x, y }); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_square_to_uniform_triangle(float x, float y, float* out) { V2 r = squareToUniformTriangle(V2{ x, y }); out[0] = r.x; out[1] = r.y; }
void orc_reflect_about(const float* wi, const float* n, float* out) { V3 r = reflectAbout(V3(wi[0], wi[1], wi[2]), V3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_refract_about(const float* wi, const float* n, float eta, float cosThetaT, float* out) { V3 r = refractAbout(V3(wi[0], wi[1], wi[2]), V3(n[0], n[1], n[2]), eta, cosThetaT); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_slab_probe(int on, uint64_t* tests_rejects) {   // tools/bvh_quality_probe.py --slab: switch the probe of traceRayFlat on / off, read its two counters
    if (tests_rejects) { tests_rejects[0] = g_slab_tests.load(); tests_rejects[1] = g_slab_rejects.load(); }
    g_slab_probe = on != 0; if (on) { g_slab_tests = 0; g_slab_rejects = 0; for (auto& t : g_top_probe) t = 0; }
}
void orc_top_probe_read(uint64_t* out8) { for (int b = 0; b < 8; b++) out8[b] = g_top_probe[b].load(); }   // visits of nodes with index < {85, 256, 341, 512, 1365, 5461, 65536, all}
float orc_interval_to_tent(float s) { return intervalToTent(s); }
float orc_cosine_hemisphere_pdf(const float* d) { return squareToCosineHemispherePdf(V3(d[0], d[1], d[2])); }
void orc_square_to_uniform_sphere(float x, float y, float* out) { V3 r = squareToUniformSphere(V2{ x, y }); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
int orc_barycentric(const float* p, const float* a, const float* b, const float* c, float* uv) {
    float u = 0, v = 0; const bool in = barycentric(V3(p[0], p[1], p[2]), V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]), V3(c[0], c[1], c[2]), u, v); uv[0] = u; uv[1] = v; return in ? 1 : 0;
}
void orc_square_to_uniform_disk_concentric(float x, float y, float* out) { V2 r = squareToUniformDiskConcentric(V2{ x, y }); out[0] = r.x; out[1] = r.y; }
float orc_fresnel_dielectric_ext(float cosThetaI, float eta, float* cosThetaT) { return fresnelDielectricExt(cosThetaI, *cosThetaT, eta); }
void orc_fresnel_conductor_exact(float cosThetaI, const float* eta, const float* k, float* out) {
    Spec r = fresnelConductorExact(cosThetaI, Spec(eta[0], eta[1], eta[2]), Spec(k[0], k[1], k[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_coordinate_system(const float* a, float* s, float* t) { V3 S, T; coordinateSystem(V3(a[0], a[1], a[2]), S, T); s[0] = S.x; s[1] = S.y; s[2] = S.z; t[0] = T.x; t[1] = T.y; t[2] = T.z; }

// microfacet: out = eval(m), smithG1(wi,m), pdf(wi,m)
void orc_microfacet_eval(int type, float aU, float aV, int sampleVisible, const float* wi, const float* m, float* out) {
    Microfacet d(type, aU, aV, sampleVisible != 0);
    V3 WI(wi[0], wi[1], wi[2]), M(m[0], m[1], m[2]);
    out[0] = d.eval(M); out[1] = d.smithG1(WI, M); out[2] = d.pdf(WI, M);
}
void orc_microfacet_sample(int type, float aU, float aV, int sampleVisible, const float* wi, float sx, float sy, float* out) {
    Microfacet d(type, aU, aV, sampleVisible != 0);
    float pdf; V3 m = d.sample(V3(wi[0], wi[1], wi[2]), V2{ sx, sy }, pdf);
    out[0] = m.x; out[1] = m.y; out[2] = m.z; out[3] = pdf;
}

// ---- sampler --------------------------------------------------------------------------------------------------
void* orc_seqgen_create() { return new SequenceGenerator(); }
void orc_seqgen_destroy(void* g) { delete (SequenceGenerator*)g; }
void orc_seqgen_compute(void* g, float* t1, float* t2) { ((SequenceGenerator*)g)->compute(t1, t2); }
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint32_t* state6) {
    Xorwow s = xorwowInit(seed, subsequence); state6[0] = s.d; for (int i = 0; i < 5; i++) state6[1 + i] = s.v[i];
}
// 800 words of T^(2^log2pow) in the cuRAND / rocRAND precalc layout (row (i*32+j) = image of bit j of word i)
void orc_xorwow_jump_matrix(int log2pow, uint32_t* out800) { XMat M = xmatPow2(log2pow); std::memcpy(out800, M.r, sizeof(M.r)); }
float orc_sampler_float(const float* t1, const float* t2, uint32_t idx, uint32_t d1) { Sampler s(t1, t2, idx); s.d1 = d1; return s.randomFloat(); }
void orc_sampler_float2(const float* t1, const float* t2, uint32_t idx, uint32_t d2, float* out) { Sampler s(t1, t2, idx); s.d2 = d2; V2 r = s.randomFloat2(); out[0] = r.x; out[1] = r.y; }

// ---- sensor ---------------------------------------------------------------------------------------------------
void orc_sensor_sample_ray(const ctl_sensor* s, float px, float py, float* o, float* d) {
    PerspectiveSensor ps; ps.update(*s); V3 O, D; ps.sampleRay(V2{ px, py }, V2{ 0.0f, 0.0f }, O, D);
    o[0] = O.x; o[1] = O.y; o[2] = O.z; d[0] = D.x; d[1] = D.y; d[2] = D.z;
}
// every sensor type, with the aperture sample; out18 = sampleRay (o, d), then sampleRayDifferential's x ray (o, d) and y ray (o, d); out6b = sampleRayDifferential's own ray
void orc_sensor_sample_rays(const ctl_sensor* s, float px, float py, float ax, float ay, float* out18, float* out6b) {
    SensorO ps; ps.update(*s); V3 O, D, o2, d2, oX, dX, oY, dY;
    ps.sampleRay(V2{ px, py }, V2{ ax, ay }, O, D);
    ps.sampleRayDifferential(V2{ px, py }, V2{ ax, ay }, o2, d2, oX, dX, oY, dY);
    const V3 v[6] = { O, D, oX, dX, oY, dY };
    for (int k = 0; k < 6; k++) { out18[3 * k] = v[k].x; out18[3 * k + 1] = v[k].y; out18[3 * k + 2] = v[k].z; }
    out6b[0] = o2.x; out6b[1] = o2.y; out6b[2] = o2.z; out6b[3] = d2.x; out6b[4] = d2.y; out6b[5] = d2.z;
}

// ---- intersect ------------------------------------------------------------------------------------------------
// intersectKernel semantics (TraceHelper.cu:326-734): tmin = ray.a.w at node and triangle level, tmax = ray.b.w.
// any_hit: bit 0 = first hit ends the ray, bit 1 = alpha-test candidate hits (traceRay<USE_ALPHA>, TraceHelper.cu:135-153)
// the flattened BVH the following orc_intersect / orc_render calls traverse instead of the two-level structure (NULL = two-level, the reference's).
// The arrays are the product's (ctl_flat_bvh_arrays): "the CPU restatement in counting mode with the same BVH" (SURVEY §8d).
static const ctl_flat_bvh_desc* g_flat = nullptr;
void orc_set_flat_bvh(const ctl_flat_bvh_desc* f) { g_flat = f; }
void orc_intersect(const ctl_scene_desc* desc, const ctl_ray* rays, uint32_t n, ctl_hit* hits, int any_hit, ctl_traversal_counts* counts, int n_threads) {
    Scene S; S.d = *desc; S.alpha_test = (any_hit & 2) != 0; any_hit &= 1; S.flat = g_flat;
    if (n_threads < 1) n_threads = 1;
    std::vector<TravCounts> tc(n_threads);
    auto work = [&](int tid) {
        for (uint32_t i = tid; i < n; i += n_threads) {
            Hit h;
            traceRay(S, V3(rays[i].a[0], rays[i].a[1], rays[i].a[2]), V3(rays[i].b[0], rays[i].b[1], rays[i].b[2]), rays[i].a[3], rays[i].b[3], any_hit != 0, rays[i].a[3], h,
                     counts ? &tc[tid] : nullptr);
            hits[i].dist = h.dist; hits[i].node_idx = (int32_t)h.node; hits[i].tri_idx = (int32_t)h.tri; hits[i].u = h.u; hits[i].v = h.v;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    if (counts) { counts->n_inner = counts->n_tri = counts->n_inst = 0; for (auto& c : tc) { counts->n_inner += c.n_inner; counts->n_tri += c.n_tri; counts->n_inst += c.n_inst; } }
}

// ---- first-hit ray differentials and filtered texture lookup (probes for the tests)
// levels and offsets of the pyramid of an image, and the pyramid's texels (caller: n_texels_cap words); returns the number of texels
uint32_t orc_mip_pyramid(const ctl_mipmap* M, uint32_t* levels_out, uint32_t* offsets16_out, uint32_t* texels_out, uint32_t cap) {
    MipPyramid P; P.build(*M);
    *levels_out = P.levels; std::memcpy(offsets16_out, P.offsets, 64);
    if (texels_out) std::memcpy(texels_out, P.texels.data(), 4 * std::min<size_t>(cap, P.texels.size()));
    return (uint32_t)P.texels.size();
}
void orc_mip_eval(const ctl_mipmap* M, float u, float v, const float* d0, const float* d1, float* rgb) {
    MipPyramid P; P.build(*M);
    const Spec s = mipEval(*M, P, V2{ u, v }, V2{ d0[0], d0[1] }, V2{ d1[0], d1[1] });
    rgb[0] = s.x; rgb[1] = s.y; rgb[2] = s.z;
}
// the lookups of KernelMIPMap one by one (same `what` / args as oracle/ref_mipmap_driver.cpp ref_mipmap_query): 0 Texel(level, uv), 1 triangle(level, uv),
// 2 evalEWA(level, uv, A, B, C), 3 eval(uv, d0, d1), 4 Sample(uv), 5 Sample(uv, width), 6 SampleAlpha(uv), 7 Sample(width, x, y); lut_out: the 64 EWA weights
void orc_mip_query(const ctl_mipmap* M, int what, int n, const float* args, float* out3, float* lut_out) {
    MipPyramid P; P.build(*M);
    if (lut_out) std::memcpy(lut_out, mipWeightLut(), 256);
    for (int i = 0; i < n; i++) {
        const float* a = args + 8 * i; const V2 uv{ a[0], a[1] }; const uint32_t level = (uint32_t)a[6];
        Spec s(0.0f);
        switch (what) {
        case 0: s = mipTexelL(*M, P, level, uv); break;
        case 1: s = mipTriangleL(*M, P, level, uv); break;
        case 2: s = mipEvalEWA(*M, P, level, uv, a[2], a[3], a[4]); break;
        case 3: s = mipEval(*M, P, uv, V2{ a[2], a[3] }, V2{ a[4], a[5] }); break;
        case 4: s = mipSample(*M, uv); break;
        case 5: s = mipSampleWidth(*M, P, uv, a[2]); break;
        case 6: s = Spec(mipSampleAlpha(*M, uv)); break;
        case 7: s = mipFetchL(*M, P, a[2], (int)a[3], (int)a[4]); break;
        }
        out3[3 * i] = s.x; out3[3 * i + 1] = s.y; out3[3 * i + 2] = s.z;
    }
}
// in: P, n, dpdu, dpdv (3 floats each), ray origin, directions of the x / y differential rays; out: dudx, dudy, dvdx, dvdy
void orc_compute_partials(const float* P, const float* n, const float* dpdu, const float* dpdv, const float* ro, const float* rxd, const float* ryd, float* out4) {
    DG dg; dg.P = V3(P[0], P[1], P[2]); dg.n = V3(n[0], n[1], n[2]); dg.dpdu = V3(dpdu[0], dpdu[1], dpdu[2]); dg.dpdv = V3(dpdv[0], dpdv[1], dpdv[2]);
    computePartials(dg, V3(ro[0], ro[1], ro[2]), V3(rxd[0], rxd[1], rxd[2]), V3(ro[0], ro[1], ro[2]), V3(ryd[0], ryd[1], ryd[2]));
    out4[0] = dg.dudx; out4[1] = dg.dudy; out4[2] = dg.dvdx; out4[3] = dg.dvdy;
}
// the same with separate origins of the two differential rays (orthographic / telecentric sensors)
void orc_compute_partials_origins(const float* P, const float* n, const float* dpdu, const float* dpdv, const float* rox, const float* rxd, const float* roy, const float* ryd, float* out4) {
    DG dg; dg.P = V3(P[0], P[1], P[2]); dg.n = V3(n[0], n[1], n[2]); dg.dpdu = V3(dpdu[0], dpdu[1], dpdu[2]); dg.dpdv = V3(dpdv[0], dpdv[1], dpdv[2]);
    computePartials(dg, V3(rox[0], rox[1], rox[2]), V3(rxd[0], rxd[1], rxd[2]), V3(roy[0], roy[1], roy[2]), V3(ryd[0], ryd[1], ryd[2]));
    out4[0] = dg.dudx; out4[1] = dg.dudy; out4[2] = dg.dvdx; out4[3] = dg.dvdy;
}
void orc_sensor_sample_ray_differential(const ctl_sensor* s, float px, float py, float* o, float* d, float* dX, float* dY) {
    PerspectiveSensor ps; ps.update(*s); V3 O, D, X, Y, oX, oY; ps.sampleRayDifferential(V2{ px, py }, V2{ 0.0f, 0.0f }, O, D, oX, X, oY, Y);
    o[0] = O.x; o[1] = O.y; o[2] = O.z; d[0] = D.x; d[1] = D.y; d[2] = D.z; dX[0] = X.x; dX[1] = X.y; dX[2] = X.z; dY[0] = Y.x; dY[1] = Y.y; dY[2] = Y.z;
}

// ---- BSDF / light probes --------------------------------------------------------------------------------------
// tables used by the BSDF probes below (roughplastic); the pointer must stay valid
static const ctl_rough_transmittance* g_probe_rt = nullptr;
void orc_set_probe_rough_transmittance(const ctl_rough_transmittance* t3) { g_probe_rt = t3; }
static const ctl_material* g_probe_mats = nullptr;   // material array the nested indices of coating / roughcoating / blend refer to
void orc_set_probe_materials(const ctl_material* m) { g_probe_mats = m; }
float orc_rough_transmittance_eval(uint32_t slot, float cosTheta, float alpha, float eta) { DG dg; dg.rough_transmittance = g_probe_rt; return roughTransmittance(dg, slot, cosTheta, alpha, eta); }
// the spline restatement alone (obsdf3.h evalCubicInterp2D / 3D), held against the reference's Math/Spline.cu through tests/golden/spline.npz
float orc_spline_eval_2d(float px, float py, const float* values, uint32_t sx, uint32_t sy) { return evalCubicInterp2D(px, py, values, sx, sy); }
float orc_spline_eval_3d(float px, float py, float pz, const float* values, uint32_t sx, uint32_t sy, uint32_t sz) { return evalCubicInterp3D(px, py, pz, values, sx, sy, sz); }
float orc_rough_transmittance_eval_diffuse(uint32_t slot, float alpha, float eta) { DG dg; dg.rough_transmittance = g_probe_rt; return roughTransmittanceDiffuse(dg, slot, alpha, eta); }
// local-frame probe: dg is an identity frame at the origin with uv = (u,v).  out = f(3), pdf, wo(3), sampledType, eta
void orc_bsdf_sample(const ctl_material* M, const float* wi, float sx, float sy, float* out) {
    BRec b; b.dg.P = V3(0.0f); b.dg.sys = Frame(V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1)); b.dg.n = V3(0, 0, 1); b.dg.uv = V2{ 0, 0 };
    b.wi = V3(wi[0], wi[1], wi[2]); b.wo = V3(0.0f); b.eta = 1.0f; b.typeMask = EAll; b.sampledType = 0; b.dg.rough_transmittance = g_probe_rt; b.dg.materials = g_probe_mats;
    float pdf = 0; Spec f = bsdfSample(*M, b, pdf, V2{ sx, sy });
    out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = pdf; out[4] = b.wo.x; out[5] = b.wo.y; out[6] = b.wo.z; out[7] = (float)b.sampledType; out[8] = b.eta;
}
void orc_bsdf_eval(const ctl_material* M, const float* wi, const float* wo, uint32_t typeMask, float* out) {
    BRec b; b.dg.P = V3(0.0f); b.dg.sys = Frame(V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1)); b.dg.n = V3(0, 0, 1); b.dg.uv = V2{ 0, 0 };
    b.wi = V3(wi[0], wi[1], wi[2]); b.wo = V3(wo[0], wo[1], wo[2]); b.eta = 1.0f; b.typeMask = typeMask; b.sampledType = 0; b.dg.rough_transmittance = g_probe_rt; b.dg.materials = g_probe_mats;
    Spec f = bsdfF(*M, b); out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = bsdfPdf(*M, b);
}
// the same probes with a texture coordinate (checkerboard textures) and the measure as an argument (mode 1 solid angle, 2 discrete): the shape of oracle/ref_bsdf_driver.cpp
void orc_bsdf_sample_uv(const ctl_material* M, const float* wi, float sx, float sy, float u, float v, float* out) {
    BRec b; b.dg.P = V3(0.0f); b.dg.sys = Frame(V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1)); b.dg.n = V3(0, 0, 1); b.dg.uv = V2{ u, v };
    b.wi = V3(wi[0], wi[1], wi[2]); b.wo = V3(0.0f); b.eta = 1.0f; b.typeMask = EAll; b.sampledType = 0; b.dg.rough_transmittance = g_probe_rt; b.dg.materials = g_probe_mats;
    float pdf = 0; Spec f = bsdfSample(*M, b, pdf, V2{ sx, sy });
    out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = pdf; out[4] = b.wo.x; out[5] = b.wo.y; out[6] = b.wo.z; out[7] = (float)b.sampledType; out[8] = b.eta;
}
void orc_bsdf_eval_uv(const ctl_material* M, const float* wi, const float* wo, uint32_t typeMask, int mode, float u, float v, float* out) {
    BRec b; b.dg.P = V3(0.0f); b.dg.sys = Frame(V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1)); b.dg.n = V3(0, 0, 1); b.dg.uv = V2{ u, v };
    b.wi = V3(wi[0], wi[1], wi[2]); b.wo = V3(wo[0], wo[1], wo[2]); b.eta = 1.0f; b.typeMask = typeMask; b.sampledType = 0; b.dg.rough_transmittance = g_probe_rt; b.dg.materials = g_probe_mats;
    const int measure = mode == 2 ? EDiscrete : ESolidAngle;
    Spec f = bsdfF(*M, b, measure); out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = bsdfPdf(*M, b, measure);
}
// f / pdf with the discrete measure (delta lobes; what the nesting BSDFs ask their children for)
void orc_bsdf_eval_discrete(const ctl_material* M, const float* wi, const float* wo, uint32_t typeMask, float* out) {
    BRec b; b.dg.P = V3(0.0f); b.dg.sys = Frame(V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1)); b.dg.n = V3(0, 0, 1); b.dg.uv = V2{ 0, 0 };
    b.wi = V3(wi[0], wi[1], wi[2]); b.wo = V3(wo[0], wo[1], wo[2]); b.eta = 1.0f; b.typeMask = typeMask; b.sampledType = 0; b.dg.rough_transmittance = g_probe_rt; b.dg.materials = g_probe_mats;
    Spec f = bsdfF(*M, b, EDiscrete); out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = bsdfPdf(*M, b, EDiscrete);
}
// out = value(3), pdf, d(3), dist, p(3), n(3)
void orc_light_sample_direct(const ctl_scene_desc* desc, uint32_t light, const float* ref, const float* refN, float sx, float sy, float* out) {
    Scene S; S.d = *desc;
    DirectRec d(V3(ref[0], ref[1], ref[2]), V3(refN[0], refN[1], refN[2]));
    Spec v = lightSampleDirect(S, desc->lights[light], d, V2{ sx, sy });
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = d.pdf; out[4] = d.d.x; out[5] = d.d.y; out[6] = d.d.z; out[7] = d.dist;
    out[8] = d.p.x; out[9] = d.p.y; out[10] = d.p.z; out[11] = d.n.x; out[12] = d.n.y; out[13] = d.n.z;
}

// KernelDynamicScene::sampleEmitter for nq samples (2 floats each) over desc's light list: buffer slot of the chosen light, emPdf, the re-scaled sample.x;
// pdf_emitter_out[i] = pdfEmitter(&lights[i]) for the first n_buf buffer slots (the reference indexes the CDF by BUFFER slot there, KernelDynamicScene.cu:42-46)
void orc_emitter_select(const ctl_scene_desc* desc, int n_buf, int nq, const float* samples, int32_t* slot_out, float* pdf_out, float* resampled_out, float* pdf_emitter_out) {
    Scene S; S.d = *desc;
    for (int i = 0; i < nq; i++) {
        V2 s{ samples[2 * i], samples[2 * i + 1] }; float emPdf = 0.0f;
        const ctl_light* L = sampleEmitter(S, emPdf, s);
        slot_out[i] = L ? (int32_t)(L - desc->lights) : -1; pdf_out[i] = emPdf; resampled_out[i] = s.x;
    }
    for (int i = 0; i < n_buf; i++) pdf_emitter_out[i] = pdfEmitter(S, desc->lights + i);
}
// KernelDynamicScene::sampleEmitterDirect: out = value(3), pdf, d(3), dist, p(3), n(3), buffer slot of dRec.object or -1
void orc_sample_emitter_direct(const ctl_scene_desc* desc, const float* ref, const float* refN, float sx, float sy, float* out) {
    Scene S; S.d = *desc;
    DirectRec d(V3(ref[0], ref[1], ref[2]), V3(refN[0], refN[1], refN[2]));
    const ctl_light* obj = nullptr;
    Spec v = sampleEmitterDirect(S, d, V2{ sx, sy }, &obj);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = d.pdf; out[4] = d.d.x; out[5] = d.d.y; out[6] = d.d.z; out[7] = d.dist;
    out[8] = d.p.x; out[9] = d.p.y; out[10] = d.p.z; out[11] = d.n.x; out[12] = d.n.y; out[13] = d.n.z; out[14] = obj ? (float)(obj - desc->lights) : -1.0f;
}

// Light::pdfDirect for a direction d seen from ref (solid-angle measure); p / n / dist describe the emitter point (area lights)
float orc_light_pdf_direct(const ctl_scene_desc* desc, uint32_t light, const float* ref, const float* refN, const float* d, float dist, const float* n) {
    Scene S; S.d = *desc;
    DirectRec r(V3(ref[0], ref[1], ref[2]), V3(refN[0], refN[1], refN[2]));
    r.d = V3(d[0], d[1], d[2]); r.dist = dist; r.n = V3(n[0], n[1], n[2]); r.measure = ESolidAngle;
    return lightPdfDirect(S, desc->lights[light], r);
}
// DiffuseLight::eval(p, Frame(n), d) (SceneTypes/Light.cu:67-81)
void orc_light_eval(const ctl_scene_desc* desc, uint32_t light, const float* p, const float* n, const float* d, float* out) {
    Scene S; S.d = *desc;
    Spec v = lightEval(S, desc->lights[light], V3(p[0], p[1], p[2]), Frame(V3(n[0], n[1], n[2])), V3(d[0], d[1], d[2]));
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
// InfiniteLight::evalEnvironment for a world direction
void orc_env_eval(const ctl_scene_desc* desc, const float* dir, float* out) {
    Scene S; S.d = *desc;
    Spec v(0.0f);
    if (desc->env_map_index != 0xffffffffu) v = envEval(S, desc->lights[desc->env_map_index], V3(dir[0], dir[1], dir[2]));
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
// ImageTexture / checkerboard / constant evaluation at a uv (Texture::Evaluate(dg))
int orc_wrap_coordinates(float u, float v, float w, float h, uint32_t mode, float* loc) { V2 l{ 0.0f, 0.0f }; const bool ok = wrapCoordinates(V2{ u, v }, V2{ w, h }, mode, l); loc[0] = l.x; loc[1] = l.y; return ok ? 1 : 0; }
void orc_texture_eval(const ctl_scene_desc* desc, const ctl_texture* t, float u, float v, float* out) {
    DG dg; dg.uv = V2{ u, v }; dg.images = desc ? desc->images : nullptr;
    Spec s = texEval(*t, dg); out[0] = s.x; out[1] = s.y; out[2] = s.z;
}

// Material::SampleNormalMap on a hand-made shading point: frame_io = s,t,n (9 floats, replaced by the perturbed frame),
// geo = geometric normal, dpdu, dpdv (9 floats).  Returns whether a map was applied.
int orc_sample_normal_map(const ctl_scene_desc* desc, const ctl_material* mat, float u, float v, float* frame_io, const float* geo) {
    DG dg; dg.uv = V2{ u, v }; dg.images = desc ? desc->images : nullptr;
    dg.sys = Frame(V3(frame_io[0], frame_io[1], frame_io[2]), V3(frame_io[3], frame_io[4], frame_io[5]), V3(frame_io[6], frame_io[7], frame_io[8]));
    dg.n = V3(geo[0], geo[1], geo[2]); dg.dpdu = V3(geo[3], geo[4], geo[5]); dg.dpdv = V3(geo[6], geo[7], geo[8]);
    const bool r = sampleNormalMap(*mat, dg);
    const V3 f[3] = { dg.sys.s, dg.sys.t, dg.sys.n };
    for (int i = 0; i < 3; i++) { frame_io[3 * i] = f[i].x; frame_io[3 * i + 1] = f[i].y; frame_io[3 * i + 2] = f[i].z; }
    return r ? 1 : 0;
}
int orc_alpha_test(const ctl_scene_desc* desc, const ctl_material* mat, float u, float v) { return materialAlphaTest(*mat, V2{ u, v }, desc ? desc->images : nullptr) ? 1 : 0; }

// ---- full render: pathKernel2<DIRECT,false> looped over all pixels (Integrators/PathTracer.cu:182-194) ---------
// samples per 64x64 block for the following orc_render calls (a block sampler's decision for one pass); NULL = one sample everywhere
static const uint8_t* g_block_counts = nullptr; static uint32_t g_blocks_x = 0;
void orc_set_block_counts(const uint8_t* counts, uint32_t blocks_x) { g_block_counts = counts; g_blocks_x = blocks_x; }
// side image of the following orc_render calls (same size as the frame; NULL = none): the samples the reference DROPS as NaN / infinite although the path's throughput had
// become exactly zero before — for each, the radiance collected up to that vertex (ocore.h ZeroStop), added as Image::AddSample would.  The product's kernels end a
// zero-throughput path at once and count that radiance: their frame = the oracle's frame + this image, weights included (tests/test_gpu_fuzz.py).
static ctl_pixel_data* g_zero_stop_img = nullptr;
void orc_set_zero_stop_image(ctl_pixel_data* img) { g_zero_stop_img = img; }
// counting mode of orc_render: traversal statistics of every ray the following renders trace.  out8 = {path rays, n_inner, n_tri, n_inst,
// occlusion rays, n_inner, n_tri, n_inst}; orc_render_counts(NULL) switches counting off, a non-NULL call reads and resets the totals.
static bool g_count_render = false; static uint64_t g_render_counts[8] = {};
void orc_render_counting(int on) { g_count_render = on != 0; for (auto& v : g_render_counts) v = 0; }
void orc_render_counts(uint64_t* out8) { for (int i = 0; i < 8; i++) { out8[i] = g_render_counts[i]; g_render_counts[i] = 0; } }
// tables: n_passes consecutive (t1[30*4096], t2[30*4096*2]) pairs, or NULL -> own SequenceGenerator
// (one Compute() per pass, as Tracer<true>::DoPass -> UpdateKernel does, Kernel/Tracer.h:229).
// Renders rows [y0,y1) only (bounded CPU-baseline samples).  Returns the number of rays traced.
uint64_t orc_render(const ctl_scene_desc* desc, uint32_t W, uint32_t H, uint32_t n_passes, const float* tables1, const float* tables2,
                    int direct, int maxPathLength, int rrStart, ctl_pixel_data* img, int n_threads, uint32_t y0, uint32_t y1, int half_host_quirk) {
    // half_host_quirk: bit 0 = half::ToFloat host branch, bit 1 = alpha test on (doAlphaMapping: every traceRay, incl. Occluded),
    // bit 2 = first-hit ray differentials and filtered texture lookups (what the megakernel PathTracer does, PathTracer.cu:60-61; the wavefront tracer does not),
    // bit 3 = PathTraceRegularization (the PathTracer plugin's Regularization = true; implies bit 2), pass k (1-based) with the mollifier of PathTracer::RenderBlock
    Scene S; S.d = *desc; S.half_host_quirk = (half_host_quirk & 1) != 0; S.alpha_test = (half_host_quirk & 2) != 0 && sceneHasAlphaMaps(*desc); S.flat = g_flat;
    PerspectiveSensor sensor; sensor.update(desc->camera);
    std::vector<MipPyramid> pyramids;
    const bool regularization = (half_host_quirk & 8) != 0;
    const bool wavefront_rules = (half_host_quirk & 16) != 0, u16bary = (half_host_quirk & 32) != 0, omitLastNEE = (half_host_quirk & 64) != 0;   // bit 4: pathIterateKernel's own path rules (pathTraceWavefront), bit 5: 16-bit barycentrics
    const bool partials = (half_host_quirk & 4) != 0 || regularization;
    if (partials) { pyramids.resize(desc->n_images); for (uint32_t i = 0; i < desc->n_images; i++) pyramids[i].build(desc->images[i]); S.pyramids = pyramids.data(); }
    if (n_threads < 1) n_threads = 1;
    if (y1 > H) y1 = H;
    SequenceGenerator gen;
    const size_t N1 = (size_t)CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH, N2 = N1 * 2;
    std::atomic<uint64_t> total(0);
    std::mutex cmu;
    // Persistent worker threads for the whole call, 64 x 16 pixel tiles handed out dynamically inside a pass; passes stay sequential (two passes
    // never add to one pixel at the same time), a reusable barrier separates them.
    const uint32_t TW = 64, TH = 16, tiles_x = (W + TW - 1) / TW, tiles_y = (y1 > y0) ? (y1 - y0 + TH - 1) / TH : 0, n_tiles = tiles_x * tiles_y;
    std::vector<std::vector<float>> gen1, gen2;
    if (!tables1) { gen1.resize(n_passes); gen2.resize(n_passes); for (uint32_t p = 0; p < n_passes; p++) { gen1[p].resize(N1); gen2[p].resize(N2); gen.compute(gen1[p].data(), gen2[p].data()); } }
    std::vector<std::atomic<uint32_t>> nextTile(n_passes);
    for (auto& a : nextTile) a.store(0);
    std::mutex bmu; std::condition_variable bcv; uint32_t arrived = 0, generation = 0;
    auto barrier = [&]() {
        std::unique_lock<std::mutex> l(bmu);
        const uint32_t g = generation;
        if (++arrived == (uint32_t)n_threads) { arrived = 0; generation++; bcv.notify_all(); }
        else bcv.wait(l, [&] { return generation != g; });
    };
    auto work = [&]() {
        uint64_t rays = 0;
        RenderCounts rc; renderCounts() = g_count_render ? &rc : nullptr;
        for (uint32_t pass = 0; pass < n_passes; pass++) {
            const float* t1 = tables1 ? tables1 + pass * N1 : gen1[pass].data();
            const float* t2 = tables1 ? tables2 + pass * N2 : gen2[pass].data();
            for (;;) {
                const uint32_t tile = nextTile[pass].fetch_add(1);
                if (tile >= n_tiles) break;
                const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
                const uint32_t xa = tx * TW, xb = std::min(W, xa + TW), ya = y0 + ty * TH, yb = std::min(y1, ya + TH);
                for (uint32_t y = ya; y < yb; y++)
                for (uint32_t x = xa; x < xb; x++)
                for (uint32_t smp = 0, n_smp = g_block_counts ? g_block_counts[(y / 64) * g_blocks_x + x / 64] : 1u; smp < n_smp; smp++) {
                    // BlockSamplerBuffer::getNumSamplesPerPixel (WavefrontPathTracer.cu:31-36): the samples of a pixel in one pass continue one sampler
                    Sampler rng(t1, t2, y * W + x);   // TracerBase::getPixelIndex
                    rng.d2 = 2 * smp;
                    V2 j = rng.randomFloat2();
                    V2 pX{ (float)x + j.x, (float)y + j.y };
                    V2 ap = rng.randomFloat2();
                    V3 o, d; RayDiff diff;
                    if (partials) sensor.sampleRayDifferential(pX, ap, o, d, diff.ox, diff.dx, diff.oy, diff.dy); else sensor.sampleRay(pX, ap, o, d);
                    Spec col;
                    if (regularization) {   // PathTracer::RenderBlock (PathTracer.cu:196-203): radius2 from the scene box and the passes done (this pass included)
                        const float initialRadius = ((desc->box_max[0] - desc->box_min[0]) + (desc->box_max[1] - desc->box_min[1]) + (desc->box_max[2] - desc->box_min[2])) / 100;
                        const float ALPHA = 0.75f;
                        const float radius2 = powf(powf(initialRadius, float(2)) / powf(float(pass + 1), 0.5f * (1 - ALPHA)), 1.0f / 2.0f);
                        col = pathTraceRegularization(S, direct != 0, o, d, diff, rng, radius2, maxPathLength, rrStart, &rays);
                    } else if (wavefront_rules) col = pathTraceWavefront(S, direct != 0, o, d, rng, maxPathLength, rrStart, &rays, u16bary);
                    else
                    col = pathTrace(S, direct != 0, o, d, rng, maxPathLength, rrStart, &rays, partials ? &diff : nullptr, omitLastNEE);   // imp == 1 (Sensor.cu:127)
                    addSample(img, (int)W, (int)H, pX.x, pX.y, col);
                    if (g_zero_stop_img && zeroStop().have) {
                        const Spec c = V3(fmax2(0.0f, col.x), fmax2(0.0f, col.y), fmax2(0.0f, col.z));
                        const bool dropped = std::isnan(c.x) || std::isnan(c.y) || std::isnan(c.z) || std::isinf(c.x) || std::isinf(c.y) || std::isinf(c.z);
                        if (dropped) addSample(g_zero_stop_img, (int)W, (int)H, pX.x, pX.y, zeroStop().cl);
                    }
                }
            }
            if (pass + 1 < n_passes) barrier();
        }
        total += rays;
        renderCounts() = nullptr;
        if (g_count_render) { std::lock_guard<std::mutex> l(cmu); for (int k = 0; k < 2; k++) { g_render_counts[k * 4 + 0] += rc.rays[k]; g_render_counts[k * 4 + 1] += rc.c[k].n_inner; g_render_counts[k * 4 + 2] += rc.c[k].n_tri; g_render_counts[k * 4 + 3] += rc.c[k].n_inst; } }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return total.load();
}

// Image::AddSample (Engine/Image.cu:22-44) over a list of samples, in order: pixels = W x H PixelData, samples = n x {sx, sy, r, g, b} (tests/golden/image.npz holds the
// reference's own function's result for such lists)
void orc_image_add_samples(ctl_pixel_data* pixels, uint32_t W, uint32_t H, int n, const float* samples) {
    for (int i = 0; i < n; i++) addSample(pixels, (int)W, (int)H, samples[5 * i], samples[5 * i + 1], Spec(samples[5 * i + 2], samples[5 * i + 3], samples[5 * i + 4]));
}

// What would a PACKET traversal of the coherent first bounce look at?  (tools/packet_union_probe.py; a feasibility probe, not a product path.)  For `n_blocks` 8 x 8 pixel blocks
// spread over the frame — the 64 primary rays a wave of k_raygen's order holds — the closest-hit traversal of the flattened Q4 tree (orc_set_flat_bvh) is run ray by ray with a
// visit log: out = { rays, sum of node steps, sum of entry tests, sum over blocks of the UNION of nodes looked at, ... of entries, blocks }.
void orc_packet_union_probe(const ctl_scene_desc* desc, uint32_t W, uint32_t H, uint32_t n_blocks, uint64_t* out6) {
    Scene S; S.d = *desc; S.flat = g_flat;
    PerspectiveSensor sensor; sensor.update(desc->camera);
    for (int i = 0; i < 6; i++) out6[i] = 0;
    if (!g_flat) return;
    const uint32_t bx = W / 8, by = H / 8, total = bx * by, step = total / (n_blocks ? n_blocks : 1) ? total / n_blocks : 1;
    std::vector<uint32_t> nodes, entries, un, ue;
    for (uint32_t b = 0; b < total; b += step) {
        un.clear(); ue.clear();
        for (uint32_t p = 0; p < 64; p++) {
            const float x = (float)((b % bx) * 8 + (p & 7)) + 0.5f, y = (float)((b / bx) * 8 + (p >> 3)) + 0.5f;
            V3 o, d; sensor.sampleRay(V2{ x, y }, V2{ 0.5f, 0.5f }, o, d);
            nodes.clear(); entries.clear();
            TravCounts tc; tc.node_log = &nodes; tc.entry_log = &entries;
            Hit h; traceRayFlat(S, o, d, S.d.ray_trace_eps, FLT_MAX, false, S.d.ray_trace_eps, h, &tc);
            out6[0]++; out6[1] += nodes.size(); out6[2] += entries.size();
            un.insert(un.end(), nodes.begin(), nodes.end()); ue.insert(ue.end(), entries.begin(), entries.end());
        }
        std::sort(un.begin(), un.end()); un.erase(std::unique(un.begin(), un.end()), un.end());
        std::sort(ue.begin(), ue.end()); ue.erase(std::unique(ue.begin(), ue.end()), ue.end());
        out6[3] += un.size(); out6[4] += ue.size(); out6[5]++;
    }
}

// debugging aid (tools/fuzz_diag.py): the path of ONE sample of orc_render — pixel (x, y), one pair of sampler tables — vertex by vertex (ocore.h pathLog: 26 floats per vertex);
// returns the number of floats written, rgb = the sample's radiance
int orc_path_log(const ctl_scene_desc* desc, uint32_t W, uint32_t H, const float* t1, const float* t2, uint32_t x, uint32_t y, int direct, int maxPathLength, int rrStart, float* out, int cap, float* rgb) {
    (void)H;
    Scene S; S.d = *desc; S.flat = g_flat;
    PerspectiveSensor sensor; sensor.update(desc->camera);
    Sampler rng(t1, t2, y * W + x);
    V2 j = rng.randomFloat2(); V2 pX{ (float)x + j.x, (float)y + j.y }; V2 ap = rng.randomFloat2();
    V3 o, d; sensor.sampleRay(pX, ap, o, d);
    std::vector<float> log; pathLog() = &log;
    uint64_t rays = 0;
    const Spec col = pathTrace(S, direct != 0, o, d, rng, maxPathLength, rrStart, &rays);
    pathLog() = nullptr;
    rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
    const int n = (int)std::min<size_t>(log.size(), (size_t)cap);
    std::memcpy(out, log.data(), sizeof(float) * n);
    return n;
}

// the transcendental functions this build of the oracle runs its path with (omath.h msin ..: glibc's, or the product's shared ones with -DORC_SHARED_MATH):
// which = 0 sin, 1 cos, 2 tan, 3 acos, 4 atan, 5 atan2(x, y), 6 exp, 7 log, 8 log2, 9 pow(x, y)
void orc_math_eval(int which, int n, const float* x, const float* y, float* out) {
    for (int i = 0; i < n; i++) {
        switch (which) {
        case 0: out[i] = msin(x[i]); break; case 1: out[i] = mcos(x[i]); break; case 2: out[i] = mtan(x[i]); break; case 3: out[i] = macos(x[i]); break;
        case 4: out[i] = matan(x[i]); break; case 5: out[i] = matan2(x[i], y[i]); break; case 6: out[i] = mexp(x[i]); break; case 7: out[i] = mlog(x[i]); break;
        case 8: out[i] = mlog2(x[i]); break; default: out[i] = mpow(x[i], y[i]); break;
        }
    }
}
// PathTracer::DebugInternal (Integrators/PathTracer.cu:172-180): PathTrace<true> for pixel (x, y) from the pixel's own position (no jitter; the aperture sample is the first draw),
// with first-hit ray differentials, over one set of sampling tables -> rgb; also the primary hit distance (FLT_MAX on a miss) for the depth-buffer test
void orc_debug_pixel(const ctl_scene_desc* desc, uint32_t W, uint32_t H, const float* t1, const float* t2, uint32_t x, uint32_t y, int maxPathLength, int rrStart, float* rgb, float* primary_dist) {
    Scene S; S.d = *desc; S.flat = g_flat; S.alpha_test = sceneHasAlphaMaps(*desc);
    PerspectiveSensor sensor; sensor.update(desc->camera);
    std::vector<MipPyramid> pyramids(desc->n_images); for (uint32_t i = 0; i < desc->n_images; i++) pyramids[i].build(desc->images[i]); S.pyramids = pyramids.data();
    Sampler rng(t1, t2, y * W + x);
    V3 o, d; RayDiff diff; sensor.sampleRayDifferential(V2{ (float)x, (float)y }, rng.randomFloat2(), o, d, diff.ox, diff.dx, diff.oy, diff.dy);
    uint64_t rays = 0;
    if (rgb) { const Spec c = pathTrace(S, true, o, d, rng, maxPathLength, rrStart, &rays, &diff); rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z; }
    if (primary_dist) { Hit h = traceRayClosest(S, o, d); *primary_dist = h.hasHit() ? h.dist : FLT_MAX; }
}

} // extern "C"
