// shading.h — device-side scene vocabulary of the path tracer: sampler, sensor, differential geometry,
// textures, BSDFs, emitters.  Each function states the reference lines whose behaviour it reproduces.
#pragma once
#include "device_scene.h"
#include "mipmap.h"

// Feature set compiled into the shading functions of this translation unit (bits = kShade* of device_scene.h).  The shade
// kernel is built twice (shade_basic.hip: 0, shade_full.hip: all); everything else uses the full set.
#ifndef CTL_SHADE_FEATURES
#define CTL_SHADE_FEATURES 0x7F
#endif
// BSDF models compiled into the dispatch switches of this translation unit: bit m = the model CTL_BSDF_* == m (bit 0 = the miss / environment branch of the shade kernel).
// The model-class builds of the shade kernel (shade_class_*.hip) each carry one group of models; everything else carries all of them.
#ifndef CTL_SHADE_MODELS
#define CTL_SHADE_MODELS 0xFFFFu
#endif
#define CTL_HAS_MODEL(m) ((((unsigned)(CTL_SHADE_MODELS)) >> (m)) & 1u)
// the rough and nesting model bodies stay out of line where a nesting model can call back into them (one copy instead of one per call site); a class build without nesting inlines them
// emitter functions that take the scene by reference: out of line they force a private copy of the kernel's dev_scene argument (496 B of scratch per lane); CTL_LIGHT_INLINE builds inline them
#ifdef CTL_LIGHT_INLINE
#define CTL_LIGHT_OUTLINE __forceinline__
#define CTL_LIGHT_MAYBE __forceinline__
#else
#define CTL_LIGHT_OUTLINE __noinline__
#define CTL_LIGHT_MAYBE
#endif
// the bodies of the rough models (called from sample, f and pdf): CTL_ROUGH_INLINE builds force them inline, so that no call takes the BSDF record by reference and it can live in registers
#ifdef CTL_ROUGH_INLINE
#define CTL_ROUGH_BODY __forceinline__
#else
#define CTL_ROUGH_BODY
#endif
#if (CTL_SHADE_FEATURES & 16) || defined(CTL_CLASS_B_OUTLINE)
#define CTL_ROUGH_OUTLINE __noinline__
#else
#define CTL_ROUGH_OUTLINE
#endif

namespace ctl {

// ---- SequenceSampler (Kernel/Sampler_device.h:59-113): u = frac(T[d % 30][i % 4096] + T[d % 30][(i / 4096) % 4096])
struct sampler {
    const float* __restrict__ t1; const float2* __restrict__ t2; uint32_t idx, d1, d2;
    __device__ __forceinline__ float next1() {
        const uint32_t e = (d1 % CTL_SAMPLER_SEQUENCE_LENGTH) * CTL_SAMPLER_NUM_SEQUENCES;
        float v = t1[e + (idx % CTL_SAMPLER_NUM_SEQUENCES)]; v += t1[e + ((idx / CTL_SAMPLER_NUM_SEQUENCES) % CTL_SAMPLER_NUM_SEQUENCES)];
        d1++; return fracf(v);
    }
    __device__ __forceinline__ f2 next2() {
        const uint32_t e = (d2 % CTL_SAMPLER_SEQUENCE_LENGTH) * CTL_SAMPLER_NUM_SEQUENCES;
        const float2 a = t2[e + (idx % CTL_SAMPLER_NUM_SEQUENCES)], b = t2[e + ((idx / CTL_SAMPLER_NUM_SEQUENCES) % CTL_SAMPLER_NUM_SEQUENCES)];
        float x = a.x; x += b.x; float y = a.y; y += b.y;
        d2++; return f2{ fracf(x), fracf(y) };
    }
};

// ---- sampleRay / sampleRayDifferential of the four projective sensors (SceneTypes/Sensor.cu: perspective :116-144, thin lens :267-311,
// orthographic :429-450, telecentric :537-574)
__device__ __forceinline__ f3 sensor_near_point(const dev_sensor& c, f2 pixelSample) {   // m_sampleToCamera.TransformPoint(pixelSample * invResolution, 0)
    const float px = pixelSample.x * c.inv_res[0], py = pixelSample.y * c.inv_res[1];
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { float s = c.s2c[i * 4] * px; s += c.s2c[i * 4 + 1] * py; s += c.s2c[i * 4 + 2] * 0.0f; s += c.s2c[i * 4 + 3] * 1.0f; r[i] = s; }
    return f3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}
__device__ __forceinline__ m34 sensor_to_world(const dev_sensor& c) {
    m34 m;
#pragma unroll
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) m.r[i][j] = c.to_world[i * 4 + j];
    return m;
}
__device__ __forceinline__ void sensor_sample_ray(const dev_sensor& c, f2 pixelSample, f2 apertureSample, f3& o, f3& d) {
    const m34 m = sensor_to_world(c);
    if (c.type == CTL_SENSOR_SPHERICAL) {   // SphericalSensor::sampleRay (Sensor.cu:6-17)
        float sinPhi, cosPhi, sinTheta, cosTheta;
        m_sincos((1.0f - pixelSample.x * c.inv_res[0]) * 2 * kPi, &sinPhi, &cosPhi);
        m_sincos((1.0f - pixelSample.y * c.inv_res[1]) * kPi, &sinTheta, &cosTheta);
        o = xform_point(m, f3(0.0f)); d = xform_dir(m, f3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta));
        return;
    }
    const f3 nearP = sensor_near_point(c, pixelSample);
    if (c.type == CTL_SENSOR_PERSPECTIVE) {
        o = xform_point(m, f3(0.0f));   // toWorld.Translation()
        d = xform_dir(m, normalize(nearP));
    } else if (c.type == CTL_SENSOR_THINLENS) {
        const f2 tmp = square_to_disk_concentric(apertureSample);
        const f3 apertureP(tmp.x * c.aperture_radius, tmp.y * c.aperture_radius, 0.0f);
        const f3 focusP = nearP * (c.focus_distance / nearP.z);
        o = xform_point(m, apertureP); d = xform_dir(m, normalize(focusP - apertureP));
    } else if (c.type == CTL_SENSOR_ORTHOGRAPHIC) {
        o = xform_point(m, f3(nearP.x, nearP.y, 0.0f)); d = xform_dir(m, f3(0.0f, 0.0f, 1.0f));   // toWorld.Forward()
    } else {
        const f2 q = square_to_disk_concentric(apertureSample); const float sc = c.aperture_radius / c.screen_scale_x;
        f3 focusP = nearP; focusP.z = c.focus_distance;
        const f3 orig(q.x * sc + focusP.x, q.y * sc + focusP.y, 0.0f);
        o = xform_point(m, orig); d = normalize(xform_dir(m, focusP - orig));
    }
}
// the ray and its x / y neighbours (perspective sensors shift the direction, orthographic ones the origin)
__device__ __forceinline__ void sensor_sample_ray_differential(const dev_sensor& c, f2 pixelSample, f2 apertureSample, f3& o, f3& d, f3& oX, f3& dX, f3& oY, f3& dY) {
    if (c.type == CTL_SENSOR_SPHERICAL) {   // SphericalSensor::sampleRayDifferential (Sensor.h:122-125) sets the ray only and leaves rayX / rayY as they were: the ray itself is used
        sensor_sample_ray(c, pixelSample, apertureSample, o, d); oX = oY = o; dX = dY = d; return;
    }
    const f3 nearP = sensor_near_point(c, pixelSample);
    const m34 m = sensor_to_world(c);
    const f3 ddx(c.dx[0], c.dx[1], c.dx[2]), ddy(c.dy[0], c.dy[1], c.dy[2]);
    if (c.type == CTL_SENSOR_PERSPECTIVE) {
        o = xform_point(m, f3(0.0f));
        d = xform_dir(m, normalize(nearP));
        oX = oY = o; dX = xform_dir(m, normalize(nearP + ddx)); dY = xform_dir(m, normalize(nearP + ddy));
    } else if (c.type == CTL_SENSOR_THINLENS) {
        const f2 tmp = square_to_disk_concentric(apertureSample);
        const f3 apertureP(tmp.x * c.aperture_radius, tmp.y * c.aperture_radius, 0.0f);
        const float fDist = c.focus_distance / nearP.z;
        const f3 focusP = nearP * fDist, focusPx = (nearP + ddx) * fDist, focusPy = (nearP + ddy) * fDist;
        o = xform_point(m, apertureP); d = xform_dir(m, normalize(focusP - apertureP));
        oX = oY = o; dX = xform_dir(m, normalize(focusPx - apertureP)); dY = xform_dir(m, normalize(focusPy - apertureP));
    } else if (c.type == CTL_SENSOR_ORTHOGRAPHIC) {
        o = xform_point(m, nearP); d = xform_dir(m, f3(0.0f, 0.0f, 1.0f));
        oX = xform_point(m, nearP + ddx); oY = xform_point(m, nearP + ddy); dX = dY = d;
    } else {
        const f2 q = square_to_disk_concentric(apertureSample); const float sc = c.aperture_radius / c.screen_scale_x;
        f3 focusP = nearP; focusP.z = c.focus_distance;
        const f3 orig(q.x * sc + focusP.x, q.y * sc + focusP.y, 0.0f);
        o = xform_point(m, orig); d = normalize(xform_dir(m, focusP - orig));
        oX = xform_point(m, orig + ddx); oY = xform_point(m, orig + ddy); dX = dY = d;
    }
}

// ---- differential geometry at a hit (Kernel/TraceHelper.cu:274-307 -> Engine/TriangleData.cu:75-103)
struct diff_geom {
    f3 P; frame sys; f3 n; f2 uv; const ctl_mipmap* images; const ctl_rough_transmittance* rough_transmittance; const ctl_material* mats;
    const float* rt_reduced;   // dev_scene::rt_reduced
#if (CTL_SHADE_FEATURES & 32) || defined(CTL_TEX_PARTIALS)
    f3 dpdu, dpdv;   // world space, for height maps and ray differentials
#endif
#ifdef CTL_TEX_PARTIALS
    // ray differentials of the first hit (DifferentialGeometry::computePartials; megakernel PathTracer only, PathTracer.cu:60-61)
    bool has_uv_partials; float dudx, dudy, dvdx, dvdy; const dev_mip_levels* mip_levels; const float* mip_weight_lut;
#endif
};   // images: g_SceneData.m_sTexData; tables of RoughTransmittanceManager (both uniform)
// The scene's small tables — emitter records, the anim blob (area lights' triangle CDFs and shape triangles, the environment map's row weights), the normal codec's sin / cos
// table — are read through these.  A shade kernel built with CTL_SHADE_LDS_TABLES = K keeps them in LDS when records + blob fit K KB (staged once per workgroup by
// scene_tables_to_lds at kernel start); next-event estimation walks them one after the other (sampled emitter -> its record -> its CDF -> the chosen triangle), each step a
// dependent load.  The struct itself is never copied: a private copy of dev_scene, whose light_cdf / light_indices are indexed per lane, would live in scratch.
#ifndef CTL_SHADE_LDS_TABLES
#define CTL_SHADE_LDS_TABLES 0
#endif
#if CTL_SHADE_LDS_TABLES
constexpr uint32_t kLdsTabBytes = CTL_SHADE_LDS_TABLES * 1024u;
__device__ __forceinline__ unsigned char* scene_lds_tab() { __shared__ __attribute__((aligned(16))) unsigned char tab[kLdsTabBytes + 4096u]; return tab; }
__device__ __forceinline__ uint32_t scene_lds_light_bytes(const dev_scene& S) { return (S.n_lights_buf * (uint32_t)sizeof(ctl_light) + 15u) & ~15u; }
__device__ __forceinline__ bool scene_lds_fits(const dev_scene& S) { return scene_lds_light_bytes(S) + ((S.n_anim_bytes + 15u) & ~15u) <= kLdsTabBytes; }
__device__ __forceinline__ const ctl_light* scene_lights(const dev_scene& S) { return scene_lds_fits(S) ? (const ctl_light*)scene_lds_tab() : S.lights; }
__device__ __forceinline__ const unsigned char* scene_anim(const dev_scene& S) { return scene_lds_fits(S) ? scene_lds_tab() + scene_lds_light_bytes(S) : S.anim; }
__device__ __forceinline__ const float2* scene_normal_lut(const dev_scene&) { return (const float2*)(scene_lds_tab() + kLdsTabBytes); }
template <uint32_t BLOCK> __device__ __forceinline__ void scene_tables_to_lds(const dev_scene& S) {
    unsigned char* tab = scene_lds_tab();
    for (uint32_t k = threadIdx.x; k < 4096u / 16u; k += BLOCK) ((uint4*)(tab + kLdsTabBytes))[k] = ((const uint4*)S.normal_lut)[k];
    if (scene_lds_fits(S)) {
        const uint32_t lw = S.n_lights_buf * (uint32_t)(sizeof(ctl_light) / 4u), lb = scene_lds_light_bytes(S), aw = (S.n_anim_bytes + 3u) / 4u;   // (device allocations are padded: the blob's last word may reach 3 bytes past it)
        for (uint32_t k = threadIdx.x; k < lw; k += BLOCK) ((uint32_t*)tab)[k] = ((const uint32_t*)S.lights)[k];
        for (uint32_t k = threadIdx.x; k < aw; k += BLOCK) ((uint32_t*)(tab + lb))[k] = ((const uint32_t*)S.anim)[k];
    }
    __syncthreads();
}
#else
__device__ __forceinline__ const ctl_light* scene_lights(const dev_scene& S) { return S.lights; }
__device__ __forceinline__ const unsigned char* scene_anim(const dev_scene& S) { return S.anim; }
__device__ __forceinline__ const float2* scene_normal_lut(const dev_scene& S) { return S.normal_lut; }
#endif
__device__ __forceinline__ void fill_dg(const dev_scene& S, float u, float v, int tri, int node, diff_geom& dg) {
    const uint4 ta = S.tri_data[tri * 2], tb = S.tri_data[tri * 2 + 1];   // {nme.x, nme.y, dpd.x, dpd.y} {dpd.z, uv0, uv1, uv2}
    dg.images = S.images; dg.rough_transmittance = S.rough_transmittance; dg.mats = S.mats; dg.rt_reduced = S.rt_reduced;
    const float4 f0 = S.inst_fwd[node * 3], f1 = S.inst_fwd[node * 3 + 1], f2_ = S.inst_fwd[node * 3 + 2];
    m34 l2w; l2w.r[0][0] = f0.x; l2w.r[0][1] = f0.y; l2w.r[0][2] = f0.z; l2w.r[0][3] = f0.w; l2w.r[1][0] = f1.x; l2w.r[1][1] = f1.y; l2w.r[1][2] = f1.z; l2w.r[1][3] = f1.w;
    l2w.r[2][0] = f2_.x; l2w.r[2][1] = f2_.y; l2w.r[2][2] = f2_.z; l2w.r[2][3] = f2_.w;
    const f3 na = uchar2_to_normal_lut(ta.x & 0xffff, scene_normal_lut(S)), nb = uchar2_to_normal_lut(ta.x >> 16, scene_normal_lut(S)), nc = uchar2_to_normal_lut(ta.y & 0xffff, scene_normal_lut(S));   // = uchar2_to_normal, from the scene's 4-KB table
    const float w = 1.0f - u - v;
    const f3 n = normalize(u * na + v * nb + w * nc);
    const f3 dpdu(half_to_float((uint16_t)ta.z), half_to_float((uint16_t)(ta.z >> 16)), half_to_float((uint16_t)ta.w));
    const f3 dpdv(half_to_float((uint16_t)(ta.w >> 16)), half_to_float((uint16_t)tb.x), half_to_float((uint16_t)(tb.x >> 16)));
    f3 s = dpdu - n * dot(n, dpdu);
    f3 t = cross(s, n);
    s = xform_dir(l2w, s); t = xform_dir(l2w, t);
    dg.sys.s = normalize(s); dg.sys.t = normalize(t); dg.sys.n = normalize(cross(t, s));
    const f3 wdpdu = xform_dir(l2w, dpdu), wdpdv = xform_dir(l2w, dpdv);
    dg.n = normalize(cross(wdpdu, wdpdv));
#if (CTL_SHADE_FEATURES & 32) || defined(CTL_TEX_PARTIALS)
    dg.dpdu = wdpdu; dg.dpdv = wdpdv;
#endif
#ifdef CTL_TEX_PARTIALS
    dg.has_uv_partials = false; dg.mip_levels = S.mip_levels; dg.mip_weight_lut = S.mip_weight_lut;   // TraceResult::fillDG (Kernel/TraceResult.cu:13-21)
#endif
    const f2 uva{ half_to_float((uint16_t)tb.y), half_to_float((uint16_t)(tb.y >> 16)) }, uvb{ half_to_float((uint16_t)tb.z), half_to_float((uint16_t)(tb.z >> 16)) },
        uvc{ half_to_float((uint16_t)tb.w), half_to_float((uint16_t)(tb.w >> 16)) };
    dg.uv = f2{ u * uva.x + v * uvb.x + w * uvc.x, u * uva.y + v * uvb.y + w * uvc.y };
    if (dot(dg.n, dg.sys.n) < 0.0f) dg.n = -dg.n;
}
#ifdef CTL_TEX_PARTIALS
// DifferentialGeometry::computePartials (Engine/DifferentialGeometry.cu:9-90); rox / roy: origins of the x / y differential rays
__device__ inline void compute_partials(diff_geom& dg, f3 rox, f3 rxd, f3 roy, f3 ryd) {
    dg.has_uv_partials = true;
    if (dot(dg.dpdu, dg.dpdu) == 0 && dot(dg.dpdv, dg.dpdv) == 0) { dg.dudx = dg.dvdx = dg.dudy = dg.dvdy = 0.0f; return; }
    const float pp = dot(dg.n, dg.P), pox = dot(dg.n, rox), poy = dot(dg.n, roy), prx = dot(dg.n, rxd), pry = dot(dg.n, ryd);
    if (prx == 0 || pry == 0) { dg.dudx = dg.dvdx = dg.dudy = dg.dvdy = 0.0f; return; }
    const float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
    const float absX = fabsf(dg.n.x), absY = fabsf(dg.n.y), absZ = fabsf(dg.n.z);
    const int a0 = (absX > absY && absX > absZ) ? 1 : 0, a1 = (absX > absY && absX > absZ) ? 2 : (absY > absZ ? 2 : 1);
    auto comp = [](f3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); };
    const float A00 = comp(dg.dpdu, a0), A01 = comp(dg.dpdv, a0), A10 = comp(dg.dpdu, a1), A11 = comp(dg.dpdv, a1);
    const f3 px = rox + rxd * tx, py = roy + ryd * ty;
    const float Bx0 = comp(px, a0) - comp(dg.P, a0), Bx1 = comp(px, a1) - comp(dg.P, a1), By0 = comp(py, a0) - comp(dg.P, a0), By1 = comp(py, a1) - comp(dg.P, a1);
    const float det = A00 * A11 - A01 * A10;   // AlgebraHelper::solveLinearSystem2x2 (Math/AlgebraHelper.h:11-24)
    if (fabsf(det) <= 2.93873587705571876e-39f) { dg.dudx = 1; dg.dvdx = 0; dg.dudy = 0; dg.dvdy = 1; return; }
    const float inverse = 1.0f / det;
    dg.dudx = (A11 * Bx0 - A01 * Bx1) * inverse; dg.dvdx = (A00 * Bx1 - A10 * Bx0) * inverse;
    dg.dudy = (A11 * By0 - A01 * By1) * inverse; dg.dvdy = (A00 * By1 - A10 * By0) * inverse;
}
#endif
__device__ __forceinline__ uint32_t tri_mat_index(const dev_scene& S, int tri) { return (S.tri_data[tri * 2].y >> 16) & 0xff; }   // TriangleData.h:40-44

// ImageTexture::Evaluate(uv) (Texture.cu:6-13); out of line: tex_eval is inlined at every BSDF parameter fetch and the bitmap
// path must not cost the constant-texture path registers
__device__ __noinline__ f3 tex_eval_image(const ctl_texture& t, f2 duv, const ctl_mipmap* images) {
    if (t.image == 0xffffffffu) return f3(0.0f);
    const f2 uv{ t.uv_scale[0] * duv.x + 0 * duv.y + t.uv_offset[0], 0 * duv.x + t.uv_scale[1] * duv.y + t.uv_offset[1] };
    const ctl_mipmap& M = images[t.image];
    return (M.filter_mode == CTL_FILTER_POINT ? mip_texel(M, uv) : mip_triangle(M, uv)) * f3(t.value[0], t.value[1], t.value[2]);
}

// ---- textures (SceneTypes/Texture.h:107-183, Texture.cu:6-29)
__device__ __forceinline__ f3 tex_eval(const ctl_texture& t, const diff_geom& dg) {
    if (t.type == CTL_TEX_CHECKER) {
        const float u = dg.uv.x * t.uv_scale[0] + t.uv_offset[0], v = dg.uv.y * t.uv_scale[1] + t.uv_offset[1];
        int xm = (int)(u * 2) % 2, ym = (int)(v * 2) % 2; if (xm < 0) xm += 2; if (ym < 0) ym += 2;
        const int x = 2 * xm - 1, y = 2 * ym - 1;
        return (x * y == 1) ? f3(t.value[0], t.value[1], t.value[2]) : f3(t.value1[0], t.value1[1], t.value1[2]);
    }
#if CTL_SHADE_FEATURES & 4
#ifdef CTL_TEX_PARTIALS
    if (t.type == CTL_TEX_IMAGE && dg.has_uv_partials && t.image != 0xffffffffu) {   // ImageTexture::Evaluate(dg) with uv partials (Texture.cu:15-29): differentiate (Texture.h:52-59, m12 = m21 = 0) -> KernelMIPMap::eval
        const f2 uv{ t.uv_scale[0] * dg.uv.x + 0 * dg.uv.y + t.uv_offset[0], 0 * dg.uv.x + t.uv_scale[1] * dg.uv.y + t.uv_offset[1] };
        const float dsdx = t.uv_scale[0] * dg.dudx + 0 * dg.dvdx, dsdy = t.uv_scale[0] * dg.dudy + 0 * dg.dvdy;
        const float dtdx = 0 * dg.dudx + t.uv_scale[1] * dg.dvdx, dtdy = 0 * dg.dudy + t.uv_scale[1] * dg.dvdy;
        return mip_eval(dg.images[t.image], dg.mip_levels[t.image], dg.mip_weight_lut, uv, f2{ dsdx, dtdx }, f2{ dsdy, dtdy }) * f3(t.value[0], t.value[1], t.value[2]);
    }
#endif
    if (t.type == CTL_TEX_IMAGE) return tex_eval_image(t, dg.uv, dg.images);
#endif
    return f3(t.value[0], t.value[1], t.value[2]);
}
#if CTL_SHADE_FEATURES & 32
// Material::SampleNormalMap (Engine/Material.cu:96-138; parallax occlusion is never enabled by the reference)
__device__ __forceinline__ void sample_normal_map(const ctl_material& mat, diff_geom& dg) {
    if (mat.map_kind == CTL_MAP_NORMAL) {
        const f3 c = tex_eval(mat.map_tex, dg);
        const f3 nWorld = normalize(dg.sys.to_world(c - f3(0.5f)));
        dg.sys.n = nWorld;
        dg.sys.t = normalize(cross(nWorld, dg.sys.s));
        dg.sys.s = normalize(cross(nWorld, dg.sys.t));
    } else if (mat.map_kind == CTL_MAP_HEIGHT && mat.map_tex.type == CTL_TEX_IMAGE && mat.map_tex.image != 0xffffffffu) {
        f3 g0, g1;
        mip_eval_gradient(dg.images[mat.map_tex.image], tex_map_point(mat.map_tex, dg.uv), g0, g1);
        const float dDispDu = luminance(g0), dDispDv = luminance(g1);
        const f3 dpdu = dg.dpdu + dg.sys.n * (dDispDu - dot(dg.sys.n, dg.dpdu));
        const f3 dpdv = dg.dpdv + dg.sys.n * (dDispDv - dot(dg.sys.n, dg.dpdv));
        dg.sys.n = normalize(cross(dpdu, dpdv));
        dg.sys.s = normalize(dpdu - dg.sys.n * dot(dg.sys.n, dpdu));
        dg.sys.t = normalize(cross(dg.sys.n, dg.sys.s));
        if (dot(dg.sys.n, dg.n) < 0) dg.sys.n = -dg.sys.n;
    }
}
#endif
__device__ __forceinline__ float avg3(f3 s) { float r = s.x; r += s.y; r += s.z; return r * (1.0f / 3); }   // Spectrum.h:180-190

#if CTL_SHADE_FEATURES & 64
// math::erfinv / math::erf (Math/MathFunc.h:343-393)
__device__ __forceinline__ float erfinv_ref(float x) {
    float w = -m_log((1.0f - x) * (1.0f + x)), p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w; p = -4.39150654e-06f + p * w; p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w; p = -0.00417768164f + p * w; p = 0.246640727f + p * w; p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3;
        p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w; p = -0.00367342844f + p * w; p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w; p = 0.00943887047f + p * w; p = 1.00167406f + p * w; p = 2.83297682f + p * w;
    }
    return p * x;
}
__device__ __forceinline__ float erf_ref(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    const float sign = copysign_bits(1.0f, x);
    x = fabsf(x);
    const float t = 1.0f / (1.0f + p * x);
    return sign * (1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * m_exp(-x * x));
}
#endif

// ---- microfacet distribution (Engine/MicrofacetDistribution.{h,cu}).  Beckmann (full-distribution sampling) and GGX are always built;
// visible-normal sampling of Beckmann and the Phong distribution come with feature bit 64 (kShadeMoreMicrofacet)
struct microfacet {
    int type; float aU, aV; bool vis;
#if CTL_SHADE_FEATURES & 64
    float eU = 0, eV = 0;   // computePhongExponent (MicrofacetDistribution.h)
    __device__ __forceinline__ microfacet(int t, float u, float v, bool sv) : type(t), aU(max2(u, 1e-4f)), aV(max2(v, 1e-4f)), vis(sv) {
        if (t == CTL_MF_PHONG) { eU = max2(2.0f / (aU * aU) - 2.0f, 0.0f); eV = max2(2.0f / (aV * aV) - 2.0f, 0.0f); }
    }
    __device__ float interp_phong_exp(f3 v) const {
        const float s2 = sin_theta2(v);
        if (iso() || s2 <= 2.93873587705571876e-39f) return eU;
        const float is2 = 1 / s2;
        return eU * (v.x * v.x * is2) + eV * (v.y * v.y * is2);
    }
    __device__ void sample_first_quadrant(float u1, float& phi, float& exponent) const {   // MicrofacetDistribution.h:161-170
        phi = m_atan(sqrtf((eU + 2.0f) / (eV + 2.0f)) * m_tan(kPi * u1 * 0.5f));
        float sp, cp; m_sincos(phi, &sp, &cp);
        exponent = eU * cp * cp + eV * sp * sp;
    }
#else
    __device__ __forceinline__ microfacet(int t, float u, float v, bool sv) : type(t), aU(max2(u, 1e-4f)), aV(max2(v, 1e-4f)), vis(sv) {}
#endif
    __device__ __forceinline__ void scale_alpha(float v) {   // MicrofacetDistribution.h:61-67
        aU *= v; aV *= v;
#if CTL_SHADE_FEATURES & 64
        if (type == CTL_MF_PHONG) { eU = max2(2.0f / (aU * aU) - 2.0f, 0.0f); eV = max2(2.0f / (aV * aV) - 2.0f, 0.0f); }
#endif
    }
    __device__ __forceinline__ bool iso() const { return aU == aV; }
    __device__ float eval(f3 m) const {   // MicrofacetDistribution.cu:6-42
        if (cos_theta(m) <= 0) return 0.0f;
        const float c2 = m.z * m.z;
        const float be = ((m.x * m.x) / (aU * aU) + (m.y * m.y) / (aV * aV)) / c2;
        float result;
        if (type == CTL_MF_BECKMANN) result = m_exp(-be) / (kPi * aU * aV * c2 * c2);
#if CTL_SHADE_FEATURES & 64
        else if (type == CTL_MF_PHONG) result = sqrtf((eU + 2) * (eV + 2)) * kInvTwoPi * m_pow(cos_theta(m), interp_phong_exp(m));
#endif
        else { const float root = (1 + be) * c2; result = 1.0f / (kPi * aU * aV * root * root); }
        if (result < 1e-20f) result = 0;
        return result;
    }
    __device__ float project_roughness(f3 v) const {
        const float is2 = 1 / sin_theta2(v);
        if (iso() || is2 <= 0) return aU;
        const float cp2 = v.x * v.x * is2, sp2 = v.y * v.y * is2;
        return sqrtf(cp2 * aU * aU + sp2 * aV * aV);
    }
    __device__ float smith_g1(f3 v, f3 m) const {   // MicrofacetDistribution.cu:309-343
        if (dot(v, m) * cos_theta(v) <= 0) return 0.0f;
        const float tt = fabsf(tan_theta(v));
        if (tt == 0.0f) return 1.0f;
        const float alpha = project_roughness(v);
        if (type == CTL_MF_GGX) {   // 2 / (1 + hypot2(1, alpha tan)) with math::hypot2's scaling (Math/MathFunc.h:326-341)
            const float root = alpha * tt; float hyp;
            if (1.0f > fabsf(root)) { const float r = root / 1.0f; hyp = 1.0f * sqrtf(1.0f + r * r); }
            else if (root != 0.0f) { const float r = 1.0f / root; hyp = fabsf(root) * sqrtf(1.0f + r * r); }
            else hyp = 0.0f;
            return 2.0f / (1.0f + hyp);
        }
        const float a = 1.0f / (alpha * tt);
        if (a >= 1.6f) return 1.0f;
        const float a2 = a * a;
        return (3.535f * a + 2.181f * a2) / (1.0f + 2.276f * a + 2.577f * a2);
    }
    __device__ float G(f3 wi, f3 wo, f3 m) const { return smith_g1(wi, m) * smith_g1(wo, m); }
    __device__ float pdf_visible(f3 wi, f3 m) const { if (cos_theta(wi) == 0) return 0.0f; return smith_g1(wi, m) * absdot(wi, m) * eval(m) / fabsf(cos_theta(wi)); }
    __device__ float pdf(f3 wi, f3 m) const { return vis ? pdf_visible(wi, m) : eval(m) * cos_theta(m); }
    __device__ f3 sample_all(f2 s, float& pdf_) const {   // MicrofacetDistribution.cu:44-149
        float cosThetaM, sinPhiM, cosPhiM, alphaSqr;
#if CTL_SHADE_FEATURES & 64
        if (type == CTL_MF_PHONG) {   // :108-137
            float phiM, exponent;
            if (iso()) { phiM = (2.0f * kPi) * s.y; exponent = eU; }
            else if (s.y < 0.25f) sample_first_quadrant(4 * s.y, phiM, exponent);
            else if (s.y < 0.5f) { sample_first_quadrant(4 * (0.5f - s.y), phiM, exponent); phiM = kPi - phiM; }
            else if (s.y < 0.75f) { sample_first_quadrant(4 * (s.y - 0.5f), phiM, exponent); phiM += kPi; }
            else { sample_first_quadrant(4 * (1 - s.y), phiM, exponent); phiM = 2 * kPi - phiM; }
            m_sincos(phiM, &sinPhiM, &cosPhiM);
            cosThetaM = m_pow(s.x, 1.0f / (exponent + 2.0f));
            pdf_ = sqrtf((eU + 2.0f) * (eV + 2.0f)) * kInvTwoPi * m_pow(cosThetaM, exponent + 1.0f);
            if (pdf_ < 1e-20f) pdf_ = 0;
            const float sinThetaP = sqrtf(max2(0.0f, 1 - cosThetaM * cosThetaM));
            return f3(sinThetaP * cosPhiM, sinThetaP * sinPhiM, cosThetaM);
        }
#endif
        if (iso()) { m_sincos((2.0f * kPi) * s.y, &sinPhiM, &cosPhiM); alphaSqr = aU * aU; }
        else {
            const float phiM = m_atan(aV / aU * m_tan(kPi + 2 * kPi * s.y)) + kPi * floorf(2 * s.y + 0.5f);
            m_sincos(phiM, &sinPhiM, &cosPhiM);
            const float cs = cosPhiM / aU, ss = sinPhiM / aV;
            alphaSqr = 1.0f / (cs * cs + ss * ss);
        }
        if (type == CTL_MF_BECKMANN) {
            const float t2 = alphaSqr * -m_log(1.0f - s.x);
            cosThetaM = 1.0f / sqrtf(1.0f + t2);
            pdf_ = (1.0f - s.x) / (kPi * aU * aV * cosThetaM * cosThetaM * cosThetaM);
        } else {
            const float t2 = alphaSqr * s.x / (1.0f - s.x);
            cosThetaM = 1.0f / sqrtf(1.0f + t2);
            const float temp = 1 + t2 / alphaSqr;
            pdf_ = kInvPi / (aU * aV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
        if (pdf_ < 1e-20f) pdf_ = 0;
        const float sinThetaM = sqrtf(max2(0.0f, 1 - cosThetaM * cosThetaM));
        return f3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }
    __device__ f2 sample_visible11(float thetaI, f2 s) const {   // MicrofacetDistribution.cu:185-307
#if CTL_SHADE_FEATURES & 64
        if (type == CTL_MF_BECKMANN) {   // :191-256: Newton / bisection on the CDF in the erf domain
            const float SQRT_PI_INV = 1 / sqrtf(kPi);
            if (thetaI < 1e-4f) { const float r = sqrtf(-m_log(1.0f - s.x)); float sp, cp; m_sincos(2 * kPi * s.y, &sp, &cp); return f2{ r * cp, r * sp }; }
            const float tanThetaI = m_tan(thetaI), cotThetaI = 1 / tanThetaI;
            float a = -1, c = erf_ref(cotThetaI);
            const float sample_x = max2(s.x, 1e-6f);
            const float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            float b = c - (1 + c) * m_pow(1 - sample_x, fit);
            const float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * m_exp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = 0.5f * (a + c);
                const float invErf = erfinv_ref(b);
                const float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * m_exp(-invErf * invErf)) - sample_x;
                const float derivative = normalization * (1 - invErf * tanThetaI);
                if (fabsf(value) < 1e-5f) break;
                if (value > 0) c = b; else a = b;
                b -= value / derivative;
            }
            return f2{ erfinv_ref(b), erfinv_ref(2.0f * max2(s.y, 1e-6f) - 1.0f) };
        }
#endif
        if (thetaI < 1e-4f) { const float r = safe_sqrt(s.x / (1 - s.x)); float sp, cp; m_sincos(2 * kPi * s.y, &sp, &cp); return f2{ r * cp, r * sp }; }
        const float tanThetaI = m_tan(thetaI), a = 1 / tanThetaI;
        const float G1 = 2.0f / (1.0f + safe_sqrt(1.0f + 1.0f / (a * a)));
        float A = 2.0f * s.x / G1 - 1.0f;
        if (fabsf(A) == 1) A -= copysign_bits(1.0f, A) * 1e-7f;
        const float tmp = 1.0f / (A * A - 1.0f), B = tanThetaI;
        const float D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
        const float sx1 = B * tmp - D, sx2 = B * tmp + D;
        f2 slope; slope.x = (A < 0.0f || sx2 > 1.0f / tanThetaI) ? sx1 : sx2;
        float Sg;
        if (s.y > 0.5f) { Sg = 1.0f; s.y = 2.0f * (s.y - 0.5f); } else { Sg = -1.0f; s.y = 2.0f * (0.5f - s.y); }
        const float z = (s.y * (s.y * (s.y * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                        (s.y * (s.y * (s.y * (s.y * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
        slope.y = Sg * z * sqrtf(1.0f + slope.x * slope.x);
        return slope;
    }
    __device__ f3 sample_visible(f3 _wi, f2 s) const {   // MicrofacetDistribution.cu:151-183
        const f3 wi = normalize(f3(aU * _wi.x, aV * _wi.y, _wi.z));
        float theta = 0, phi = 0;
        if (wi.z < 0.99999f) { theta = m_acos(wi.z); phi = m_atan2(wi.y, wi.x); }
        float sp, cp; m_sincos(phi, &sp, &cp);
        f2 slope = sample_visible11(theta, s);
        slope = f2{ cp * slope.x - sp * slope.y, sp * slope.x + cp * slope.y };
        slope.x *= aU; slope.y *= aV;
        const float nrm = 1.0f / sqrtf(slope.x * slope.x + slope.y * slope.y + 1.0f);
        return f3(-slope.x * nrm, -slope.y * nrm, nrm);
    }
    __device__ f3 sample(f3 wi, f2 s, float& pdf_) const { if (vis) { const f3 m = sample_visible(wi, s); pdf_ = pdf_visible(wi, m); return m; } return sample_all(s, pdf_); }
};
__device__ __forceinline__ f3 reflect_about(f3 wi, f3 n) { return normalize(2 * dot(wi, n) * n - wi); }   // FresnelHelper.h:148-151

// ---- BSDFs (SceneTypes/BSDF_Simple.cu) in the local shading frame
enum { kESmooth = 0x2 | 0x4 | 0x8 | 0x10, kEDelta = 0x1 | 0x20 | 0x40, kEAll = 0x1ff };
struct bsdf_rec {
    diff_geom dg; f3 wi, wo; float eta; uint32_t type_mask, sampled_type;
#if CTL_SHADE_FEATURES & 2
    // RoughTransmittanceManager lookups for the incident direction, memoised per vertex: sample, f and pdf of a rough plastic (and the NEE
    // evaluation that follows) ask for T(cos wi, alpha, eta) five times and for the diffuse table twice; each is a 64-tap spline lookup.
    // Keyed by the arguments, so a nested BSDF evaluated with another wi or another table simply misses.
    mutable float rt_cos = -2.0f, rt_alpha = 0, rt_eta = 0, rt_val = 0, rtd_alpha = -1.0f, rtd_eta = 0, rtd_val = 0; mutable uint32_t rt_type = 0, rtd_type = 0;
#endif
};

} // namespace ctl
#include "bsdf_more.h"
namespace ctl {

__device__ f3 bsdf_sample(const ctl_material& M, bsdf_rec& b, float& pdf, f2 smp) {
    switch (M.bsdf_type) {
    case CTL_BSDF_DIFFUSE: { if (!CTL_HAS_MODEL(CTL_BSDF_DIFFUSE)) return f3(0.0f);   // BSDF_Simple.cu:7-36
        const uint32_t ct = M.combined_type;
        if (!(b.type_mask & ct) || (ct == CTL_EDiffuseReflection && cos_theta(b.wi) <= 0)) return f3(0.0f);
        b.sampled_type = ct;
        float sc = 1;
        if (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission)) {
            b.sampled_type = smp.x < 0.5f ? CTL_EDiffuseReflection : CTL_EDiffuseTransmission;
            smp.x = smp.x < 0.5f ? smp.x * 2 : (smp.x - 0.5f) * 2;
            sc = 0.5f;
        }
        b.wo = square_to_cosine_hemisphere(smp);
        if ((ct == CTL_EDiffuseTransmission || (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission) && b.sampled_type == CTL_EDiffuseTransmission)) && cos_theta(b.wi) > 0) b.wo.z *= -1;
        b.eta = 1.0f;
        pdf = fabsf(kInvPi * cos_theta(b.wo)) * sc;
        return tex_eval(M.tex[0], b.dg) * sc;
    }
    case CTL_BSDF_DIELECTRIC: { if (!CTL_HAS_MODEL(CTL_BSDF_DIELECTRIC)) return f3(0.0f);   // BSDF_Simple.cu:174-224; no dispersion -> eta = B + C / 0.6 (Dispersion.h)
        const bool sr = (b.type_mask & CTL_EDeltaReflection) != 0, st = (b.type_mask & CTL_EDeltaTransmission) != 0;
        float cosThetaT; const float eta = M.f[0] + M.f[1] / (600 / 1e3f), invEta = 1.0f / eta;
        const float F = fresnel_dielectric_ext(cos_theta(b.wi), cosThetaT, eta);
        if (st && sr) {
            if (smp.x <= F) { b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); b.eta = 1.0f; pdf = F; return tex_eval(M.tex[1], b.dg); }
            b.sampled_type = CTL_EDeltaTransmission; b.wo = refract_local(b.wi, cosThetaT, eta, invEta);
            b.eta = cosThetaT < 0 ? eta : invEta; pdf = (1 - F) * 1.0f;
            const float factor = (cosThetaT < 0 ? invEta : eta);
            return f3(1.0f) * tex_eval(M.tex[0], b.dg) * (factor * factor);
        } else if (sr) { b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); b.eta = 1.0f; pdf = 1.0f; return tex_eval(M.tex[1], b.dg); }
        else if (st) {
            b.sampled_type = CTL_EDeltaTransmission; b.wo = refract_local(b.wi, cosThetaT, eta, invEta);
            b.eta = cosThetaT < 0 ? eta : invEta; pdf = 1.0f * 1.0f;
            const float factor = (cosThetaT < 0 ? invEta : eta);
            return f3(1.0f) * tex_eval(M.tex[0], b.dg) * (factor * factor * (1 - F));
        }
        return f3(0.0f);
    }
    case CTL_BSDF_CONDUCTOR: { if (!CTL_HAS_MODEL(CTL_BSDF_CONDUCTOR)) return f3(0.0f);   // BSDF_Simple.cu:617-630
        if (!(b.type_mask & CTL_EDeltaReflection) || cos_theta(b.wi) <= 0) return f3(0.0f);
        b.sampled_type = CTL_EDeltaReflection; b.wo = reflect_local(b.wi); b.eta = 1.0f; pdf = 1;
        return tex_eval(M.tex[0], b.dg) * fresnel_conductor_exact(cos_theta(b.wi), f3(M.f[0], M.f[1], M.f[2]), f3(M.f[3], M.f[4], M.f[5]));
    }
    case CTL_BSDF_ROUGHCONDUCTOR: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHCONDUCTOR)) return f3(0.0f);   // BSDF_Simple.cu:662-705
        if (cos_theta(b.wi) < 0 || !(b.type_mask & CTL_EGlossyReflection)) return f3(0.0f);
        const microfacet distr((int)M.u[0], avg3(tex_eval(M.tex[1], b.dg)), avg3(tex_eval(M.tex[2], b.dg)), M.u[1] != 0);
        const f3 m = distr.sample(b.wi, smp, pdf);
        if (pdf == 0) return f3(0.0f);
        b.wo = reflect_about(b.wi, m); b.eta = 1.0f; b.sampled_type = CTL_EGlossyReflection;
        if (cos_theta(b.wo) <= 0) return f3(0.0f);
        const f3 F = fresnel_conductor_exact(dot(b.wi, m), f3(M.f[0], M.f[1], M.f[2]), f3(M.f[3], M.f[4], M.f[5])) * tex_eval(M.tex[0], b.dg);
        float weight;
        if (distr.vis) weight = distr.smith_g1(b.wo, m);
        else weight = distr.eval(m) * distr.G(b.wi, b.wo, m) * dot(b.wi, m) / (pdf * cos_theta(b.wi));
        pdf /= 4.0f * dot(b.wo, m);
        return F * weight;
    }
#if CTL_SHADE_FEATURES & 3
    default: return bsdf_more_sample(M, b, pdf, smp);
#else
    default: return f3(0.0f);
#endif
    }
}

// f() and pdf() for solid-angle measure (the only measure the path asks for: TraceAlgorithms.cu:55,61)
__device__ f3 bsdf_f(const ctl_material& M, const bsdf_rec& b) {
    switch (M.bsdf_type) {
    case CTL_BSDF_DIFFUSE: { if (!CTL_HAS_MODEL(CTL_BSDF_DIFFUSE)) return f3(0.0f);   // BSDF_Simple.cu:38-56
        const uint32_t ct = M.combined_type;
        if (!(b.type_mask & ct)) return f3(0.0f);
        const bool vr = ct == CTL_EDiffuseReflection && cos_theta(b.wi) > 0 && cos_theta(b.wo) > 0;
        const bool vt = ct == CTL_EDiffuseTransmission && cos_theta(b.wi) * cos_theta(b.wo) < 0;
        const f3 s = tex_eval(M.tex[0], b.dg) * (kInvPi * fabsf(cos_theta(b.wo)));
        if (vr || vt) return s;
        if (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission)) return s * 0.5f;
        return f3(0.0f);
    }
    case CTL_BSDF_ROUGHCONDUCTOR: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHCONDUCTOR)) return f3(0.0f);   // BSDF_Simple.cu:707-740
        if (cos_theta(b.wi) < 0 || cos_theta(b.wo) < 0 || !(b.type_mask & CTL_EGlossyReflection)) return f3(0.0f);
        const f3 H = normalize(b.wo + b.wi);
        const microfacet distr((int)M.u[0], avg3(tex_eval(M.tex[1], b.dg)), avg3(tex_eval(M.tex[2], b.dg)), M.u[1] != 0);
        const float D = distr.eval(H);
        if (D == 0) return f3(0.0f);
        const f3 F = fresnel_conductor_exact(dot(b.wi, H), f3(M.f[0], M.f[1], M.f[2]), f3(M.f[3], M.f[4], M.f[5])) * tex_eval(M.tex[0], b.dg);
        const float G = distr.G(b.wi, b.wo, H);
        const float value = D * G / (4.0f * cos_theta(b.wi));
        return F * value;
    }
    case CTL_BSDF_DIELECTRIC: case CTL_BSDF_CONDUCTOR: case CTL_BSDF_THINDIELECTRIC: return f3(0.0f);   // delta lobes have no solid-angle density (BSDF_Simple.cu:226-252, 632-646)
#if CTL_SHADE_FEATURES & 3
    default: return bsdf_more_f(M, b);
#else
    default: return f3(0.0f);
#endif
    }
}
__device__ float bsdf_pdf(const ctl_material& M, const bsdf_rec& b) {
    switch (M.bsdf_type) {
    case CTL_BSDF_DIFFUSE: { if (!CTL_HAS_MODEL(CTL_BSDF_DIFFUSE)) return 0.0f;   // BSDF_Simple.cu:58-75
        const uint32_t ct = M.combined_type;
        if (!(b.type_mask & ct)) return 0.0f;
        const bool vr = ct == CTL_EDiffuseReflection && cos_theta(b.wi) > 0 && cos_theta(b.wo) > 0;
        const bool vt = ct == CTL_EDiffuseTransmission && cos_theta(b.wi) * cos_theta(b.wo) < 0;
        const float f = fabsf(kInvPi * cos_theta(b.wo));
        if (vr || vt) return f;
        if (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission)) return f * 0.5f;
        return 0.0f;
    }
    case CTL_BSDF_ROUGHCONDUCTOR: { if (!CTL_HAS_MODEL(CTL_BSDF_ROUGHCONDUCTOR)) return 0.0f;   // BSDF_Simple.cu:742-763
        if (cos_theta(b.wi) < 0 || cos_theta(b.wo) < 0 || !(b.type_mask & CTL_EGlossyReflection)) return 0.0f;
        const f3 H = normalize(b.wo + b.wi);
        const microfacet distr((int)M.u[0], avg3(tex_eval(M.tex[1], b.dg)), avg3(tex_eval(M.tex[2], b.dg)), M.u[1] != 0);
        if (distr.vis) return distr.eval(H) * distr.smith_g1(b.wi, H) / (4.0f * cos_theta(b.wi));
        return distr.pdf(b.wi, H) / (4 * absdot(b.wo, H));
    }
    case CTL_BSDF_DIELECTRIC: case CTL_BSDF_CONDUCTOR: case CTL_BSDF_THINDIELECTRIC: return 0.0f;
#if CTL_SHADE_FEATURES & 3
    default: return bsdf_more_pdf(M, b);
#else
    default: return 0.0f;
#endif
    }
}

} // namespace ctl
#if CTL_SHADE_FEATURES & 16
#include "bsdf_complex.h"
#endif
namespace ctl {
// BSDFALL::sample / f / pdf (SceneTypes/BSDF.h:141-207): the nesting models on top of the simple ones
__device__ __forceinline__ f3 bsdf_sample_top(const ctl_material& M, bsdf_rec& b, float& pdf, f2 smp) {
#if CTL_SHADE_FEATURES & 16
    if (M.bsdf_type >= CTL_BSDF_COATING) return bsdf_complex_sample(M, b, pdf, smp);
#endif
    return bsdf_sample(M, b, pdf, smp);
}
__device__ __forceinline__ f3 bsdf_f_top(const ctl_material& M, const bsdf_rec& b) {
#if CTL_SHADE_FEATURES & 16
    if (M.bsdf_type >= CTL_BSDF_COATING) return bsdf_complex_f(M, b, 1);
#endif
    return bsdf_f(M, b);
}
__device__ __forceinline__ float bsdf_pdf_top(const ctl_material& M, const bsdf_rec& b) {
#if CTL_SHADE_FEATURES & 16
    if (M.bsdf_type >= CTL_BSDF_COATING) return bsdf_complex_pdf(M, b, 1);
#endif
    return bsdf_pdf(M, b);
}

// ---- emitters
enum { kMeasureSolidAngle = 1, kMeasureArea = 3, kMeasureDiscrete = 4 };
struct direct_rec { f3 p, n, ref, refN, d; float pdf, dist; int measure; };

// MonteCarlo::sampleReuse (Math/MonteCarlo.cu:7-14): lower_bound over cdf[0..size]
__device__ __forceinline__ uint32_t sample_reuse(const float* __restrict__ cdf, uint32_t size, float& s, float& pdf) {
    uint32_t lo = 0, n = size + 1;
    while (n > 0) { const uint32_t half = n >> 1; if (cdf[lo + half] < s) { lo += half + 1; n -= half + 1; } else n = half; }
    int index = (int)lo - 1; if (index < 0) index = 0; if (index > (int)size - 1) index = (int)size - 1;
    pdf = cdf[index + 1] - cdf[index];
    s = (s - cdf[index]) / pdf;
    return (uint32_t)index;
}
__device__ __forceinline__ frame light_frame(const ctl_light& L) {
    frame f; f.s = f3(L.to_world[0], L.to_world[1], L.to_world[2]); f.t = f3(L.to_world[4], L.to_world[5], L.to_world[6]); f.n = f3(L.to_world[8], L.to_world[9], L.to_world[10]);
    return f;
}
// SpotLight::falloffCurve (SceneTypes/Light.cu:327-336)
__device__ __forceinline__ f3 spot_falloff(const ctl_light& L, f3 d) {
    const float cosTheta = d.z;
    if (cosTheta <= L.cos_cutoff_angle) return f3(0.0f);
    if (cosTheta >= L.cos_beam_width) return f3(1.0f);
    return f3((L.cutoff_angle - m_acos(cosTheta)) * L.inv_transition_width);
}
__device__ __forceinline__ float interval_to_tent(float sample) {   // Math/Warp.h:13-27
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; } else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - sqrtf(sample));
}
// InfiniteLight::internalSampleDirection (SceneTypes/Light.cu:420-463)
__device__ void env_sample_direction(const dev_scene& S, const ctl_light& L, f2 sample, f3& d, f3& value, float& pdf) {
    const ctl_mipmap& map = S.images[L.env_image];
    const float* cdfRows = (const float*)(scene_anim(S) + L.cdf_rows_index), *cdfCols = (const float*)(scene_anim(S) + L.cdf_cols_index), *rowWeights = (const float*)(scene_anim(S) + L.row_weights_index);
    const float sizeX = (float)map.width, sizeY = (float)map.height;
    float qpdf;
    const uint32_t row = sample_reuse(cdfRows, (uint32_t)sizeY, sample.y, qpdf);
    const uint32_t col = sample_reuse(cdfCols + row * (uint32_t)(sizeX + 1), (uint32_t)sizeX, sample.x, qpdf);
    const f2 pos{ (float)col + interval_to_tent(sample.x), (float)row + interval_to_tent(sample.y) };
    const int xPos = clampi((int)floorf(pos.x), 0, (int)(sizeX - 1)), yPos = clampi((int)floorf(pos.y), 0, (int)(sizeY - 1));
    const float dx1 = pos.x - xPos, dx2 = 1.0f - dx1, dy1 = pos.y - yPos, dy2 = 1.0f - dy1;
    const f3 value1 = mip_fetch(map, xPos, yPos) * dx2 * dy2 + mip_fetch(map, xPos + 1, yPos) * dx1 * dy2;
    const f3 value2 = mip_fetch(map, xPos, yPos + 1) * dx2 * dy1 + mip_fetch(map, xPos + 1, yPos + 1) * dx1 * dy1;
    value = (value1 + value2) * f3(L.env_scale[0], L.env_scale[1], L.env_scale[2]);
    pdf = (luminance(value1) * rowWeights[(int)clampf((float)yPos, 0.0f, sizeY - 1.0f)] +
           luminance(value2) * rowWeights[(int)clampf((float)(yPos + 1), 0.0f, sizeY - 1.0f)]) * L.normalization;
    const float pixX = 2 * kPi / sizeX, pixY = kPi / sizeY;
    const float sinPhi = m_sin(pixX * (pos.x + 0.5f)), cosPhi = m_cos(pixX * (pos.x + 0.5f));
    const float sinTheta = m_sin(pixY * (pos.y + 0.5f)), cosTheta = m_cos(pixY * (pos.y + 0.5f));
    d = f3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= fmaxf(fabsf(sinTheta), 0.000001f);
}
__device__ __forceinline__ f3 xform_dir_transpose(const float* m, f3 d) {   // OrthogonalAffineMap::TransformDirectionTranspose (float4x4.h:424-427)
    return f3(dot(d, f3(m[0], m[4], m[8])), dot(d, f3(m[1], m[5], m[9])), dot(d, f3(m[2], m[6], m[10])));
}
// InfiniteLight::internalPdfDirection (SceneTypes/Light.cu:465-486)
__device__ float env_pdf_direction(const dev_scene& S, const ctl_light& L, f3 d) {
    const ctl_mipmap& map = S.images[L.env_image];
    const float* rowWeights = (const float*)(scene_anim(S) + L.row_weights_index);
    const float sizeX = (float)map.width, sizeY = (float)map.height;
    const f2 uv{ m_atan2(d.x, -d.z) * kInvTwoPi, m_acos(fminf(1.0f, fmaxf(-1.0f, d.y))) * kInvPi };
    const float u = uv.x * sizeX - 0.5f, v = uv.y * sizeY - 0.5f;
    const int xPos = (int)floorf(u), yPos = (int)floorf(v);
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    const f3 value1 = mip_fetch(map, xPos, yPos) * dx2 * dy2 + mip_fetch(map, xPos + 1, yPos) * dx1 * dy2;
    const f3 value2 = mip_fetch(map, xPos, yPos + 1) * dx2 * dy1 + mip_fetch(map, xPos + 1, yPos + 1) * dx1 * dy1;
    const float sinTheta = sqrtf(fmaxf(0.0f, 1 - d.y * d.y));
    return (luminance(value1) * rowWeights[clampi(yPos, 0, (int)sizeY - 1)] + luminance(value2) * rowWeights[clampi(yPos + 1, 0, (int)sizeY - 1)])
        * L.normalization / fmaxf(fabsf(sinTheta), 0.000001f);
}
// InfiniteLight::evalEnvironment (SceneTypes/Light.cu:488-501)
__device__ CTL_LIGHT_MAYBE f3 env_eval(const dev_scene& S, const ctl_light& L, f3 dir) {
    const f3 v = xform_dir_transpose(L.to_world, dir);
    const f2 uv{ m_atan2(v.x, -v.z) * kInvTwoPi, m_acos(fminf(1.0f, fmaxf(-1.0f, v.y))) * kInvPi };
    return mip_triangle(S.images[L.env_image], uv) * f3(L.env_scale[0], L.env_scale[1], L.env_scale[2]);
}
// InfiniteLight::pdfDirect, solid-angle measure (SceneTypes/Light.cu:368-378)
__device__ __forceinline__ float env_pdf_direct(const dev_scene& S, const ctl_light& L, f3 d) { return env_pdf_direction(S, L, xform_dir_transpose(L.to_world, d)); }
#if CTL_SHADE_FEATURES & 8
// ---- area lights with a radiance texture or m_bOrthogonal (SceneTypes/Light.cu:50-53, :67-81, :83-134; Engine/ShapeSet.cu:25-31, :71-91)
__device__ __forceinline__ bool light_needs_uv(const ctl_light& L) { return L.rad_texture.type == CTL_TEX_CHECKER || L.rad_texture.type == CTL_TEX_IMAGE; }
__device__ __forceinline__ f2 shape_tri_uv(const dev_scene& S, const ctl_shape_tri& sn, f2 bary) {   // getUV: TriangleData::getUVSetData(0, a, b, c)
    const uint4 tb = S.tri_data[sn.t_dat * 2 + 1];
    // getUVSetData takes u from the HIGH half of each word and v from the low one (Engine/TriangleData.cu:27) — the other way round than fillDG: the reference looks an
    // area light's radiance texture up with the surface's u and v exchanged (pinned on its own code: tests/golden/scene_lights.npz), and so does this
    const f2 a{ half_to_float((uint16_t)(tb.y >> 16)), half_to_float((uint16_t)tb.y) }, b{ half_to_float((uint16_t)(tb.z >> 16)), half_to_float((uint16_t)tb.z) },
        c{ half_to_float((uint16_t)(tb.w >> 16)), half_to_float((uint16_t)tb.w) };
    const float u = bary.x, v = bary.y, w = 1 - u - v;
    return f2{ u * a.x + v * b.x + w * c.x, u * a.y + v * b.y + w * c.y };
}
__device__ __forceinline__ bool barycentric(f3 p, f3 a, f3 b, f3 c, float& u, float& v) {   // AlgebraHelper::Barycentric (Math/AlgebraHelper.h:46-59)
    const f3 v0 = b - a, v1 = c - a, v2 = p - a;
    const float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    const float denom = d00 * d11 - d01 * d01;
    v = (d11 * d20 - d01 * d21) / denom;
    const float w = (d00 * d21 - d01 * d20) / denom;
    u = 1.0f - v - w;
    return 0 <= v && v <= 1 && 0 <= u && u <= 1 && 0 <= w && w <= 1;
}
__device__ CTL_LIGHT_OUTLINE f2 shape_get_position_uv(const dev_scene& S, const ctl_light& L, f3 pos) {   // ShapeSet::getPosition: first triangle of the set that holds the point
    const ctl_shape_tri* tris = (const ctl_shape_tri*)(scene_anim(S) + L.triangles_index);
    for (uint32_t i = 0; i < L.count; i++) {
        const ctl_shape_tri& sn = tris[i]; f2 b;
        if (barycentric(pos, f3(sn.p[0][0], sn.p[0][1], sn.p[0][2]), f3(sn.p[1][0], sn.p[1][1], sn.p[1][2]), f3(sn.p[2][0], sn.p[2][1], sn.p[2][2]), b.x, b.y)) return shape_tri_uv(S, sn, b);
    }
    return f2{ 0.0f, 0.0f };
}
__device__ CTL_LIGHT_OUTLINE f3 light_radiance_tex(const dev_scene& S, const ctl_light& L, f2 uv) {   // m_rad_texture.Evaluate(dg), dg = {P, bary, uv}
    diff_geom dg; dg.uv = uv; dg.images = S.images;
#ifdef CTL_TEX_PARTIALS
    dg.has_uv_partials = false;
#endif
    return tex_eval(L.rad_texture, dg);
}
#endif
// DiffuseLight / PointLight / SpotLight / DistantLight / InfiniteLight ::sampleDirect (SceneTypes/Light.cu:83-137, 13-31, 287-301, 224-245, 350-366)
// with ShapeSet::SamplePosition (Engine/ShapeSet.cu:51-69)
__device__ CTL_LIGHT_MAYBE f3 light_sample_direct(const dev_scene& S, const ctl_light& L, direct_rec& r, f2 smp) {
    if (L.type == CTL_LIGHT_POINT) {
        r.p = f3(L.position[0], L.position[1], L.position[2]);
        const f3 dir = r.p - r.ref;
        r.dist = length(dir);
        const float invDist = 1.0f / r.dist;
        r.d = dir * invDist; r.n = f3(0.0f); r.pdf = 1; r.measure = kMeasureDiscrete;
        return f3(L.radiance[0], L.radiance[1], L.radiance[2]) * (invDist * invDist);
    }
#if CTL_SHADE_FEATURES & 8
    if (L.type == CTL_LIGHT_SPOT) {   // Light.cu:287-301
        r.p = f3(L.position[0], L.position[1], L.position[2]);
        const f3 dir = r.p - r.ref;
        r.dist = length(dir);
        const float invDist = 1.0f / r.dist;
        r.d = dir * invDist; r.n = f3(0.0f); r.pdf = 1; r.measure = kMeasureDiscrete;
        return f3(L.radiance[0], L.radiance[1], L.radiance[2]) * spot_falloff(L, light_frame(L).to_local(-r.d)) * (invDist * invDist);
    }
    if (L.type == CTL_LIGHT_DISTANT) {   // Light.cu:224-245
        const f3 d = light_frame(L).to_world(f3(0.0f, 0.0f, 1.0f));
        const f3 diskCenter = d * L.bsphere_radius;
        const float distance = dot(r.ref - diskCenter, d);
        if (distance < 0) { r.pdf = 0.0f; return f3(0.0f); }
        r.p = r.ref - distance * d; r.d = -d; r.n = d; r.dist = distance; r.pdf = 1.0f; r.measure = kMeasureDiscrete;
        return f3(L.radiance[0], L.radiance[1], L.radiance[2]);
    }
    if (L.type == CTL_LIGHT_INFINITE) {   // Light.cu:350-366
        f3 value, d; float pdf;
        env_sample_direction(S, L, smp, d, value, pdf);
        d = f3(L.to_world[0] * d.x + L.to_world[1] * d.y + L.to_world[2] * d.z + L.to_world[3] * 0.0f,
               L.to_world[4] * d.x + L.to_world[5] * d.y + L.to_world[6] * d.z + L.to_world[7] * 0.0f,
               L.to_world[8] * d.x + L.to_world[9] * d.y + L.to_world[10] * d.z + L.to_world[11] * 0.0f);
        r.pdf = pdf;
        r.p = f3(L.bsphere_center[0], L.bsphere_center[1], L.bsphere_center[2]) + d * L.bsphere_radius;
        r.n = -normalize(d);
        r.dist = L.bsphere_radius;
        r.d = normalize(d);
        r.measure = kMeasureSolidAngle;
        return sdiv(value, pdf);
    }
#endif
    const float* cdf = (const float*)(scene_anim(S) + L.area_dist_index);
    const ctl_shape_tri* tris = (const ctl_shape_tri*)(scene_anim(S) + L.triangles_index);
    float pdfTri, sc = 1;
#if CTL_SHADE_FEATURES & 8
    f2 uv{ 0.0f, 0.0f };
    if (L.orthogonal) {   // the point of a random triangle's plane straight above / below the reference point (Light.cu:87-107)
        sc = kPi;
        const ctl_shape_tri& sn = tris[sample_reuse(cdf, L.count, smp.x, pdfTri)];   // ShapeSet::sampleTriangle
        const f3 p0(sn.p[0][0], sn.p[0][1], sn.p[0][2]), p1(sn.p[1][0], sn.p[1][1], sn.p[1][2]), p2(sn.p[2][0], sn.p[2][1], sn.p[2][2]);
        const f3 n = normalize(cross(p1 - p0, p2 - p0));
        const float lambda = dot(p0, n) - dot(r.ref, n);
        r.p = r.ref + lambda * n;
        f2 b;
        if (!barycentric(r.p, p0, p1, p2, b.x, b.y)) { r.pdf = 0.0f; return f3(0.0f); }
        r.n = n;
        r.pdf = 1.0f / float(L.count);
        if (light_needs_uv(L)) uv = shape_tri_uv(S, sn, b);
    } else
#endif
    {
        const uint32_t index = sample_reuse(cdf, L.count, smp.y, pdfTri);
        const ctl_shape_tri& sn = tris[index];
        const f2 bary = square_to_uniform_triangle(smp);
        const f3 p0(sn.p[0][0], sn.p[0][1], sn.p[0][2]), p1(sn.p[1][0], sn.p[1][1], sn.p[1][2]), p2(sn.p[2][0], sn.p[2][1], sn.p[2][2]);
        r.p = bary.x * p0 + bary.y * p1 + (1.f - bary.x - bary.y) * p2;
        r.n = f3(sn.n[0], sn.n[1], sn.n[2]);
        r.pdf = 1.0f / L.sum_area;
#if CTL_SHADE_FEATURES & 8
        if (light_needs_uv(L)) uv = shape_tri_uv(S, sn, bary);
#endif
    }
    const f3 dir = r.p - r.ref;
    const float distSquared = len_sqr(dir);
    r.dist = sqrtf(distSquared);
    r.d = dir / r.dist;
    const float dp = absdot(r.d, r.n);
    r.measure = kMeasureSolidAngle;
#if CTL_SHADE_FEATURES & 8
    if (L.orthogonal) r.measure = kMeasureDiscrete; else
#endif
    r.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
    if (dot(r.d, r.refN) >= 0 && dot(r.d, r.n) < 0 && r.pdf != 0) {
#if CTL_SHADE_FEATURES & 8
        if (light_needs_uv(L)) return sdiv(light_radiance_tex(S, L, uv), r.pdf) * sc;
#endif
        return sdiv(f3(L.radiance[0], L.radiance[1], L.radiance[2]), r.pdf) * sc;
    }
    r.pdf = 0.0f;
    return f3(0.0f);
}
// DiffuseLight::pdfDirect (SceneTypes/Light.cu:139-159), solid-angle measure
__device__ __forceinline__ float light_pdf_direct(const ctl_light& L, f3 d, f3 refN, f3 n, float dist) {
    if (L.type != CTL_LIGHT_DIFFUSE) return 0.0f;
#if CTL_SHADE_FEATURES & 8
    if (L.orthogonal) return 0.0f;   // Light.cu:140-141: an orthogonal light has a pdf in the discrete measure only; the path tracer asks in solid angle
#endif
    if (dot(d, refN) >= 0 && dot(d, n) < 0) { const float pdfPos = 1.0f / L.sum_area; return pdfPos * (dist * dist) / absdot(d, n); }
    return 0.0f;
}
// DiffuseLight::eval (SceneTypes/Light.cu:67-81)
__device__ __forceinline__ f3 light_eval(const dev_scene& S, const ctl_light& L, f3 p, f3 sys_n, f3 d) {
    if (L.type != CTL_LIGHT_DIFFUSE || dot(sys_n, d) <= 0) return f3(0.0f);
#if CTL_SHADE_FEATURES & 8
    if (L.orthogonal && dot(d, sys_n) < 1 - 1e-3f) return f3(0.0f);   // DeltaEpsilon (Math/MathFunc.h:26)
    if (light_needs_uv(L)) return light_radiance_tex(S, L, shape_get_position_uv(S, L, p));
#endif
    return f3(L.radiance[0], L.radiance[1], L.radiance[2]);
}
// KernelDynamicScene::sampleEmitter / pdfEmitter (Engine/KernelDynamicScene.cu:25-46).  (Walking all num_lights entries with a uniform index — scalar loads from the
// argument segment instead of a per-lane dependent walk — measured no gain: 1.447 against 1.437 ms per pass, profiles/r05_shade_experiments.log.)
struct emitter_choice { uint32_t light; float fL, fU; };
__device__ __forceinline__ emitter_choice choose_emitter(const dev_scene& S, float sx) {
    const uint32_t n = S.num_lights;
    uint32_t idx = 0;
    while (idx < n && !(sx < S.light_cdf[idx])) idx++;   // upper_bound
    if (idx >= n) idx = n - 1;
    return emitter_choice{ S.light_indices[idx], idx > 0 ? S.light_cdf[idx - 1] : 0.0f, S.light_cdf[idx] };
}
__device__ __forceinline__ int sample_emitter(const dev_scene& S, float& emPdf, float sx) {
    if (S.num_lights == 0) return -1;
    const emitter_choice c = choose_emitter(S, sx);
    emPdf = c.fU - c.fL;
    return (int)c.light;
}
// the same choice, re-using the sample: sample.x = (sample.x - fL) / (fU - fL) (KernelDynamicScene.cu:36) — what sampleEmitterDirect hands on to the light
__device__ __forceinline__ int sample_emitter_reuse(const dev_scene& S, float& emPdf, float& sx) {
    if (S.num_lights == 0) return -1;
    const emitter_choice c = choose_emitter(S, sx);
    sx = (sx - c.fL) / (c.fU - c.fL);
    emPdf = c.fU - c.fL;
    return (int)c.light;
}
__device__ __forceinline__ float pdf_emitter(const dev_scene& S, uint32_t light) { return S.light_cdf[light] - (light == 0 ? 0.0f : S.light_cdf[light - 1]); }

} // namespace ctl
