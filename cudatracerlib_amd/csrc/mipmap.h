// mipmap.h — KernelMIPMap level 0 on the device, and the material alpha test that single-ray traversal runs on candidate hits
// (Material::AlphaTest, Engine/Material.cu:141-190, called from TraceHelper.cu:135-153).  Shared by the shade kernels and the
// traversal kernels.
#pragma once
#include "device_scene.h"

namespace ctl {

__device__ __forceinline__ float luminance(f3 s) { return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f; }   // Spectrum.cu:174-177

// ---- KernelMIPMap level 0 (Engine/MIPMap.cu:21-57,116-121,155-172; MIPMap_device.h:34-55)
__device__ __forceinline__ f3 texel_decode(uint32_t v, uint32_t type) {
    const uint32_t x = v & 0xff, y = (v >> 8) & 0xff, z = (v >> 16) & 0xff, w = v >> 24;
    if (type == CTL_TEXEL_RGBE) {   // SpectrumConverter::RGBEToFloat3 (Math/Spectrum.h:557-565)
        if (!w) return f3(0.0f);
        const float e = ldexpf(1.0f, (int)w - (128 + 8));
        return f3(x * e, y * e, z * e);
    }
    return f3(float(x) / 255.0f, float(y) / 255.0f, float(z) / 255.0f);   // COLORREFToFloat3 (:528-532)
}
__device__ __forceinline__ float fracf_(float f) { return f - floorf(f); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ bool wrap_coordinates(f2 uv, f2 dim, uint32_t w, f2& loc) {
    switch (w) {
    case CTL_WRAP_REPEAT: loc = f2{ fracf_(uv.x) * dim.x, fracf_(1.0f - uv.y) * dim.y }; return true;
    case CTL_WRAP_CLAMP: loc = f2{ clampf(uv.x, 0.0f, 1.0f) * dim.x, clampf(1.0f - uv.y, 0.0f, 1.0f) * dim.y }; return true;
    case CTL_WRAP_MIRROR: {   // the reference tests the parity of uv.x for both axes
        const float lx = (int)uv.x % 2 == 0 ? fracf_(uv.x) : 1.0f - fracf_(uv.x), ly = (int)uv.x % 2 == 0 ? fracf_(uv.y) : 1.0f - fracf_(uv.y);
        loc = f2{ lx * dim.x, ly * dim.y }; return true; }
    case CTL_WRAP_BLACK:
        if (uv.x < 0 || uv.x >= 1 || uv.y < 0 || uv.y >= 1) return false;
        loc = f2{ uv.x * dim.x, uv.y * dim.y }; return true;
    }
    return false;
}
__device__ __forceinline__ f3 mip_texel(const ctl_mipmap& M, f2 uv) {
    f2 l;
    if (!wrap_coordinates(uv, f2{ (float)M.width, (float)M.height }, M.wrap_mode, l)) return f3(0.0f);
    const int x = clampi((int)l.x, 0, (int)M.width - 1), y = clampi((int)l.y, 0, (int)M.height - 1);
    return texel_decode(M.texels[(size_t)y * M.width + x], M.texel_type);
}
// The four texels of a bilinear lookup (and of evalGradient): addresses first, then the four loads together, then the decoding — the values and the order of the arithmetic are
// mip_texel's; one trip to memory instead of four behind each other (a texel that wraps to "outside" reads texel 0 and counts as black, like mip_texel's early return).
__device__ __forceinline__ bool mip_texel_index(const ctl_mipmap& M, f2 uv, uint32_t& idx) {
    f2 l; idx = 0u;
    if (!wrap_coordinates(uv, f2{ (float)M.width, (float)M.height }, M.wrap_mode, l)) return false;
    const int x = clampi((int)l.x, 0, (int)M.width - 1), y = clampi((int)l.y, 0, (int)M.height - 1);
    idx = (uint32_t)y * M.width + (uint32_t)x;
    return true;
}
__device__ __forceinline__ void mip_texels4(const ctl_mipmap& M, f2 a, f2 b, f2 c, f2 d, f3& ta, f3& tb, f3& tc, f3& td) {
    uint32_t ia, ib, ic, id;
    const bool va = mip_texel_index(M, a, ia), vb = mip_texel_index(M, b, ib), vc = mip_texel_index(M, c, ic), vd = mip_texel_index(M, d, id);
    const uint32_t* __restrict__ t = M.texels;
    const uint32_t wa = t[ia], wb = t[ib], wc = t[ic], wd = t[id];
    const uint32_t type = M.texel_type;
    ta = va ? texel_decode(wa, type) : f3(0.0f); tb = vb ? texel_decode(wb, type) : f3(0.0f); tc = vc ? texel_decode(wc, type) : f3(0.0f); td = vd ? texel_decode(wd, type) : f3(0.0f);
}
__device__ __forceinline__ f3 mip_triangle(const ctl_mipmap& M, f2 uv) {
    const f2 sz{ (float)M.width, (float)M.height }, is{ 1.0f / sz.x, 1.0f / sz.y };
    const float ds = fracf_(uv.x * sz.x), dt = fracf_(uv.y * sz.y);
    f3 t00, t01, t10, t11;
    mip_texels4(M, uv, f2{ uv.x + 0, uv.y + is.y }, f2{ uv.x + is.x, uv.y + 0 }, f2{ uv.x + is.x, uv.y + is.y }, t00, t01, t10, t11);
    return ((1.f - ds) * (1.f - dt)) * t00 + ((1.f - ds) * dt) * t01 + (ds * (1.f - dt)) * t10 + (ds * dt) * t11;
}
__device__ __forceinline__ f3 mip_fetch(const ctl_mipmap& M, int x, int y) {
    x = clampi(x, 0, (int)M.width - 1); y = clampi(y, 0, (int)M.height - 1);
    return texel_decode(M.texels[(size_t)y * M.width + x], M.texel_type);
}

// KernelMIPMap::SampleAlpha (MIPMap.cu:123-138); the texel index is clamped here, the reference does not
__device__ __forceinline__ float mip_sample_alpha(const ctl_mipmap& M, f2 uv) {
    f2 l;
    if (!wrap_coordinates(uv, f2{ (float)M.width, (float)M.height }, M.wrap_mode, l)) return 0.0f;
    if (M.texel_type == CTL_TEXEL_RGBE) return 1.0f;
    const int x = clampi((int)l.x, 0, (int)M.width - 1), y = clampi((int)l.y, 0, (int)M.height - 1);
    return float(M.texels[(size_t)y * M.width + x] >> 24) / 255.0f;
}
// KernelMIPMap::evalGradient (MIPMap.cu:174-191)
__device__ __forceinline__ void mip_eval_gradient(const ctl_mipmap& M, f2 uv, f3& g0, f3& g1) {
    const f2 dim{ (float)M.width, (float)M.height };
    const float u = uv.x * dim.x - 0.5f, v = uv.y * dim.y - 0.5f;
    const int xPos = (int)u, yPos = (int)v;
    const float dx = u - xPos, dy = v - yPos;
    f3 p00, p10, p01, p11;
    mip_texels4(M, f2{ (float)xPos / dim.x, (float)yPos / dim.y }, f2{ ((float)xPos + 1) / dim.x, (float)yPos / dim.y }, f2{ (float)xPos / dim.x, ((float)yPos + 1) / dim.y },
                f2{ ((float)xPos + 1) / dim.x, ((float)yPos + 1) / dim.y }, p00, p10, p01, p11);
    const f3 tmp = p01 + p10 - p11;
    g0 = (p10 + p00 * (dy - 1) - tmp * dy) * dim.x;
    g1 = (p01 + p00 * (dx - 1) - tmp * dx) * dim.y;
}
// ---- the pyramid behind level 0 and the filtered lookup of a first hit with ray differentials (PathTracer plugin only; Engine/MIPMap.cu:21-114,193-278)
__device__ __forceinline__ uint32_t mip_offset(const dev_mip_levels& L, uint32_t level) { return level ? L.offsets[level - 1] : 0u; }
__device__ __forceinline__ f3 mip_texel_l(const ctl_mipmap& M, const dev_mip_levels& L, uint32_t level, f2 uv) {   // KernelMIPMap::Texel(level, uv)
    const int wl = (int)(M.width >> level), hl = (int)(M.height >> level);
    f2 l;
    if (!wrap_coordinates(uv, f2{ (float)wl, (float)hl }, M.wrap_mode, l)) return f3(0.0f);
    const int x = clampi((int)l.x, 0, wl - 1), y = clampi((int)l.y, 0, hl - 1);
    return texel_decode(M.texels[(size_t)mip_offset(L, level) + (size_t)y * wl + x], M.texel_type);
}
__device__ __forceinline__ f3 mip_triangle_l(const ctl_mipmap& M, const dev_mip_levels& L, uint32_t level, f2 uv) {   // KernelMIPMap::triangle(level, uv)
    level = level > L.levels - 1 ? L.levels - 1 : level;
    const f2 sz{ (float)(M.width >> level), (float)(M.height >> level) }, is{ 1.0f / sz.x, 1.0f / sz.y };
    const float ds = fracf_(uv.x * sz.x), dt = fracf_(uv.y * sz.y);
    return ((1.f - ds) * (1.f - dt)) * mip_texel_l(M, L, level, uv) + ((1.f - ds) * dt) * mip_texel_l(M, L, level, f2{ uv.x + 0, uv.y + is.y }) +
           (ds * (1.f - dt)) * mip_texel_l(M, L, level, f2{ uv.x + is.x, uv.y + 0 }) + (ds * dt) * mip_texel_l(M, L, level, f2{ uv.x + is.x, uv.y + is.y });
}
__device__ inline f3 mip_eval_ewa(const ctl_mipmap& M, const dev_mip_levels& L, const float* __restrict__ lut, uint32_t level, f2 uv, float A, float B, float C) {   // KernelMIPMap::evalEWA
    if (level >= L.levels) return mip_texel_l(M, L, L.levels - 1, f2{ 0, 0 });
    const f2 size{ (float)(M.width >> level), (float)(M.height >> level) };
    const float u = uv.x * size.x - 0.5f, v = uv.y * size.y - 0.5f;
    const f2 ratio{ size.x / (float)M.width, size.y / (float)M.height };
    A /= ratio.x * ratio.x; B /= ratio.x * ratio.y; C /= ratio.y * ratio.y;
    const float invDet = 1.0f / (-B * B + 4.0f * A * C), deltaU = 2.0f * sqrtf(C * invDet), deltaV = 2.0f * sqrtf(A * invDet);
    const int u0 = (int)ceilf(u - deltaU), u1 = (int)floorf(u + deltaU), v0 = (int)ceilf(v - deltaV), v1 = (int)floorf(v + deltaV);
    const float As = A * 64, Bs = B * 64, Cs = C * 64;
    f3 result(0.0f); float denominator = 0.0f; const float ddq = 2 * As, uu0 = u0 - u;
    for (int vt = v0; vt <= v1; ++vt) {
        const float vv = vt - v;
        float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv, dq = As * (2 * uu0 + 1) + Bs * vv;
        for (int ut = u0; ut <= u1; ++ut) {
            if (q < 64) { const unsigned qi = (unsigned)q; if (qi < 64) { const float w = lut[(int)q]; result = result + mip_texel_l(M, L, level, f2{ (float)ut / size.x, (float)vt / size.y }) * w; denominator += w; } }
            q += dq; dq += ddq;
        }
    }
    if (denominator == 0) return mip_triangle_l(M, L, level, uv);
    return sdiv(result, denominator);
}
__device__ __noinline__ f3 mip_eval(const ctl_mipmap& M, const dev_mip_levels& L, const float* __restrict__ lut, f2 uv, f2 d0, f2 d1) {   // KernelMIPMap::eval(uv, d0, d1)
    const float dimx = (float)M.width, dimy = (float)M.height;
    const float du0 = d0.x * dimx, dv0 = d0.y * dimy, du1 = d1.x * dimx, dv1 = d1.y * dimy, du = (du0 + du1) / 2.0f, dv = (dv0 + dv1) / 2.0f;
    if (M.filter_mode == CTL_FILTER_POINT) return mip_texel_l(M, L, 0, uv);
    if (M.filter_mode == CTL_FILTER_BILINEAR) return mip_triangle_l(M, L, 0, uv);
    if (M.filter_mode == CTL_FILTER_TRILINEAR) {
        const float levela = m_log(dimx / fabsf(du)) / m_log(2.0f), levelb = m_log(dimy / fabsf(dv)) / m_log(2.0f),   /* math::log2 on the reference's host path (MathFunc.h:258-265) */ level = (float)L.levels - clampf((levela + levelb) / 2.0f, 1.0f, (float)L.levels);
        const int iLevel = (int)floorf(level), iLevel2 = clampi(iLevel + 1, 0, (int)L.levels - 1);
        const float p = level - iLevel;
        return p * mip_triangle_l(M, L, (uint32_t)iLevel, uv) + (1 - p) * mip_triangle_l(M, L, (uint32_t)iLevel2, uv);
    }
    float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
    const float root = sqrtf((A - C) * (A - C) + B * B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root);
    const float majorRadius = Aprime != 0 ? sqrtf(F / Aprime) : 0; float minorRadius = Cprime != 0 ? sqrtf(F / Cprime) : 0;
    if (!(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        const float level = m_log2(max2(majorRadius, 1e-4f)); const int ilevel = (int)floorf(level);
        if (ilevel < 0) return mip_triangle_l(M, L, 0, uv);
        const float a = level - ilevel;
        return mip_triangle_l(M, L, (uint32_t)ilevel, uv) * (1.0f - a) + mip_triangle_l(M, L, (uint32_t)(ilevel + 1), uv) * a;
    }
    const float maxAnisotropy = 16;
    if (minorRadius * maxAnisotropy < majorRadius) {
        minorRadius = majorRadius / maxAnisotropy;
        const float theta = 0.5f * m_atan(B / (A - C)), sinTheta = m_sin(theta), cosTheta = m_cos(theta);
        const float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta, sin2Theta = 2 * sinTheta * cosTheta;
        A = a2 * cosTheta2 + b2 * sinTheta2; B = (a2 - b2) * sin2Theta; C = a2 * sinTheta2 + b2 * cosTheta2; F = a2 * b2;
    }
    const float scale = 1.0f / F; A *= scale; B *= scale; C *= scale;
    const float level = max2(0.0f, m_log2(minorRadius)); const int ilevel = (int)level; const float a = level - ilevel;
    if (majorRadius < 1 || !(A > 0 && C > 0)) return mip_triangle_l(M, L, (uint32_t)ilevel, uv);
    return mip_eval_ewa(M, L, lut, (uint32_t)ilevel, uv, A, B, C) * (1.0f - a) + mip_eval_ewa(M, L, lut, (uint32_t)(ilevel + 1), uv, A, B, C) * a;
}

__device__ __forceinline__ f2 tex_map_point(const ctl_texture& t, f2 uv) { return f2{ t.uv_scale[0] * uv.x + 0 * uv.y + t.uv_offset[0], 0 * uv.x + t.uv_scale[1] * uv.y + t.uv_offset[1] }; }   // TextureMapping2D::TransformPoint

// sample_fast (Material.cu:141-158): constant / checkerboard / image texture at an interpolated uv
__device__ __forceinline__ f3 tex_eval_uv(const ctl_texture& t, f2 duv, const ctl_mipmap* images) {
    if (t.type == CTL_TEX_CHECKER) {
        const f2 p = tex_map_point(t, duv);
        int xm = (int)(p.x * 2) % 2, ym = (int)(p.y * 2) % 2; if (xm < 0) xm += 2; if (ym < 0) ym += 2;
        return ((2 * xm - 1) * (2 * ym - 1) == 1) ? f3(t.value[0], t.value[1], t.value[2]) : f3(t.value1[0], t.value1[1], t.value1[2]);
    }
    if (t.type == CTL_TEX_IMAGE) {
        if (t.image == 0xffffffffu) return f3(0.0f);
        const ctl_mipmap& M = images[t.image];
        const f2 uv = tex_map_point(t, duv);
        return (M.filter_mode == CTL_FILTER_POINT ? mip_texel(M, uv) : mip_triangle(M, uv)) * f3(t.value[0], t.value[1], t.value[2]);
    }
    return f3(t.value[0], t.value[1], t.value[2]);
}

// Material::AlphaTest for the triangle a ray is about to accept.  Out of line and fed with plain pointers: it sits in the
// traversal loops, where only scenes with alpha maps may pay for it.
__device__ __noinline__ bool alpha_survives(const uint4* __restrict__ tri_data, const uint4* __restrict__ node_info, const ctl_material* __restrict__ mats,
                                            const ctl_mipmap* __restrict__ images, int tri, int node, float u, float v) {
    const uint4 ta = tri_data[tri * 2], tb = tri_data[tri * 2 + 1];
    const ctl_material& mat = mats[node_info[node].x + ((ta.y >> 16) & 0xff)];
    const uint32_t st = mat.alpha_state;
    if (st == CTL_ALPHA_DISABLED) return true;
    // tri->getUVSetData(0, a, b, c) (Kernel/TraceHelper.cu:149 -> Engine/TriangleData.cu:25-32): u from the HIGH half of each word, v from the low one — the other way round than
    // fillDG: the reference looks the alpha map up with the surface's u and v exchanged (the same function is pinned on its own code through ShapeSet, tests/golden/scene_lights.npz)
    const f2 a{ half_to_float((uint16_t)(tb.y >> 16)), half_to_float((uint16_t)tb.y) }, b{ half_to_float((uint16_t)(tb.z >> 16)), half_to_float((uint16_t)tb.z) },
        c{ half_to_float((uint16_t)(tb.w >> 16)), half_to_float((uint16_t)tb.w) };
    const float w = 1 - u - v;
    const f2 uv{ u * a.x + v * b.x + w * c.x, u * a.y + v * b.y + w * c.y };
    const ctl_texture& refl = mat.tex[0];
    if ((st == CTL_ALPHA_MAP_ALPHA && mat.alpha_tex.type == CTL_TEX_IMAGE) || (st == CTL_ALPHA_REFLECTANCE_ALPHA && refl.type == CTL_TEX_IMAGE)) {
        const ctl_texture& t = st == CTL_ALPHA_MAP_ALPHA ? mat.alpha_tex : refl;
        if (t.image == 0xffffffffu) return 0.0f >= mat.alpha_test_scalar;   // an IMAGE texture without an image evaluates to 0 (tex_eval_uv)
        return mip_sample_alpha(images[t.image], tex_map_point(t, uv)) >= mat.alpha_test_scalar;
    }
    const f3 val = tex_eval_uv((st & 4) ? refl : mat.alpha_tex, uv, images);
    if ((st & 3) == 1) return luminance(val) >= mat.alpha_test_scalar;
    if ((st & 3) == 3) {
        const f3 d = val - f3(mat.alpha_test_color[0], mat.alpha_test_color[1], mat.alpha_test_color[2]);
        return max2(max2(fabsf(d.x), fabsf(d.y)), fabsf(d.z)) <= mat.alpha_test_scalar;
    }
    return true;
}

} // namespace ctl
