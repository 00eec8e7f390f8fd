// mip_pyramid.h — host side of KernelMIPMap's pyramid: the levels behind level 0 as MIPMap::CompileToBinary builds them (Engine/MIPMap.cpp:41-95: nLevels = 1 + log2(min(w, h)),
// level i = the 2x2 box average of level i-1 — decoded, summed in the reference's order, scaled by 0.25 and re-encoded in the image's texel type), and what
// ImageTexture::Average() reads off it (SceneTypes/Texture.cu:31-37: getTexture().Sample(Vec2f(0), 1) = Texel(nLevels - 1, (0, 0)), Engine/MIPMap.cu:140-146, :21-44).
// Used by the device scene (tracer.hip: the texel pool of the filtered first-hit lookups) and by the scene builder (BSDF::Update() of models whose sampling weights
// come from texture averages, scene_builder.cpp finalize = the reference's UpdateMaterialsPhase2, Engine/DynamicScene.cpp:84-89).
#pragma once
#include "../../include/ctl_amd.h"
#include "image_io.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace ctl {

struct mip_level_table { uint32_t levels; uint32_t offsets[15]; };   // = dev_mip_levels (device_scene.h): texel offsets of levels 1.. relative to the image's level 0

inline void mip_texel_decode(uint32_t v, uint32_t type, float c[3]) {   // Spectrum::fromRGBE / fromRGBCOL (Math/Spectrum.cu:260-284)
    const uint32_t x = v & 0xff, y = (v >> 8) & 0xff, z = (v >> 16) & 0xff, w = v >> 24;
    if (type == CTL_TEXEL_RGBE) { if (!w) { c[0] = c[1] = c[2] = 0; return; } const float e = std::ldexp(1.0f, (int)w - (128 + 8)); c[0] = x * e; c[1] = y * e; c[2] = z * e; }
    else { c[0] = float(x) / 255.0f; c[1] = float(y) / 255.0f; c[2] = float(z) / 255.0f; }
}

// appends level 0 and the pyramid of `m` to `pool`; returns the offset of level 0 in the pool
inline size_t mip_pyramid_append(const ctl_mipmap& m, std::vector<uint32_t>& pool, mip_level_table& L) {
    const size_t base = pool.size();
    pool.insert(pool.end(), m.texels, m.texels + (size_t)m.width * m.height);
    std::memset(&L, 0, sizeof(L)); L.levels = 1;
    for (uint32_t mn = std::min(m.width, m.height); (mn >>= 1) && L.levels < 16;) L.levels++;
    uint32_t o = m.width * m.height, pw = m.width; size_t prev = base;
    for (uint32_t l = 1, j = m.width / 2, k = m.height / 2; l < L.levels; l++, j >>= 1, k >>= 1) {
        L.offsets[l - 1] = o; pool.resize(base + o + (size_t)j * k);
        for (uint32_t t = 0; t < k; t++) for (uint32_t x = 0; x < j; x++) {
            float a[3], b[3], c[3], e[3];
            mip_texel_decode(pool[prev + (size_t)(2 * t) * pw + 2 * x], m.texel_type, a); mip_texel_decode(pool[prev + (size_t)(2 * t) * pw + 2 * x + 1], m.texel_type, b);
            mip_texel_decode(pool[prev + (size_t)(2 * t + 1) * pw + 2 * x], m.texel_type, c); mip_texel_decode(pool[prev + (size_t)(2 * t + 1) * pw + 2 * x + 1], m.texel_type, e);
            float v[3]; for (int q = 0; q < 3; q++) { float s2 = a[q] + b[q]; s2 = s2 + c[q]; s2 = s2 + e[q]; v[q] = 0.25f * s2; }
            pool[base + o + (size_t)t * j + x] = m.texel_type == CTL_TEXEL_RGBE ? float3_to_rgbe(v[0], v[1], v[2]) : float3_to_rgbcol(v[0], v[1], v[2]);
        }
        prev = base + o; pw = j; o += j * k;
    }
    return base;
}

// KernelMIPMap::Sample(Vec2f(0), 1) of the image's pyramid: the texel of the coarsest level that uv = (0, 0) addresses (WrapCoordinates, Engine/MIPMap_device.h:33-55:
// REPEAT / MIRROR / BLACK -> (0, 0); CLAMP -> x = 0, y = clamp01(1 - 0) * h = the last row)
inline void mip_image_average(const ctl_mipmap& m, float rgb[3]) {
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    if (!m.texels || !m.width || !m.height) return;
    std::vector<uint32_t> pool; mip_level_table L;
    mip_pyramid_append(m, pool, L);
    const uint32_t level = L.levels - 1, wl = m.width >> level, hl = m.height >> level;
    const uint32_t y = m.wrap_mode == CTL_WRAP_CLAMP ? hl - 1 : 0u;
    const size_t off = level ? L.offsets[level - 1] : 0u;
    mip_texel_decode(pool[off + (size_t)y * wl], m.texel_type, rgb);
}

} // namespace ctl
