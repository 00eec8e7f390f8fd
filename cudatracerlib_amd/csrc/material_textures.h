// material_textures.h — BSDF::Update() where it reads Texture::Average() of a texture that is not a constant (host code).
// plastic / roughplastic / phong / ward: m_specularSamplingWeight = sAvg / (dAvg + sAvg) with the luminances of the diffuse and specular reflectance's averages
// (SceneTypes/BSDF_Simple.h:255-264, :298-304, :332-337, :371-376); coating / roughcoating: 1 / (avg(exp(-2 thickness sigmaA.Average())) + 1) (BSDF_Complex.h:37-44, :117-125).
// Texture::Average(): a constant's value (Texture.h:99), a checkerboard's mean of its two colours (Texture.h:148-151), an ImageTexture's scale x the coarsest pyramid texel
// of its bitmap (Texture.cu:31-37 -> mip_pyramid.h).  The reference runs Update() after LoadTextures (MaterialStream::UpdateMaterialsPhase2, Engine/DynamicScene.cpp:74-89);
// the scene builder calls material_update_textures at finalize, when every image is there.  material_factory.h's constructors — used before a scene exists — count a
// bitmap as white; what they make of constants is final.
#pragma once
#include "material_factory.h"
#include "mip_pyramid.h"

namespace ctl {

struct image_set {
    const ctl_mipmap* images = nullptr; uint32_t n = 0;
    float (*cache)[4] = nullptr;   // optional, n entries {r, g, b, valid}: an image's average is made once however many materials use it (a pyramid of a 4096² bitmap is 22 M texels)
};

inline void tex_average(const ctl_texture& t, float out[3], const image_set& I) {
    for (int q = 0; q < 3; q++) out[q] = t.value[q];
    if (t.type == CTL_TEX_CHECKER) { for (int q = 0; q < 3; q++) out[q] = (t.value[q] + t.value1[q]) * 0.5f; }
    else if (t.type == CTL_TEX_IMAGE && I.images) {
        float a[3] = { 0.0f, 0.0f, 0.0f };                 // tex_idx == 0xffffffff: Spectrum(0) (Texture.cu:33-34)
        if (t.image < I.n) {
            if (I.cache && I.cache[t.image][3] != 0.0f) { a[0] = I.cache[t.image][0]; a[1] = I.cache[t.image][1]; a[2] = I.cache[t.image][2]; }
            else { mip_image_average(I.images[t.image], a); if (I.cache) { I.cache[t.image][0] = a[0]; I.cache[t.image][1] = a[1]; I.cache[t.image][2] = a[2]; I.cache[t.image][3] = 1.0f; } }
        }
        for (int q = 0; q < 3; q++) out[q] = a[q] * t.value[q];
    }
}
inline float tex_average_luminance(const ctl_texture& t, const image_set& I) {
    float c[3]; tex_average(t, c, I);
    return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f;   // Spectrum::getLuminance (Spectrum.cu:174-177)
}
// does Update() of this model read the average of a texture that material_factory.h does not average itself (an image; for sigmaA also a checkerboard)?
inline bool material_reads_texture_average(const ctl_material& m) {
    switch (m.bsdf_type) {
    case CTL_BSDF_PLASTIC: case CTL_BSDF_ROUGHPLASTIC: case CTL_BSDF_PHONG: case CTL_BSDF_WARD: return m.tex[0].type == CTL_TEX_IMAGE || m.tex[1].type == CTL_TEX_IMAGE;
    case CTL_BSDF_COATING: case CTL_BSDF_ROUGHCOATING: return m.tex[0].type == CTL_TEX_IMAGE || m.tex[0].type == CTL_TEX_CHECKER;
    default: return false;
    }
}
// the sampling weight of `m` from its textures' averages; everything else in the record stays
inline void material_update_textures(ctl_material& m, const image_set& I) {
    if (!material_reads_texture_average(m)) return;
    if (m.bsdf_type == CTL_BSDF_COATING || m.bsdf_type == CTL_BSDF_ROUGHCOATING) {
        float s[3]; tex_average(m.tex[0], s, I);
        const float th = m.f[2], a = (expf(s[0] * (-2 * th)) + expf(s[1] * (-2 * th)) + expf(s[2] * (-2 * th))) * (1.0f / 3);
        m.f[3] = 1.0f / (a + 1.0f);
        return;
    }
    const float dAvg = tex_average_luminance(m.tex[0], I), sAvg = tex_average_luminance(m.tex[1], I), w = sAvg / (dAvg + sAvg);
    if (m.bsdf_type == CTL_BSDF_PLASTIC) m.f[4] = w; else if (m.bsdf_type == CTL_BSDF_ROUGHPLASTIC) m.f[2] = w; else m.f[0] = w;
}

} // namespace ctl
