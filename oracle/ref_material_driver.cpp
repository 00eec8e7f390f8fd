// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_material_driver.cpp — extern "C" driver around the reference's Material::SampleNormalMap (Engine/Material.cu:96-137: normal map, height map through
// KernelMIPMap::evalGradient) and Material::AlphaTest (Material.cu:160-189, with sample_fast :139-158).  Engine/Material.cu is compiled WHOLE and unmodified where it lies
// (oracle/Makefile REF_SRCS); what it calls — ImageTexture::Evaluate / getTexture (the generated unit scene_lights.cpp: Texture.cu:6-43 over g_SceneData), KernelMIPMap::Sample /
// SampleAlpha / evalGradient (MIPMap.cu:13-278), Spectrum::getLuminance / toLinearRGB — is reference code built by the same recipe.
// A query rebuilds a reference Material from the product's flat ctl_material the way INTEGRATION.md's converter maps it back (NormalMap / HeightMap / AlphaMap and, for the
// reflectance-map alpha modes, a diffuse BSDF around tex[0] so that bsdf.As()->getTexture(0) is that texture) and points g_SceneData at the product's scene description.
// This file contains no reference source.
#include <Engine/Material.h>
#include <Engine/DifferentialGeometry.h>
#include <SceneTypes/BSDF.h>
#include "../include/ctl_amd.h"
#include <cstdint>
#include <cstring>
#include <stdexcept>

using namespace CudaTracerLib;
void ref_bind_scene(const ctl_scene_desc* d);   // ref_scene_light_driver.cpp

namespace {
Spectrum spec3(const float* v) { return Spectrum(v[0], v[1], v[2]); }
Texture tex_of(const ctl_texture& t) {
    Texture r;
    const TextureMapping2D map(t.uv_scale[0], t.uv_scale[1], t.uv_offset[0], t.uv_offset[1]);
    if (t.type == CTL_TEX_CHECKER) { CheckerboardTexture c(spec3(t.value), spec3(t.value1), map); r.SetData(c); }
    else if (t.type == CTL_TEX_IMAGE) { ImageTexture it(map, std::string(), spec3(t.value)); it.tex_idx = t.image; r.SetData(it); }
    else { ConstantTexture c(spec3(t.value)); r.SetData(c); }
    return r;
}
Material material_of(const ctl_material& M) {
    Material m;
    if (M.map_kind == CTL_MAP_NORMAL) m.SetNormalMap(tex_of(M.map_tex));
    else if (M.map_kind == CTL_MAP_HEIGHT) m.SetHeightMap(tex_of(M.map_tex));
    if (M.alpha_state != CTL_ALPHA_DISABLED) {
        m.SetAlphaMap(tex_of(M.alpha_tex), (AlphaBlendState)M.alpha_state);
        m.AlphaMap.test_val_scalar = M.alpha_test_scalar; m.AlphaMap.test_val_color = spec3(M.alpha_test_color);
    }
    diffuse d(tex_of(M.tex[0])); m.bsdf.SetData(d);
    return m;
}
}  // namespace

extern "C" {

// Material::SampleNormalMap(dg, wi).  q: 20 floats per query = uv(2), sys.s(3), sys.t(3), sys.n(3), geometric n(3), dpdu(3), dpdv(3); out: 10 floats = used, s(3), t(3), n(3)
int ref_material_sample_normal_map(const ctl_scene_desc* desc, const ctl_material* mat, int n, const float* q, float* out) {
    if (desc) ref_bind_scene(desc);
    const Material m = material_of(*mat);
    for (int i = 0; i < n; i++) {
        const float* a = q + 20 * i; float* o = out + 10 * i;
        DifferentialGeometry dg; std::memset((void*)&dg, 0, sizeof dg);
        dg.uv[0] = Vec2f(a[0], a[1]); dg.hasUVPartials = false;
        dg.sys = Frame(NormalizedT<Vec3f>(a[2], a[3], a[4]), NormalizedT<Vec3f>(a[5], a[6], a[7]), NormalizedT<Vec3f>(a[8], a[9], a[10]));
        dg.n = NormalizedT<Vec3f>(a[11], a[12], a[13]); dg.dpdu = Vec3f(a[14], a[15], a[16]); dg.dpdv = Vec3f(a[17], a[18], a[19]);
        o[0] = m.SampleNormalMap(dg, Vec3f(0.0f, 0.0f, 1.0f)) ? 1.0f : 0.0f;
        const Vec3f f[3] = { dg.sys.s, dg.sys.t, dg.sys.n };
        for (int k = 0; k < 3; k++) { o[1 + 3 * k] = f[k].x; o[2 + 3 * k] = f[k].y; o[3 + 3 * k] = f[k].z; }
    }
    return 0;
}
// Material::AlphaTest(bary, uv).  q: 4 floats per query = bary(2), uv(2); out: 1 = the hit survives
int ref_material_alpha_test(const ctl_scene_desc* desc, const ctl_material* mat, int n, const float* q, int32_t* out) {
    if (desc) ref_bind_scene(desc);
    const Material m = material_of(*mat);
    for (int i = 0; i < n; i++) out[i] = m.AlphaTest(Vec2f(q[4 * i], q[4 * i + 1]), Vec2f(q[4 * i + 2], q[4 * i + 3])) ? 1 : 0;
    return 0;
}

}  // extern "C"
