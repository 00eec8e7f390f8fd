// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_scene_light_driver.cpp — extern "C" driver around the reference's emitters that sample through the SCENE's buffers: DiffuseLight (area lights:
// ShapeSet::SamplePosition / sampleTriangle / getPosition / PdfTriangle, Engine/ShapeSet.cu:24-105) and InfiniteLight (the environment map's importance sampling:
// internalSampleDirection / internalPdfDirection, SceneTypes/Light.cu:420-479).  Those functions read the global KernelDynamicScene `g_SceneData`; the reference declares
// it in Kernel/TraceHelper.h (which also pulls curand_kernel.h in — not in this image) and defines it in Kernel/TraceHelper.cu.  `make ref` therefore builds ONE
// generated translation unit (oracle/_ref/gen/scene_lights.cpp, git-ignored) out of the reference's own lines: TraceHelper.h:15 (the host declaration) and :24 (the
// `g_SceneData` macro), TraceHelper.cu:27 (the definition), ShapeSet.cu:24-105, Light.cu:420-479 — behind the reference's own headers and the `using std::min; using
// std::max;` preface; no declaration is written by hand.  The same unit carries SceneTypes/Texture.cu:6-43 — ImageTexture::Evaluate / Average / getTexture, the file's whole body,
// which reads the same global — so that image textures (BSDF parameters, area-light radiance) run the reference's own code too.  With it, Light.cu's DiffuseLight::sampleDirect / pdfDirect / eval and InfiniteLight::sampleDirect / pdfDirect —
// compiled all along by the copy of Light.cu without those lines — become callable.
// A query points the global at the product's own scene description (ctl_scene_desc: the anim blob with the lights' triangle CDFs / ShapeSet::triData records /
// environment-map tables, the TriangleData array, level 0 of the images) and rebuilds the reference's light object from the product's flat ctl_light the way
// INTEGRATION.md's converter maps it back.  ShapeSet's members are private: its six words are laid out in the member order of Engine/ShapeSet.h:51-56 (checked against sizeof).
// This file contains no reference source.
#include <SceneTypes/Light.h>
#include <Engine/KernelDynamicScene.h>
#include <Engine/TriangleData.h>
#include "../include/ctl_amd.h"
#include <cstdint>
#include <cstring>
#include <vector>

namespace CudaTracerLib { extern KernelDynamicScene g_SceneDataHost; }   // (declared by the generated unit from TraceHelper.h:15; repeated here only to name it)
using namespace CudaTracerLib;

namespace {
struct shape_layout { unsigned int areaDistributionIndex, areaDistributionLength, trianglesIndex, trianglesLength; float sumArea; unsigned int count; };   // Engine/ShapeSet.h:51-56
static_assert(sizeof(shape_layout) == sizeof(ShapeSet), "member layout of ShapeSet");
static_assert(sizeof(ShapeSet::triData) == sizeof(ctl_shape_tri), "ShapeSet::triData is the anim blob's 64-B record");
static_assert(sizeof(TriangleData) == sizeof(ctl_triangle_data), "TriangleData is 32 B");

Spectrum spec3(const float* v) { return Spectrum(v[0], v[1], v[2]); }
std::vector<KernelMIPMap> g_maps;
void bind_scene(const ctl_scene_desc* d) {
    KernelDynamicScene& S = g_SceneDataHost;
    S.m_sAnimData.Data = (char*)const_cast<uint8_t*>(d->anim); S.m_sAnimData.UsedCount = S.m_sAnimData.Length = (unsigned)d->n_anim_bytes;
    S.m_sTriData.Data = (TriangleData*)const_cast<ctl_triangle_data*>(d->tri_data); S.m_sTriData.UsedCount = S.m_sTriData.Length = d->n_tri_data;
    g_maps.assign(d->n_images, KernelMIPMap());
    for (uint32_t i = 0; i < d->n_images; i++) {
        KernelMIPMap& K = g_maps[i]; std::memset((void*)&K, 0, sizeof K); const ctl_mipmap& m = d->images[i];
        K.m_pHostData = const_cast<unsigned int*>(m.texels); K.m_uWidth = m.width; K.m_uHeight = m.height; K.m_fDim = Vec2f((float)m.width, (float)m.height);
        K.m_uType = (Texture_DataType)m.texel_type; K.m_uWrapMode = (ImageWrap)m.wrap_mode; K.m_uFilterMode = (ImageFilter)m.filter_mode; K.m_uLevels = 1;
    }
    S.m_sTexData.Data = g_maps.data(); S.m_sTexData.UsedCount = S.m_sTexData.Length = d->n_images;
}
ImageTexture image_of(const ctl_texture& t) {   // ImageTexture (SceneTypes/Texture.h:159-183): value = m_scale, uv_scale / uv_offset = the diagonal TextureMapping2D, image = tex_idx
    ImageTexture it(TextureMapping2D(t.uv_scale[0], t.uv_scale[1], t.uv_offset[0], t.uv_offset[1]), std::string(), spec3(t.value));
    it.tex_idx = t.image;
    return it;
}
DiffuseLight area_of(const ctl_light& L) {
    shape_layout sl{ L.area_dist_index, (L.count + 1) * 4u, L.triangles_index, L.count * (unsigned)sizeof(ctl_shape_tri), L.sum_area, L.count };
    ShapeSet s; std::memcpy((void*)&s, &sl, sizeof sl);
    DiffuseLight d(spec3(L.radiance), s, L.node_idx);
    if (L.rad_texture.type == CTL_TEX_CHECKER) {
        const ctl_texture& t = L.rad_texture;
        CheckerboardTexture c(spec3(t.value), spec3(t.value1), TextureMapping2D(t.uv_scale[0], t.uv_scale[1], t.uv_offset[0], t.uv_offset[1])); d.m_rad_texture.SetData(c);
    } else if (L.rad_texture.type == CTL_TEX_IMAGE) { ImageTexture it = image_of(L.rad_texture); d.m_rad_texture.SetData(it); }
    d.m_bOrthogonal = L.orthogonal != 0;
    return d;
}
InfiniteLight env_of(const ctl_scene_desc* d, const ctl_light& L) {
    InfiniteLight e; const KernelMIPMap& K = g_maps[L.env_image];
    e.radianceMap = K;
    e.m_cdfRowsIdx = L.cdf_rows_index; e.m_cdfColsIdx = L.cdf_cols_index; e.m_rowWeightsIdx = L.row_weights_index;
    e.m_cdfRowsLength = (K.m_uHeight + 1) * 4u; e.m_cdfColsLength = K.m_uHeight * (K.m_uWidth + 1) * 4u; e.m_rowWeightsLength = K.m_uHeight * 4u;
    e.m_SceneCenter = Vec3f(L.bsphere_center[0], L.bsphere_center[1], L.bsphere_center[2]); e.m_SceneRadius = L.bsphere_radius;
    e.m_normalization = L.normalization; e.m_size = Vec2f((float)K.m_uWidth, (float)K.m_uHeight); e.m_pixelSize = Vec2f(2 * PI / e.m_size.x, PI / e.m_size.y);   // Light.cpp:26,58
    e.m_scale = spec3(L.env_scale);
    float4x4 M; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = L.to_world[4 * r + c];
    e.m_worldTransform = NormalizedT<OrthogonalAffineMap>(OrthogonalAffineMap(M));
    (void)d; return e;
}
template <class LT> void sample_direct(const LT& light, int n, const float* q, float* out) {
    for (int i = 0; i < n; i++) {
        const float* a = q + 8 * i; float* o = out + 14 * i;
        DirectSamplingRecord d(Vec3f(a[0], a[1], a[2]), NormalizedT<Vec3f>(a[3], a[4], a[5]));
        Spectrum v = light.sampleDirect(d, Vec2f(a[6], a[7]));
        float r, g, b; v.toLinearRGB(r, g, b);
        o[0] = r; o[1] = g; o[2] = b; o[3] = d.pdf; o[4] = d.d.x; o[5] = d.d.y; o[6] = d.d.z; o[7] = d.dist;
        o[8] = d.p.x; o[9] = d.p.y; o[10] = d.p.z; o[11] = d.n.x; o[12] = d.n.y; o[13] = d.n.z;
    }
}
template <class LT> void pdf_direct(const LT& light, int n, const float* q, float* out) {   // q: 14 floats = ref(3), refN(3), d(3), dist, n(3), -
    for (int i = 0; i < n; i++) {
        const float* a = q + 14 * i;
        DirectSamplingRecord d(Vec3f(a[0], a[1], a[2]), NormalizedT<Vec3f>(a[3], a[4], a[5]));
        d.d = NormalizedT<Vec3f>(a[6], a[7], a[8]); d.dist = a[9]; d.n = NormalizedT<Vec3f>(a[10], a[11], a[12]); d.measure = ESolidAngle;
        d.p = d.ref + d.d * d.dist;
        out[i] = light.pdfDirect(d);
    }
}
}  // namespace

void ref_bind_scene(const ctl_scene_desc* d) { bind_scene(d); }   // for ref_material_driver.cpp: the same binding of g_SceneData

extern "C" {

// sampleDirect of light `light` of the scene (area or environment light).  q: 8 floats per query = {ref(3), refN(3), sample(2)}; out: 14 floats per query
// = {value rgb, pdf, d(3), dist, p(3), n(3)}.  Returns -1 for the light types ref_light_driver.cpp drives.
int ref_scene_light_sample_direct(const ctl_scene_desc* desc, uint32_t light, int n, const float* q, float* out) {
    bind_scene(desc); const ctl_light& L = desc->lights[light];
    if (L.type == CTL_LIGHT_DIFFUSE) { const DiffuseLight d = area_of(L); sample_direct(d, n, q, out); return 0; }
    if (L.type == CTL_LIGHT_INFINITE) { const InfiniteLight e = env_of(desc, L); sample_direct(e, n, q, out); return 0; }
    return -1;
}
// ImageTexture::Evaluate(uv) / Average() (SceneTypes/Texture.cu:6-13, 32-38) over the scene's images.  q: 2 floats per query; out: 3 floats per query, then 3 for Average()
int ref_scene_image_texture_eval(const ctl_scene_desc* desc, const ctl_texture* t, int n, const float* q, float* out) {
    if (t->type != CTL_TEX_IMAGE) return -1;
    bind_scene(desc); const ImageTexture it = image_of(*t);
    for (int i = 0; i <= n; i++) {
        const Spectrum s = i < n ? it.Evaluate(Vec2f(q[2 * i], q[2 * i + 1])) : it.Average();
        float r, g, b; s.toLinearRGB(r, g, b); out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
    }
    return 0;
}
// DiffuseLight::eval(p, frame with normal n, d) — with a radiance texture it finds the triangle again through ShapeSet::getPosition.  q: 9 floats = p(3), n(3), d(3); out: 3 per query
int ref_scene_light_eval(const ctl_scene_desc* desc, uint32_t light, int n, const float* q, float* out) {
    bind_scene(desc); const ctl_light& L = desc->lights[light];
    if (L.type != CTL_LIGHT_DIFFUSE) return -1;
    const DiffuseLight d = area_of(L);
    for (int i = 0; i < n; i++) {
        const float* a = q + 9 * i;
        const Spectrum s = d.eval(Vec3f(a[0], a[1], a[2]), Frame(NormalizedT<Vec3f>(a[3], a[4], a[5])), NormalizedT<Vec3f>(a[6], a[7], a[8]));
        float r, g, b; s.toLinearRGB(r, g, b); out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
    }
    return 0;
}
// pdfDirect (solid-angle measure) for a direction d seen from ref that meets the emitter at distance dist with emitter normal n
int ref_scene_light_pdf_direct(const ctl_scene_desc* desc, uint32_t light, int n, const float* q, float* out) {
    bind_scene(desc); const ctl_light& L = desc->lights[light];
    if (L.type == CTL_LIGHT_DIFFUSE) { const DiffuseLight d = area_of(L); pdf_direct(d, n, q, out); return 0; }
    if (L.type == CTL_LIGHT_INFINITE) { const InfiniteLight e = env_of(desc, L); pdf_direct(e, n, q, out); return 0; }
    return -1;
}

}  // extern "C"
