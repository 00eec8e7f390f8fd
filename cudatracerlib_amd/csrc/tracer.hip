// tracer.hip — host side of the plugin: scene upload / re-layout, Image, the Tracer<true> pass loop and the
// WavefrontPathTracer bounce loop (Integrators/PseudoRealtime/WavefrontPathTracer.cu:166-191 re-designed:
// no host synchronisation inside a pass — queue lengths stay on the device).
#include "tracer.h"
#include "scene_builder.h"
#include "flatten.h"
#include "image_io.h"
#include "mitsuba_loader.h"   // unsupported_error
#include "spline.h"
#include "mip_pyramid.h"
#include "ctl_fmath.h"
#include "knobs.h"
#include <algorithm>
#include <thread>
#include <cctype>
#include <cstring>
#include <cstdlib>
#include <string>

namespace ctl {

void throw_hip(hipError_t e, const char* file, int line) {   // ThrowCudaErrors (Defines.cpp:15-29)
    throw hip_error(std::string("In file ") + file + ", line " + std::to_string(line) + " : " + hipGetErrorString(e));
}
int device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
void require_device() { if (device_count() <= 0) throw hip_error("no HIP device: the MI355X path tracer has no CPU fallback"); apply_tuning_from_env(); }

// ------------------------------------------------------------------------------------------------ Scene
Scene::Scene(const ctl_scene_desc& d, bool flatten, int flat_format, bool reduced_rough_transmittance) {
    for (int k = 0; k < 3; k++) { box_min[k] = d.box_min[k]; box_max[k] = d.box_max[k]; }
    near_depth = d.camera.near_depth; far_depth = d.camera.far_depth;
    require_device();
    if (!d.n_nodes) throw std::runtime_error("ctl_scene_create: scene has no nodes");
    if (d.env_map_index != 0xffffffffu && (d.env_map_index >= d.n_lights_buf || d.lights[d.env_map_index].type != CTL_LIGHT_INFINITE))
        throw std::runtime_error("ctl_scene_create: env_map_index does not name an InfiniteLight");
    n_nodes = d.n_nodes;
    std::vector<float4> tmp;
    // scene BVH and mesh BVHs keep the reference's 64-B node (one aligned 64-B fetch group per visit)
    tmp.assign(std::max<size_t>(4, (size_t)d.n_scene_bvh_nodes * 4), make_float4(0, 0, 0, 0));
    if (d.n_scene_bvh_nodes) std::memcpy(tmp.data(), d.scene_bvh_nodes, (size_t)d.n_scene_bvh_nodes * 64);
    top_nodes_.upload(tmp.data(), tmp.size());
    tmp.assign(std::max<size_t>(4, (size_t)d.n_bvh_nodes * 4), make_float4(0, 0, 0, 0));
    if (d.n_bvh_nodes) std::memcpy(tmp.data(), d.bvh_nodes, (size_t)d.n_bvh_nodes * 64);
    bot_nodes_.upload(tmp.data(), tmp.size());
    // leaf entries: Woop rows + index word interleaved to a 64-B stride so a leaf streams as whole 64-B groups
    // (the reference reads 3 float4 from t_tris and 1 uint from t_triIndices, Kernel/TraceHelper.cu:641-644)
    std::vector<std::pair<uint32_t, uint32_t>> ranges;   // (first woop entry, mesh)
    for (uint32_t m = 0; m < d.n_meshes; m++) ranges.emplace_back(d.meshes[m].bvh_tri_offset / 3, m);
    std::sort(ranges.begin(), ranges.end());
    tmp.assign(std::max<size_t>(4, (size_t)d.n_woop * 4), make_float4(0, 0, 0, 0));
    for (size_t r = 0; r < ranges.size(); r++) {
        const uint32_t first = ranges[r].first, last = (r + 1 < ranges.size()) ? ranges[r + 1].first : d.n_woop;
        const ctl_kernel_mesh& km = d.meshes[ranges[r].second];
        for (uint32_t w = first; w < last; w++) {
            const ctl_woop_tri& t = d.woop[w];
            tmp[w * 4 + 0] = make_float4(t.a[0], t.a[1], t.a[2], t.a[3]);
            tmp[w * 4 + 1] = make_float4(t.b[0], t.b[1], t.b[2], t.b[3]);
            tmp[w * 4 + 2] = make_float4(t.c[0], t.c[1], t.c[2], t.c[3]);
            const uint32_t idx = d.woop_index[km.bvh_index_offset + (w - first)].index;
            tmp[w * 4 + 3] = make_float4(__builtin_bit_cast(float, idx), 0, 0, 0);
        }
    }
    leaf_tris_.upload(tmp.data(), tmp.size());
    // instances: inverse transform rows + the mesh offsets the reference fetches from Node / KernelMesh (TraceHelper.cu:530,557-560)
    std::vector<float4> inst((size_t)d.n_nodes * 4), fwd((size_t)d.n_nodes * 3);
    std::vector<uint4> ninfo(d.n_nodes);
    for (uint32_t k = 0; k < d.n_nodes; k++) {
        const float* im = d.node_inv_transforms[k].m; const float* fm = d.node_transforms[k].m;
        if (im[12] != 0.0f || im[13] != 0.0f || im[14] != 0.0f || fm[12] != 0.0f || fm[13] != 0.0f || fm[14] != 0.0f)
            throw std::runtime_error("ctl_scene_create: node transforms must be affine");
        const ctl_node& N = d.nodes[k];
        if (N.mesh_index >= d.n_meshes) throw std::runtime_error("ctl_scene_create: node references a missing mesh");
        const ctl_kernel_mesh& km = d.meshes[N.mesh_index];
        for (int r = 0; r < 3; r++) { inst[k * 4 + r] = make_float4(im[r * 4], im[r * 4 + 1], im[r * 4 + 2], im[r * 4 + 3]); fwd[k * 3 + r] = make_float4(fm[r * 4], fm[r * 4 + 1], fm[r * 4 + 2], fm[r * 4 + 3]); }
        inst[k * 4 + 3] = make_float4(im[15], __builtin_bit_cast(float, km.bvh_node_offset), __builtin_bit_cast(float, km.bvh_tri_offset / 3), __builtin_bit_cast(float, km.tri_offset));
        ninfo[k] = make_uint4(N.material_offset, N.lights[0], N.lights[1], N.n_lights);
    }
    inst_.upload(inst.data(), inst.size()); inst_fwd_.upload(fwd.data(), fwd.size()); node_info_.upload(ninfo.data(), ninfo.size());
    { std::vector<float> lut(1024); normal_codec_lut(lut.data()); normal_lut_.upload((const float2*)lut.data(), 512); }
    static_assert(sizeof(ctl_triangle_data) == 32, "TriangleData is 32 B");
    static_assert(sizeof(ctl_material) == 384 && sizeof(ctl_texture) == 48, "material descriptor layout (include/ctl_amd.h)");
    tri_data_.upload((const uint4*)d.tri_data, (size_t)d.n_tri_data * 2);
    std::vector<ctl_material> dmats(d.materials, d.materials + d.n_materials);   // uploaded below, once the reduced transmittance tables are attached
    for (auto& m : dmats) m.reserved_[0] = m.reserved_[1] = 0;
    if (d.n_lights_buf) lights_.upload(d.lights, d.n_lights_buf); else lights_.alloc(1);
    if (d.n_anim_bytes) anim_.upload(d.anim, d.n_anim_bytes); else anim_.alloc(16);
    // KernelMIPMap of every image: one texel pool (level 0, then the pyramid MIPMap::CompileToBinary builds, Engine/MIPMap.cpp:41-95: nLevels = 1 + log2(min(w, h)),
    // level i = the 2x2 box average of level i-1, decoded, averaged and re-encoded) + a table of level-0 descriptors with device pointers + the level offsets
    {
        std::vector<uint32_t> pool;
        std::vector<size_t> off(d.n_images);
        std::vector<dev_mip_levels> lv(std::max<uint32_t>(1, d.n_images));
        static_assert(sizeof(mip_level_table) == sizeof(dev_mip_levels), "one layout");
        for (uint32_t i = 0; i < d.n_images; i++) {
            const ctl_mipmap& m = d.images[i];
            if (!m.texels || !m.width || !m.height) throw std::runtime_error("ctl_scene_create: empty image");
            mip_level_table L; off[i] = mip_pyramid_append(m, pool, L);   // mip_pyramid.h
            std::memcpy(&lv[i], &L, sizeof(L));
        }
        if (!pool.empty()) texels_.upload(pool.data(), pool.size()); else texels_.alloc(4);
        std::vector<ctl_mipmap> tab(d.n_images);
        for (uint32_t i = 0; i < d.n_images; i++) { tab[i] = d.images[i]; tab[i].texels = texels_.p + off[i]; }
        if (d.n_images) images_.upload(tab.data(), tab.size()); else images_.alloc(1);
        mip_levels_.upload(lv.data(), lv.size());
        float lut[64];   // MIPMap.cpp:87-92
        for (int i = 0; i < 64; i++) { const float r2 = (float)i / (float)(64 - 1); lut[i] = std::exp(-2.0f * r2) - std::exp(-2.0f); }
        mip_lut_.upload(lut, 64);
        S.images = images_.p; S.mip_levels = mip_levels_.p; S.mip_weight_lut = mip_lut_.p;
    }
    // RoughTransmittanceManager's tables (roughplastic): one float pool + 3 descriptors with device pointers
    S.rough_transmittance = nullptr;
    if (d.rough_transmittance) {
        std::vector<float> pool; size_t off_t[3] = {}, off_d[3] = {};
        for (int i = 0; i < 3; i++) {
            const ctl_rough_transmittance& t = d.rough_transmittance[i];
            if (!t.trans || !t.diff_trans) continue;
            const size_t nt = (size_t)2 * t.eta_samples * t.alpha_samples * t.theta_samples, nd = (size_t)2 * t.eta_samples * t.alpha_samples;
            off_t[i] = pool.size(); pool.insert(pool.end(), t.trans, t.trans + nt);
            off_d[i] = pool.size(); pool.insert(pool.end(), t.diff_trans, t.diff_trans + nd);
        }
        if (!pool.empty()) {
            rt_data_.upload(pool.data(), pool.size());
            ctl_rough_transmittance tab[3];
            for (int i = 0; i < 3; i++) { tab[i] = d.rough_transmittance[i]; const bool ok = tab[i].trans && tab[i].diff_trans; tab[i].trans = ok ? rt_data_.p + off_t[i] : nullptr; tab[i].diff_trans = ok ? rt_data_.p + off_d[i] : nullptr; }
            rt_.upload(tab, 3);
            S.rough_transmittance = rt_.p;
        }
    }
    // Rough plastics with a CONSTANT roughness texture look RoughTransmittanceManager's table up at a fixed (alpha, eta) (RoughTransmittance.cu:55-88 -> Math/Spline.cu:376-453).
    // What depends on those two alone is made here, once per material (bsdf_rough.h roughplastic_T):
    //  * default — the sixteen rows the 3-D interpolation reads for this (alpha, eta) and the sixteen products wy * wz of its weights; the device runs the reference's own sum
    //    over them: the same value to the bit, without the warp's two pow(), two sets of spline weights, the strided addressing and the per-tap branch of every lookup;
    //  * CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE (opt-in) — the rows summed over alpha / eta beforehand (a 1-D table, 4 taps instead of 64): equal up to fp32 rounding, which is NOT
    //    equal: the last bits move the rescaled lobe sample, and a texture boundary under the next hit turns that into another colour (profiles/r05_fuzz.log: up to 0.13 % of a
    //    textured scene's pixels beyond the tolerance, 5 % of the bathroom miniature's pixels equal to the bit against 100 %).
    // The arithmetic is the device's: ctl_fmath.h's pow (the kernels' m_pow), fp32, no contraction.
    S.rt_reduced = nullptr;
    if (d.rough_transmittance) {
        std::vector<float> pool;
        for (auto& m : dmats) {
            if (m.bsdf_type != CTL_BSDF_ROUGHPLASTIC || m.tex[2].type != CTL_TEX_CONSTANT || m.u[2] > CTL_MF_PHONG) continue;
            const ctl_rough_transmittance& T = d.rough_transmittance[m.u[2]];
            if (!T.trans || !T.diff_trans || T.theta_samples < 2 || T.alpha_samples < 2 || T.eta_samples < 2) continue;
            float r = m.tex[2].value[0]; r += m.tex[2].value[1]; r += m.tex[2].value[2];
            const float avg = r * (1.0f / 3), alpha = avg > 1e-4f ? avg : 1e-4f;   // avg3 (shading.h) + the MicrofacetDistribution constructor's max(alpha, 1e-4f)
            float eta = m.f[0];
            const float* data = T.trans; const float* ddata = T.diff_trans;
            if (eta < 1) { data += (size_t)T.eta_samples * T.alpha_samples * T.theta_samples; ddata += (size_t)T.eta_samples * T.alpha_samples; eta = 1.0f / eta; }
            if (eta < T.eta_min) eta = T.eta_min;
            const float wa = fm::pow((alpha - T.alpha_min) / (T.alpha_max - T.alpha_min), 0.25f), we = fm::pow((eta - T.eta_min) / (T.eta_max - T.eta_min), 0.25f);
            const float dv = spline_eval_2d(wa, we, ddata, T.alpha_samples, T.eta_samples);
            const float diffuse = std::min(1.0f, std::max(0.0f, dv));   // RoughTransmittanceManager::EvaluateDiffuse clamps
            const uint32_t sx = T.theta_samples;
            if (reduced_rough_transmittance) {
                m.reserved_[0] = (uint32_t)pool.size() + 1; m.reserved_[1] = sx;
                pool.resize(pool.size() + sx);
                spline_reduce_3d_to_1d(wa, we, data, sx, T.alpha_samples, T.eta_samples, pool.data() + pool.size() - sx);
                pool.push_back(diffuse);
                continue;
            }
            if (const char* e = knob_env("CTL_RT_ROWS")) { if (std::atoi(e) == 0) continue; }   // measurement build only (knobs.h): every lookup through the generic 3-D function (profiles/r05_fuzz.log)
            float wy[4], wz[4]; uint32_t ky, kz;
            if (!spline_weights(wa, T.alpha_samples, wy, ky) || !spline_weights(we, T.eta_samples, wz, kz)) continue;   // outside the table: the device's generic lookup returns its 0
            m.reserved_[0] = (uint32_t)pool.size() + 1; m.reserved_[1] = sx | 0x80000000u;   // kRtRows (bsdf_rough.h)
            const size_t base = pool.size();
            pool.resize(base + 16 + (size_t)16 * sx + 1, 0.0f);
            for (int z = -1, zy = 0; z <= 2; ++z) for (int y = -1; y <= 2; ++y, ++zy) {
                const float wyz = wy[y + 1] * wz[z + 1];      // spline_eval_3d's own product
                pool[base + zy] = wyz;
                // a row whose weight is zero is never read (spline_eval_3d skips a zero product); at the table's border it would lie outside
                if (wyz != 0) std::memcpy(&pool[base + 16 + (size_t)zy * sx], data + ((size_t)(kz + z) * T.alpha_samples + (ky + y)) * sx, sizeof(float) * sx);
            }
            pool[base + 16 + (size_t)16 * sx] = diffuse;
            // the device adds every tap where spline_eval_3d skips a zero weight: the same value only over FINITE entries (0 x inf is NaN) — a table that holds anything else
            // keeps the generic lookup
            bool finite = true;
            for (size_t q = base; q < pool.size(); q++) finite = finite && std::isfinite(pool[q]);
            if (!finite) { pool.resize(base); m.reserved_[0] = m.reserved_[1] = 0; }
        }
        if (!pool.empty()) { rt_reduced_.upload(pool.data(), pool.size()); S.rt_reduced = rt_reduced_.p; }
        else for (auto& m : dmats) m.reserved_[0] = 0;
    }
    mats_.upload(dmats.data(), dmats.size());
    // which shade-kernel build this scene needs (kernels.hip launch_shade)
    S.shade_features = 0; S.alpha_maps = 0; S.shade_models = 0;
    for (uint32_t i = 0; i < d.n_lights_buf; i++) if (d.lights[i].type != CTL_LIGHT_POINT && d.lights[i].type != CTL_LIGHT_DIFFUSE) S.shade_features |= kShadeMoreLights;
    for (uint32_t i = 0; i < d.n_materials; i++) {
        const uint32_t t = d.materials[i].bsdf_type;
        S.shade_models |= 1u << (t & 15u);
        if (t == CTL_BSDF_THINDIELECTRIC || t == CTL_BSDF_ROUGHDIELECTRIC || t == CTL_BSDF_PLASTIC || t == CTL_BSDF_PHONG) S.shade_features |= kShadeMoreBsdfs;
        if (t == CTL_BSDF_ROUGHDIFFUSE || t == CTL_BSDF_WARD || t == CTL_BSDF_ROUGHPLASTIC) S.shade_features |= kShadeRoughBsdfs;
        if (t == CTL_BSDF_COATING || t == CTL_BSDF_ROUGHCOATING || t == CTL_BSDF_BLEND) S.shade_features |= kShadeNestingBsdfs | kShadeMoreBsdfs | kShadeRoughBsdfs;
        for (int k = 0; k < 4; k++) if (d.materials[i].tex[k].type == CTL_TEX_IMAGE) S.shade_features |= kShadeImageTextures;
        if (d.materials[i].map_kind != CTL_MAP_NONE) S.shade_features |= kShadeSurfaceMaps | kShadeImageTextures;
        if (d.materials[i].alpha_state != CTL_ALPHA_DISABLED) S.alpha_maps = 1;
        // visible-normal sampling of the Beckmann distribution (erf / erfinv iteration) and the Phong distribution live in the full build only
        if (t == CTL_BSDF_ROUGHCONDUCTOR || t == CTL_BSDF_ROUGHDIELECTRIC || t == CTL_BSDF_ROUGHPLASTIC || t == CTL_BSDF_ROUGHCOATING) {
            const uint32_t dist = t == CTL_BSDF_ROUGHPLASTIC ? d.materials[i].u[2] : d.materials[i].u[0], vis = d.materials[i].u[1];
            if (dist == CTL_MF_PHONG || (dist == CTL_MF_BECKMANN && vis)) S.shade_features |= kShadeMoreMicrofacet;
        }
    }
    for (uint32_t i = 0; i < d.n_lights_buf; i++) {
        const ctl_light& L = d.lights[i];
        if (L.type < CTL_LIGHT_POINT || L.type > CTL_LIGHT_INFINITE) throw std::runtime_error("ctl_scene_create: unknown light type " + std::to_string(L.type));
        if (L.type == CTL_LIGHT_DIFFUSE && (L.orthogonal || L.rad_texture.type == CTL_TEX_CHECKER || L.rad_texture.type == CTL_TEX_IMAGE)) {
            S.shade_features |= kShadeMoreLights;   // orthogonal / textured area lights live in the full build
            if (L.rad_texture.type == CTL_TEX_IMAGE) {
                if (L.rad_texture.image != 0xffffffffu && L.rad_texture.image >= d.n_images) throw std::runtime_error("ctl_scene_create: light texture references a missing image");
                S.shade_features |= kShadeImageTextures;
            }
        }
        if (L.type == CTL_LIGHT_INFINITE && L.env_image >= d.n_images) throw std::runtime_error("ctl_scene_create: InfiniteLight references a missing image");
    }
    for (uint32_t i = 0; i < d.n_materials; i++) {
        const ctl_material& mi = d.materials[i];
        if (mi.map_kind > CTL_MAP_HEIGHT) throw std::runtime_error("ctl_scene_create: unknown surface map kind");
        if (mi.alpha_state > CTL_ALPHA_REFLECTANCE_COLOR || mi.alpha_state == 4) throw std::runtime_error("ctl_scene_create: unknown alpha blend state");
        for (int k = 0; k < 6; k++) {
            if (k == 4 && mi.map_kind == CTL_MAP_NONE) continue;
            if (k == 5 && mi.alpha_state == CTL_ALPHA_DISABLED) continue;
            const ctl_texture& t = k < 4 ? mi.tex[k] : (k == 4 ? mi.map_tex : mi.alpha_tex);
            if (t.type == CTL_TEX_IMAGE && t.image != 0xffffffffu && t.image >= d.n_images) throw std::runtime_error("ctl_scene_create: texture references a missing image");
            if (t.type != CTL_TEX_CONSTANT && t.type != CTL_TEX_CHECKER && t.type != CTL_TEX_IMAGE && t.type != 0) throw std::runtime_error("ctl_scene_create: texture type " + std::to_string(t.type) + " has no HIP implementation yet");
        }
        const uint32_t t = d.materials[i].bsdf_type;
        const bool ok = t == CTL_BSDF_DIFFUSE || t == CTL_BSDF_DIELECTRIC || t == CTL_BSDF_THINDIELECTRIC || t == CTL_BSDF_ROUGHDIELECTRIC || t == CTL_BSDF_CONDUCTOR ||
                        t == CTL_BSDF_ROUGHCONDUCTOR || t == CTL_BSDF_PLASTIC || t == CTL_BSDF_PHONG || t == CTL_BSDF_ROUGHDIFFUSE || t == CTL_BSDF_WARD || t == CTL_BSDF_ROUGHPLASTIC ||
                        t == CTL_BSDF_COATING || t == CTL_BSDF_ROUGHCOATING || t == CTL_BSDF_BLEND;
        if (t == CTL_BSDF_COATING || t == CTL_BSDF_ROUGHCOATING || t == CTL_BSDF_BLEND) {
            for (int k = 0; k < (t == CTL_BSDF_BLEND ? 2 : 1); k++) {
                const uint32_t ni = d.materials[i].u[2 + k];
                if (ni >= d.n_materials || d.materials[ni].bsdf_type >= CTL_BSDF_HK) throw std::runtime_error("ctl_scene_create: nested BSDF index out of range or not a simple BSDF (BSDFFirst)");
            }
            if (t == CTL_BSDF_ROUGHCOATING) {
                const uint32_t slot = d.materials[i].u[0];
                if (slot > CTL_MF_PHONG || !d.rough_transmittance || !d.rough_transmittance[slot].trans) throw std::runtime_error("ctl_scene_create: roughcoating needs the rough-transmittance table of its distribution");
            }
        }
        if (t == CTL_BSDF_ROUGHPLASTIC) {
            const uint32_t slot = d.materials[i].u[2];
            if (slot > CTL_MF_PHONG) throw std::runtime_error("ctl_scene_create: unknown microfacet distribution");
            if (!d.rough_transmittance || !d.rough_transmittance[slot].trans || !d.rough_transmittance[slot].diff_trans)
                throw std::runtime_error("ctl_scene_create: roughplastic needs the rough-transmittance table of its distribution (ctl_builder_set_rough_transmittance)");
        }
        if ((t == CTL_BSDF_ROUGHCONDUCTOR || t == CTL_BSDF_ROUGHDIELECTRIC) && d.materials[i].u[0] > CTL_MF_PHONG)
            throw std::runtime_error("ctl_scene_create: unknown microfacet distribution");
        if (!ok)
            throw std::runtime_error("ctl_scene_create: BSDF type " + std::to_string(t) + " has no HIP implementation yet");
    }
    {   // the two-level traversal keeps (scene-BVH depth + exit marker + mesh-BVH depth) entries on its per-lane stack: check it fits
        auto depth_of = [](const ctl_bvh_node* nodes, size_t n_nodes, int root) {   // child >= 0: float4 index of an inner node
            int best = 0; std::vector<std::pair<int, int>> st;
            if (root >= 0 && (size_t)(root / 4) < n_nodes) st.emplace_back(root / 4, 1);
            while (!st.empty()) {
                const auto [i, dpt] = st.back(); st.pop_back();
                best = std::max(best, dpt);
                if (dpt > 4 * kStackSize) throw std::runtime_error("ctl_scene_create: BVH child links form a cycle");
                for (int c : { nodes[i].child0, nodes[i].child1 }) if (c >= 0 && c != 0x76543210 && (size_t)(c / 4) < n_nodes) st.emplace_back(c / 4, dpt + 1);
            }
            return best;
        };
        const int top = d.scene_start_node >= 0 ? depth_of(d.scene_bvh_nodes, d.n_scene_bvh_nodes, d.scene_start_node) : 0;
        int bottom = 0;
        for (uint32_t m = 0; m < d.n_meshes; m++) {
            const uint32_t first = d.meshes[m].bvh_node_offset / 4;
            if (first < d.n_bvh_nodes) bottom = std::max(bottom, depth_of(d.bvh_nodes + first, d.n_bvh_nodes - first, 0));
        }
        if (top + bottom + 3 > kStackSize)
            throw std::runtime_error("ctl_scene_create: scene BVH depth " + std::to_string(top) + " + mesh BVH depth " + std::to_string(bottom) +
                                     " does not fit the traversal stack of " + std::to_string(kStackSize) + " entries (rebuild the meshes with CTL_BVH_BINNED, whose depth is bounded)");
    }
    S.hit_key_out = nullptr; S.flat_leaf_keys = 0;
    S.flat_nodes = nullptr; S.flat_leaves = nullptr; S.flat_root = 0; S.flat_format = 0; S.flat_compact = 0; S.inst_w_one = 0; S.flat_top_cached = 0;
    if (flatten) {
        // node format: Q4 (64-B 4-wide nodes with 8-bit child boxes) unless the caller or $CTL_FLAT_FORMAT asks for F4 / F2 (DESIGN.md §3 has the measurements)
        flat_scene F;
        if (flatten_scene(d, F, (size_t)1 << 30, flat_format < 0 ? default_flat_format() : flat_format)) {   // up to 2^30 instanced triangles (64 GiB of leaf entries)
            if (F.stack_need() + 2 > (F.format == kFlatQ8 ? kFlat8StackGroups : kStackSize)) throw std::runtime_error("ctl_scene_create: flattened BVH too deep for the traversal stack");
#ifndef CTL_FLAT_EXPERIMENTS
            if (F.format != kFlatQ4 && F.format != kFlatQ8) throw unsupported_error("ctl_scene_create: the F4 / F2 node formats are measurement builds (-DCTL_FLAT_EXPERIMENTS)");
#else
            if (F.format == kFlatQ8) throw unsupported_error("ctl_scene_create: a -DCTL_FLAT_EXPERIMENTS build has no kernel for the 8-wide node format");
#endif
            if (F.format == kFlatQ4) flat_nodes_.upload((const float4*)F.nodes.data(), F.nodes.size() * 4);
            else if (F.format == kFlatQ8) flat_nodes_.upload((const float4*)F.nodes_q8.data(), F.nodes_q8.size() * 8);
            else if (F.format == kFlatF4) flat_nodes_.upload((const float4*)F.nodes_f4.data(), F.nodes_f4.size() * 8);
            else flat_nodes_.upload((const float4*)F.nodes_f2.data(), F.nodes_f2.size() * 4);
            for (const flat_leaf& L : F.leaves) if (L.node >= d.n_nodes) throw std::runtime_error("ctl_scene_create: flattened leaf entry out of range");
            S.flat_leaf_keys = 0;
#ifndef CTL_FLAT_EXPERIMENTS
            // The DEVICE copy of the entries carries the BSDF model of each entry's material in bits 28..31 of its index word (the host arrays, the cache and the oracle's view stay
            // as flatten.cpp made them): a closest-hit traversal leaves it per ray (dev_scene::hit_key_out) and the shade kernel regroups its lanes by it without the
            // hit -> node -> triangle -> material chain of dependent loads that made the regrouping cost more than it won.
            if (d.n_tri_data < (1u << 27) && d.n_materials) {
                for (flat_leaf& L : F.leaves) {
                    const uint32_t tri = L.index >> 1, mi = d.nodes[L.node].material_offset + ((d.tri_data[tri].nor_mat_extra[1] >> 16) & 0xffu);   // TriangleData::getMatIndex (TriangleData.h:40-44)
                    // key 0 is the miss key: a hit whose material does not exist would be partitioned as a hit and shaded from a record outside the scene's materials
                    if (mi >= d.n_materials) throw std::runtime_error("ctl_scene_create: triangle " + std::to_string(tri) + " of node " + std::to_string(L.node) + " names material " + std::to_string(mi) + " of " + std::to_string(d.n_materials));
                    L.index |= (d.materials[mi].bsdf_type & 15u) << 28;
                }
                S.flat_leaf_keys = 1;
            }
            // ... and, in a scene with alpha maps, bit 31 of the node word says that THIS entry's material has one: the alpha-testing traversal kernels (AlphaTest) test those
            // entries only, and the wavefront kernel collects their candidates for a phase of its own (flat_woop_test, intersect_flat)
            if (S.alpha_maps && d.n_materials)
                for (flat_leaf& L : F.leaves) {
                    const uint32_t tri = (S.flat_leaf_keys ? (L.index & 0x0fffffffu) : L.index) >> 1, mi = d.nodes[L.node].material_offset + ((d.tri_data[tri].nor_mat_extra[1] >> 16) & 0xffu);
                    if (mi >= d.n_materials || d.materials[mi].alpha_state != CTL_ALPHA_DISABLED) L.node |= 0x80000000u;
                }
#endif
            F.leaves.emplace_back(); std::memset(&F.leaves.back(), 0, sizeof(flat_leaf)); F.leaves.back().index = 1;   // one spare (closing) entry behind the last leaf
            flat_leaves_.upload((const float4*)F.leaves.data(), F.leaves.size() * 8);
            // every node transform affine with w == 1 exactly (what add_node produces): the kernels skip the load of w and the division by it
            S.inst_w_one = 1; for (uint32_t k = 0; k < d.n_nodes; k++) if (d.node_inv_transforms[k].m[15] != 1.0f) S.inst_w_one = 0;
            CTL_HIP(hipDeviceSynchronize());
            S.flat_nodes = flat_nodes_.p; S.flat_leaves = flat_leaves_.p; S.flat_format = F.format; S.flat_compact = (F.format == kFlatQ4 && F.compact_links) ? 1 : 0;
            S.flat_root = ((S.flat_compact || F.format == kFlatQ8) && F.root_slab) ? 1 : 0;   // bit 0 of an inner link: the node carries an oriented slab (flat_slab.h); Q8 links are node index << 1 | that bit
            S.flat_top_cached = S.flat_compact ? (int)std::min<size_t>(F.nodes.size(), (size_t)flat_top_cache_nodes()) : 0;
        }
    }
    CTL_HIP(hipDeviceSynchronize());
    S.top_nodes = top_nodes_.p; S.bot_nodes = bot_nodes_.p; S.leaf_tris = leaf_tris_.p; S.inst = inst_.p; S.inst_fwd = inst_fwd_.p; S.normal_lut = normal_lut_.p;
    S.tri_data = tri_data_.p; S.node_info = node_info_.p; S.mats = mats_.p; S.lights = lights_.p; S.anim = anim_.p; S.n_lights_buf = d.n_lights_buf; S.n_anim_bytes = (uint32_t)d.n_anim_bytes; S.n_materials_probe = std::max(1u, d.n_materials);
    S.start_node = d.scene_start_node; S.n_nodes = d.n_nodes; S.num_lights = d.num_lights; S.env_map_index = d.env_map_index; S.eps = d.ray_trace_eps;
    for (int i = 0; i < CTL_MAX_NUM_LIGHTS; i++) { S.light_indices[i] = d.light_indices[i]; S.light_cdf[i] = d.light_cdf[i]; }
    // PerspectiveSensor / ThinLensSensor / OrthographicSensor / TelecentricSensor ::Update (SceneTypes/Sensor.cu:76-96, :226-246, :408-427, :515-535)
    const ctl_sensor& c = d.camera;
    if (c.type < CTL_SENSOR_SPHERICAL || c.type > CTL_SENSOR_TELECENTRIC) throw std::runtime_error("ctl_scene_create: unknown sensor type " + std::to_string(c.type));
    const bool ortho = c.type == CTL_SENSOR_ORTHOGRAPHIC || c.type == CTL_SENSOR_TELECENTRIC;
    const float aspect = c.resolution[0] / c.resolution[1];
    const float recip = 1.0f / (c.far_depth - c.near_depth), cot = 1.0f / tanf(c.fov / 2.0f);
    float persp[16] = { cot, 0, 0, 0, 0, cot, 0, 0, 0, 0, c.far_depth * recip, -c.near_depth * c.far_depth * recip, 0, 0, 1, 0 };
    if (ortho) {   // float4x4::orthographic (float4x4.h:625-628) = Scale(1, 1, 1 / (far - near)) % Translate(0, 0, -near)
        const float so[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1.0f / (c.far_depth - c.near_depth), 0, 0, 0, 0, 1 };
        const float to[16] = { 1, 0, 0, 0.0f, 0, 1, 0, 0.0f, 0, 0, 1, -c.near_depth, 0, 0, 0, 1 };
        mat_mul(so, to, persp);
    }
    S.cam.type = c.type; S.cam.aperture_radius = c.aperture_radius; S.cam.focus_distance = c.focus_distance; S.cam.screen_scale_x = c.screen_scale[0] != 0.0f ? c.screen_scale[0] : 1.0f;
    const float sc[16] = { -0.5f, 0, 0, 0, 0, -0.5f * aspect, 0, 0, 0, 0, 1.0f, 0, 0, 0, 0, 1 };
    const float tr[16] = { 1, 0, 0, -1.0f, 0, 1, 0, -1.0f / aspect, 0, 0, 1, 0.0f, 0, 0, 0, 1 };
    float a[16], c2s[16];
    mat_mul(sc, tr, a); mat_mul(a, persp, c2s);
    mat_inverse(c2s, S.cam.s2c);
    std::memcpy(S.cam.to_world, c.to_world, 48);
    S.cam.inv_res[0] = 1.0f / c.resolution[0]; S.cam.inv_res[1] = 1.0f / c.resolution[1];
    {   // m_dx, m_dy (Sensor.cu:86-89): sampleToCamera(1/w, 0, 0) - sampleToCamera(0), projective TransformPoint
        auto tp = [&](float x, float y, float out[3]) {
            float r[4];
            for (int i = 0; i < 4; i++) { float s2 = 0.0f; s2 += S.cam.s2c[i * 4] * x; s2 += S.cam.s2c[i * 4 + 1] * y; s2 += S.cam.s2c[i * 4 + 2] * 0.0f; s2 += S.cam.s2c[i * 4 + 3] * 1.0f; r[i] = s2; }
            out[0] = r[0] / r[3]; out[1] = r[1] / r[3]; out[2] = r[2] / r[3];
        };
        float p0[3], px[3], py[3]; tp(0, 0, p0); tp(S.cam.inv_res[0], 0, px); tp(0, S.cam.inv_res[1], py);
        for (int k = 0; k < 3; k++) { S.cam.dx[k] = px[k] - p0[k]; S.cam.dy[k] = py[k] - p0[k]; }
    }
}

// ------------------------------------------------------------------------------------------------ Image
Image::Image(uint32_t w, uint32_t h) : w_(w), h_(h) { require_device(); px_.alloc((size_t)w * h); Clear(); }
void Image::Clear() { CTL_HIP(hipMemset(px_.p, 0, px_.n * sizeof(ctl_pixel_data))); reduced_ = false; }
void Image::read(ctl_pixel_data* host) { CTL_HIP(hipDeviceSynchronize()); CTL_HIP(hipMemcpy(host, px_.p, px_.n * sizeof(ctl_pixel_data), hipMemcpyDeviceToHost)); }
void Image::write(const ctl_pixel_data* host) { CTL_HIP(hipMemcpy(px_.p, host, px_.n * sizeof(ctl_pixel_data), hipMemcpyHostToDevice)); reduced_ = false; }
void Image::add_samples(uint32_t n, const float* host_samples5) {
    if (!n) return;
    dbuf<float> tmp; tmp.alloc((size_t)n * 5);
    CTL_HIP(hipDeviceSynchronize());   // the frame is complete (a tracer renders on its own stream)
    CTL_HIP(hipMemcpy(tmp.p, host_samples5, (size_t)n * 5 * sizeof(float), hipMemcpyHostToDevice));
    launch_ctx lc{ nullptr, 1024 };
    launch_add_samples(lc, px_.p, w_, h_, n, tmp.p);
    CTL_HIP(hipDeviceSynchronize());
}
void Image::resolve_rgb(float splat_scale, float* host_rgb) {
    if (!rgb_.p) rgb_.alloc(px_.n * 3);
    CTL_HIP(hipDeviceSynchronize());
    launch_ctx lc{ nullptr, 1024 };
    launch_resolve_rgb(lc, px_.p, (uint32_t)px_.n, splat_scale, rgb_.p);
    CTL_HIP(hipDeviceSynchronize());
    CTL_HIP(hipMemcpy(host_rgb, rgb_.p, px_.n * 3 * sizeof(float), hipMemcpyDeviceToHost));
}

void Image::apply_pipeline(float splat_scale, uint32_t* host_rgbcol) {
    if (!out_.p) out_.alloc(px_.n);
    CTL_HIP(hipDeviceSynchronize());
    launch_ctx lc{ nullptr, 1024 };
    launch_apply_pipeline(lc, px_.p, (uint32_t)px_.n, splat_scale, out_.p);
    CTL_HIP(hipDeviceSynchronize());
    CTL_HIP(hipMemcpy(host_rgbcol, out_.p, px_.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
}
void Image::write_file(float splat_scale, const char* path) {
    const std::string p(path); const size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot + 1);
    for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
    if (ext == "hdr" || ext == "pfm") {   // toFreeImage(true): linear float data
        std::vector<float> rgb(px_.n * 3); resolve_rgb(splat_scale, rgb.data());
        if (ext == "hdr") write_hdr(p, rgb.data(), w_, h_); else write_pfm(p, rgb.data(), w_, h_);
    } else if (ext == "png") {            // toFreeImage(false): the processed (gamma-corrected, 8-bit) data
        std::vector<uint32_t> col(px_.n); apply_pipeline(splat_scale, col.data());
        std::vector<float> rgb(px_.n * 3);
        for (size_t i = 0; i < px_.n; i++) for (int c = 0; c < 3; c++) rgb[i * 3 + c] = (float)((col[i] >> (8 * c)) & 0xff) / 255.0f;
        write_png(p, rgb.data(), w_, h_, false);
    } else throw unsupported_error("Failed saving Screenshot! (only .png, .hdr and .pfm writers are built in)");
}

// ------------------------------------------------------------------------------------------------ timing
event_timer::~event_timer() { for (auto& r : used_) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } for (auto e : free_) (void)hipEventDestroy(e); }
hipEvent_t event_timer::get() { if (!free_.empty()) { hipEvent_t e = free_.back(); free_.pop_back(); return e; } hipEvent_t e; CTL_HIP(hipEventCreate(&e)); return e; }
void event_timer::begin(hipStream_t s, int cls) { rec r{ get(), get(), cls }; CTL_HIP(hipEventRecord(r.a, s)); used_.push_back(r); }
void event_timer::end(hipStream_t s) { CTL_HIP(hipEventRecord(used_.back().b, s)); }
void event_timer::collect(double ms_out[5]) {
    for (auto& r : used_) { float ms = 0; CTL_HIP(hipEventElapsedTime(&ms, r.a, r.b)); ms_out[r.cls] += ms; free_.push_back(r.a); free_.push_back(r.b); }
    used_.clear();
}

// ------------------------------------------------------------------------------------------------ TracerBase / Tracer<true>
TracerBase::TracerBase() {
    require_device();
    m_sParameters.addEnum("BlockSamplerType", 0, { "Uniform", "Variance", "Difference", "Select" });   // Tracer.cpp:19 (BlockSamplerTypes::Uniform), an enum parameter as there
    m_sParameters.addInterval("FractionDeterministic", 2, 1, INT_MAX);   // IBlockSampler.h:157-163 (the sampler's own parameter collection there)
    m_sParameters.addInterval("FractionWeighted", 4, 1, INT_MAX);
    CTL_HIP(hipEventCreate(&start)); CTL_HIP(hipEventCreate(&stop));
    CTL_HIP(hipStreamCreate(&stream));
}
TracerBase::~TracerBase() {
    if (start) (void)hipEventDestroy(start);
    if (stop) (void)hipEventDestroy(stop);
    if (stream) (void)hipStreamDestroy(stream);
}
BlockSampler* TracerBase::getBlockSampler() {   // setCorrectBlockSampler (Tracer.cpp:89-101)
    if (w == 0xffffffffu) throw std::runtime_error("the block sampler exists once Resize was called");
    const auto type = (BlockSampler::Type)m_sParameters.getValue("BlockSamplerType");
    if (!block_sampler_ || block_sampler_->type() != type || block_sampler_->width() != w || block_sampler_->height() != h) block_sampler_.reset(new BlockSampler(type, w, h));
    block_sampler_->fraction_deterministic = m_sParameters.getValue("FractionDeterministic");
    block_sampler_->fraction_weighted = m_sParameters.getValue("FractionWeighted");
    return block_sampler_.get();
}
void TracerBase::setSamplerTables(const float* t1, const float* t2) {
    const size_t n1 = (size_t)CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH;
    user_t1.assign(t1, t1 + n1); user_t2.assign(t2, t2 + 2 * n1); have_user_tables = true;
}
void TracerBase::getKernelStats(ctl_tracer_stats& s) const {
    std::memset(&s, 0, sizeof(s));
    s.rays_last_pass = m_uLastNumRaysTraced; s.rays_total = m_uAccNumRaysTraced; s.seconds_last_pass = m_fLastRuntime; s.seconds_total = m_fAccRuntime;
    s.passes_done = m_uPassesDone; s.ms_raygen = kernel_ms[0]; s.ms_intersect = kernel_ms[1]; s.ms_shade = kernel_ms[2]; s.ms_intersect_any = kernel_ms[3]; s.ms_fused = kernel_ms[4];
    s.intersect_rays = intersect_rays; s.intersect_launches = intersect_launches; s.shadow_rays = shadow_rays; s.shadow_launches = shadow_launches; s.fused_launches = fused_launches; s.fused_shadow_rays = fused_shadow_rays; s.fused_closest_rays = fused_closest_rays;
    s.closest_counts = closest_counts; s.any_counts = any_counts;
}

// Tracer<true>::DoPass (Kernel/Tracer.h:209-248), generalised to n passes per call.  As in the reference the host
// regenerates the sampling tables once per pass (UpdateKernel -> SamplingSequenceGeneratorHost::Compute); here the
// generation of pass k+1 overlaps the GPU work of pass k because nothing in a pass synchronises with the host.
template <bool PROGRESSIVE> void Tracer<PROGRESSIVE>::ensureTableRing(unsigned int B) {
    const size_t n1 = (size_t)CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH, n2 = n1 * 2;
    const unsigned int ring = kTableRing;
    if (d_t1.n < n1 * ring * B) { d_t1.alloc(n1 * ring * B); d_t2.alloc(n2 * ring * B); }
    if (h_cap < n1 * ring) {   // pinned staging for the one pass per batch whose tables the caller may hand in (setSamplerTables)
        if (h_t1) { (void)hipHostFree(h_t1); (void)hipHostFree(h_t2); h_t1 = h_t2 = nullptr; }
        CTL_HIP(hipHostMalloc((void**)&h_t1, n1 * ring * sizeof(float))); CTL_HIP(hipHostMalloc((void**)&h_t2, n2 * ring * sizeof(float)));
        h_cap = n1 * ring;
    }
    if (slot_done.empty()) { slot_done.resize(ring); for (auto& e : slot_done) CTL_HIP(hipEventCreate(&e)); }
    if (starts_cap < (size_t)ring * B) {
        if (h_starts) { (void)hipHostFree(h_starts); h_starts = nullptr; }
        CTL_HIP(hipHostMalloc((void**)&h_starts, (size_t)ring * B * sizeof(sequence_generator::pass_start)));
        d_starts.alloc((size_t)ring * B * sizeof(sequence_generator::pass_start) / sizeof(uint32_t));
        starts_cap = (size_t)ring * B;
    }
    if (!d_jumps.p) {
        const std::vector<uint32_t>& J = sequence_generator::chunk_jump_matrices();
        d_jumps.alloc(J.size()); CTL_HIP(hipMemcpy(d_jumps.p, J.data(), J.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
}
void TracerBase::setDepthBuffer(float*, unsigned int, unsigned int) { throw unsupported_error("setDepthBuffer: this tracer is not an IDepthTracer (only the WavefrontPathTracer is, WavefrontPathTracer.h:24)"); }
template <bool PROGRESSIVE> void Tracer<PROGRESSIVE>::Debug(Image* I, unsigned int x, unsigned int y, float rgb[3]) {
    if (!m_pScene) throw std::runtime_error("Debug: InitializeScene was not called");
    if (w == 0xffffffffu) throw std::runtime_error("Debug: Resize was not called");
    if (x >= w || y >= h) throw std::runtime_error("Debug: pixel outside the film");
    // UpdateKernel(scene, generator) -> GenerateNewRandomSequences (Kernel/TraceHelper.cu:182-185): one more set of tables is drawn from the tracer's stream
    const size_t n1 = (size_t)CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH, n2 = n1 * 2;
    ensureTableRing(std::max(1u, passBatch()));
    for (auto e : slot_done) CTL_HIP(hipEventSynchronize(e));
    if (have_user_tables) {
        std::memcpy(h_t1, user_t1.data(), n1 * 4); std::memcpy(h_t2, user_t2.data(), n2 * 4); have_user_tables = false;
        CTL_HIP(hipMemcpyAsync(d_t1.p, h_t1, n1 * 4, hipMemcpyHostToDevice, stream)); CTL_HIP(hipMemcpyAsync(d_t2.p, h_t2, n2 * 4, hipMemcpyHostToDevice, stream));
    } else {
        m_SamplingSequenceGenerator.take_pass_starts(1, h_starts);
        CTL_HIP(hipMemcpyAsync(d_starts.p, h_starts, sizeof(sequence_generator::pass_start), hipMemcpyHostToDevice, stream));
        launch_sequence_fill(stream, d_jumps.p, d_starts.p, 1, d_t1.p, d_t2.p);
    }
    float out[3] = { 0, 0, 0 };
    DebugInternal(I, x, y, d_t1.p, d_t2.p, out);
    CTL_HIP(hipStreamSynchronize(stream));
    if (rgb) { rgb[0] = out[0]; rgb[1] = out[1]; rgb[2] = out[2]; }
}
template <bool PROGRESSIVE> void Tracer<PROGRESSIVE>::DoPasses(Image* I, bool a_NewTrace, unsigned int n) {
    if (!m_pScene) throw std::runtime_error("DoPass: InitializeScene was not called");
    if (w == 0xffffffffu) throw std::runtime_error("DoPass: Resize was not called");
    if (I->getWidth() != w || I->getHeight() != h) throw std::runtime_error("DoPass: image size differs from the tracer size");
    if (n == 0) return;
    if (a_NewTrace || !PROGRESSIVE) { m_uPassesDone = 0; m_uAccNumRaysTraced = 0; m_fAccRuntime = 0; I->Clear(); }
    // a block sampler that does not simply take every block once decides pass by pass, from the frame so far (Tracer.h:209-248)
    BlockSampler* bs = (m_sParameters.getValue("BlockSamplerType") != 0 || block_sampler_) ? getBlockSampler() : nullptr;
    const bool adaptive = bs && !bs->every_block_once();
    if (adaptive && shard_world > 1) throw std::runtime_error("block samplers other than Uniform need the whole frame on one rank");
    if (bs && (a_NewTrace || !PROGRESSIVE)) bs->start_new_rendering(stream);
    const size_t n1 = (size_t)CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH, n2 = n1 * 2;
    // passes are rendered in batches of `B` (one wavefront carries the paths of B passes; each path uses its own pass's
    // tables), B chosen so that a launch holds enough paths to fill 256 CUs even when a rank owns 1/8 of the tiles
    const unsigned int B = adaptive ? 1u : std::min(passBatch(), n);
    const unsigned int ring = kTableRing;   // batches in flight; slot reuse is guarded by an event per slot
    ensureTableRing(B);
    for (int i = 0; i < 5; i++) kernel_ms[i] = 0;
    intersect_rays = intersect_launches = shadow_rays = shadow_launches = fused_launches = fused_shadow_rays = fused_closest_rays = 0;
    CTL_HIP(hipEventRecord(start, stream));
    unsigned int batch_idx = 0;
    for (unsigned int k = 0; k < n; batch_idx++) {
        const unsigned int nb = std::min(B, n - k), slot = batch_idx % ring;
        if (batch_idx >= ring) CTL_HIP(hipEventSynchronize(slot_done[slot]));   // the batch that last used this slot has finished
        // One XORWOW stream as in the reference (Kernel/Sampler.h:57-85).  The host only advances it (one GF(2) jump per pass); the tables of the batch are
        // written in HBM by k_sequence_fill — 256 lanes per pass, each two jumps and 1440 draws from the pass's start state — in the time the host threads
        // needed for one pass, and without the 1.5 MB-per-pass upload (measured: 1.9 ms at the head of every 20-pass call -> 0.1 ms).
        float* dst1 = d_t1.p + (size_t)slot * B * n1; float* dst2 = d_t2.p + (size_t)slot * B * n2;
        unsigned int j0 = 0;
        if (have_user_tables) {   // setSamplerTables: the caller's tables serve the first pass of the call
            float* a = h_t1 + (size_t)slot * n1; float* b = h_t2 + (size_t)slot * n2;
            std::memcpy(a, user_t1.data(), n1 * 4); std::memcpy(b, user_t2.data(), n2 * 4); have_user_tables = false; j0 = 1;
            CTL_HIP(hipMemcpyAsync(dst1, a, n1 * 4, hipMemcpyHostToDevice, stream));
            CTL_HIP(hipMemcpyAsync(dst2, b, n2 * 4, hipMemcpyHostToDevice, stream));
        }
        if (nb > j0) {
            sequence_generator::pass_start* hs = h_starts + (size_t)slot * B;
            uint32_t* ds = d_starts.p + (size_t)slot * B * (sizeof(sequence_generator::pass_start) / sizeof(uint32_t));
            m_SamplingSequenceGenerator.take_pass_starts(nb - j0, hs);
            CTL_HIP(hipMemcpyAsync(ds, hs, (size_t)(nb - j0) * sizeof(sequence_generator::pass_start), hipMemcpyHostToDevice, stream));
            launch_sequence_fill(stream, d_jumps.p, ds, nb - j0, dst1 + (size_t)j0 * n1, dst2 + (size_t)j0 * n2);
        }
        m_uPassesDone += nb;
        std::vector<unsigned char> block_counts;
        pass_block_counts_ = nullptr; pass_max_block_count_ = 1;
        if (adaptive) {
            bs->counts(block_counts);
            pass_block_counts_ = bs->upload_counts(block_counts, stream);
            pass_max_block_count_ = 0; pass_paths_ = 0;
            for (unsigned char c : block_counts) { pass_max_block_count_ = std::max<uint32_t>(pass_max_block_count_, c); pass_paths_ += (uint64_t)c * 4096u; }
        }
        DoRender(I, d_t1.p + (size_t)slot * B * n1, d_t2.p + (size_t)slot * B * n2, nb);
        if (adaptive) bs->add_pass(I->device(), getSplatScale(), block_counts, stream);   // PixelVarianceBuffer::AddPass + IBlockSampler::AddPass; synchronises
        CTL_HIP(hipEventRecord(slot_done[slot], stream));
        k += nb;
    }
    CTL_HIP(hipEventRecord(stop, stream));
    CTL_HIP(hipEventSynchronize(stop));
    CTL_HIP(hipGetLastError());
    float ms = 0; CTL_HIP(hipEventElapsedTime(&ms, start, stop));
    timer.collect(kernel_ms);
    m_fLastRuntime = ms / 1000.0f;
    takeRayCounts(intersect_rays, shadow_rays);
    m_uLastNumRaysTraced = intersect_rays + shadow_rays;
    m_fAccRuntime += m_fLastRuntime; m_uAccNumRaysTraced += m_uLastNumRaysTraced;
}
template class Tracer<true>;

// ------------------------------------------------------------------------------------------------ WavefrontPathTracer
WavefrontPathTracer::WavefrontPathTracer() {
    // WavefrontPathTracer.h:29-39
    m_sParameters.addBool("Direct", true);
    m_sParameters.addInterval("MaxPathLength", 50, 1, INT_MAX);
    m_sParameters.addInterval("RRStartDepth", 5, 1, INT_MAX);
    // build-specific: passes rendered together in one wavefront; 0 = choose so that a launch carries ~64 M paths (passBatch())
    m_sParameters.addInterval("PassBatch", 0, 0, 128);
    // build-specific: run Material::AlphaTest on candidate hits.  Off = the reference's wavefront tracer (its intersectKernel has no
    // alpha test, only the single-ray traceRay of the megakernel integrators does, TraceHelper.cu:135-153)
    m_sParameters.addBool("AlphaTest", false);
    // build-specific: group the shading queue by BSDF model before the full shade kernel runs (no effect on scenes served by the basic build).
    // Off by default: measured on the synthetic-bathroom workload it LOSES (shade 9.1 -> 11.5 ms / pass) — the time goes into the spline
    // lookups of the rough plastics, not into divergence, and the sorted order turns the path-state reads into gathers
    m_sParameters.addBool("SortMaterials", false);
    // build-specific: the full shade kernel regroups the paths of each workgroup by BSDF model before shading them (shade_kernel.inc)
    m_sParameters.addBool("BlockSort", true);
    // build-specific: a scene that needs the full feature set is shaded by one launch per MODEL CLASS present in it (shade_class_a/b/p/c.hip: basic models + misses / the other single-layer
    // models / rough plastic / the nesting models), each register-allocated for its own models, instead of the one kernel that carries all fifteen (k_shade_full: 104 spilled registers).  Needs the
    // BSDF model per hit from the closest-hit traversal (flattened BVH).  false = k_shade_full
    m_sParameters.addBool("ShadeByModelClass", true);
    // build-specific: the rays a shade workgroup emits are appended grouped by direction octant (compaction.h block_append3_keyed).
    // Off by default: measured on synthetic-SM the traversal kernels gain 1 % (6.10 -> 6.05 ms / pass) and the shade kernel pays 0.5 ms for it
    m_sParameters.addBool("SortOctants", false);
    // build-specific: trace a bounce's path rays and the previous bounce's shadow rays in one persistent launch (k_intersect_pair).  Every traversal launch
    // pays ~0.4 ms of ramp and drain whatever its size (tools/shard_time_probe.py: 4 % of the time at 20 passes per launch on one GPU, a quarter on one rank
    // of eight); the fused launch fills the drain of the first set with the second: +2 % on the whole frame, +7 % on a rank of eight.  false = two launches
    m_sParameters.addBool("FuseTraversal", true);
    // build-specific: whose per-path rules the shading stage follows.  PathTrace (default): the reference's deterministic megakernel PathTrace<DIRECT> (PathTracer.cu:10-113),
    // what the oracle pins and the PathTracer plugin renders.  Wavefront: pathIterateKernel's own rules (WavefrontPathTracer.cu:51-164) — Russian roulette before sampling at
    // pathDepth >= RRStartDepth, no sampling / next-event estimation at the last bounce, sampleEmitterDirect with one 2-D sample, 16-bit previous normal, the
    // t >= dDist (1 - eps) shadow rule: what the reference's PT_Wave converges to (darker than PT by the last bounce's direct term, and fewer rays)
    m_sParameters.addEnum("PathSemantics", 0, { "PathTrace", "Wavefront" });
    // build-specific: hit barycentrics through the 16-bit pair of the reference's traversal result (Kernel/TraceHelper.cu:722-731); off = full floats (single-ray traceRay)
    m_sParameters.addBool("U16Barycentrics", false);
    m_sParameters.addInterval("OrderedAccumulationMaxMB", 4096, 0, 1 << 20);   // build-specific: largest stage of the ordered accumulation (MB of HBM); a batch that needs more accumulates with atomics
    m_sParameters.addBool("OrderedAccumulation", true);   // build-specific: finished paths are staged per (pass, pixel) and added to the frame in pass order (kernels.h pass_params::stage); false = four float atomics per path as Image::AddSample does
    int dev = 0; hipDeviceProp_t prop; CTL_HIP(hipGetDevice(&dev)); CTL_HIP(hipGetDeviceProperties(&prop, dev));
    grid_blocks = prop.multiProcessorCount * 8;   // 8 x 256-thread workgroups per CU = 32 waves/CU
}
float4* WavefrontPathTracer::new_f4(size_t n) { f4_.emplace_back(new dbuf<float4>()); f4_.back()->alloc(n); return f4_.back()->p; }

// a ray's slot number travels in 31 bits (bit 31 of the traversal kernels' ray word is the "hit found" flag, traverse_flat.h) and the batch position of a path in 8.
// Checked BEFORE any state changes (Resize, DoRender, reservePasses): a refused size or batch leaves the tracer as it was.
static void check_batch(uint64_t n_local, unsigned int batch, const char* who) {
    if (n_local * std::max(1u, batch) >= (1ull << 31) || batch > 255u)
        throw std::runtime_error(std::string("WavefrontPathTracer::") + who + ": " + std::to_string(n_local) + " pixels x " + std::to_string(batch) + " passes per wavefront exceed the 2^31 ray slots (or 255 passes) of a batch: lower PassBatch");
}
// the stage of the ordered accumulation for a batch of b passes, or nullptr (parameter off, over OrderedAccumulationMaxMB, allocation failed).  Also called by reservePasses:
// the allocation (663 MB for 20 passes of a 1080p frame) then happens before the render, not inside its first call.
float4* WavefrontPathTracer::ensureStage(unsigned int b) {
    if (m_sParameters.getValue("OrderedAccumulation") == 0 || n_local_pixels == 0) return nullptr;
    const size_t need = (size_t)n_local_pixels * b;
    if (need * sizeof(float4) > (size_t)m_sParameters.getValue("OrderedAccumulationMaxMB") << 20) return nullptr;
    if (stage_.n < need) {
        try { stage_.alloc(need); CTL_HIP(hipMemsetAsync(stage_.p, 0, need * sizeof(float4), stream)); }
        catch (const std::exception&) { stage_.free(); (void)hipGetLastError(); }
    }
    return stage_.n >= need ? stage_.p : nullptr;
}
void WavefrontPathTracer::growBatch(unsigned int b, const char* who) {
    if (w == 0xffffffffu || (uint64_t)n_local_pixels * b <= capacity) return;
    check_batch(n_local_pixels, b, who);
    alloc_batch_ = b; Resize(w, h);
}
void WavefrontPathTracer::Resize(unsigned int _w, unsigned int _h) {
    const uint64_t new_local = shard_pixel_count(_w, _h, shard_rank, shard_world);
    check_batch(new_local, 1, "Resize");
    if (new_local * alloc_batch_ >= (1ull << 31)) alloc_batch_ = 1;   // a batch reserved for a smaller frame / shard: the queues grow again with the next render's batch
    Tracer<true>::Resize(_w, _h);
    // DoubleRayBuffer(w*h, w*h) (WavefrontPathTracer.h:59) — here per rank: its tile shard's pixels
    n_local_pixels = shard_pixel_count(_w, _h, shard_rank, shard_world);
    capacity = n_local_pixels * std::max(1u, alloc_batch_);   // grown by DoRender when a larger batch arrives
    f4_.clear();
    for (int b = 0; b < 2; b++) {
        path_soa& p = Q.path[b];
        p.ray_o = new_f4(capacity); p.ray_d = new_f4(capacity); p.thr = new_f4(capacity); p.rad = new_f4(capacity); p.nor = new_f4(capacity); p.pend = new_f4(capacity);
        px_[b].alloc(capacity); p.px = px_[b].p;
        Q.sh_o[b] = new_f4(capacity); Q.sh_d[b] = new_f4(capacity); occ_[b].alloc(capacity); Q.sh_occ[b] = occ_[b].p;
    }
    Q.hit = new_f4(capacity); hit_node_.alloc(capacity); Q.hit_node = hit_node_.p;
    Q.fin.rad = new_f4(capacity); Q.fin.dir = new_f4(capacity); Q.fin.px = new_f4(capacity);
    stats_.alloc(14); Q.stats = stats_.p; CTL_HIP(hipMemset(stats_.p, 0, 14 * sizeof(unsigned long long)));
    Q.capacity = capacity;
    order_.alloc(capacity); Q.order = order_.p; class_order_.free(); for (int c = 0; c < 5; c++) Q.class_order[c] = nullptr;   // the model-class lists (20 B per slot) are allocated by the first render that shades by class
    mat_key_.alloc(capacity); Q.mat_key = mat_key_.p;
    counts_.free(); work_.free(); mat_counts_.free(); stage_.free();
}

// stats: [0] path rays, [1] shadow rays, [2..6] closest-hit traversal counts, [7..11] any-hit traversal counts
void WavefrontPathTracer::takeRayCounts(uint64_t& path_rays, uint64_t& shadow_rays_) {
    unsigned long long r[14];   // [12]: shadow rays of the last bounce, [13]: path rays of the first (never part of a fused launch)
    CTL_HIP(hipMemcpy(r, stats_.p, sizeof(r), hipMemcpyDeviceToHost));
    CTL_HIP(hipMemset(stats_.p, 0, sizeof(r)));
    path_rays = r[0]; shadow_rays_ = r[1];
    fused_shadow_rays = fused_launches ? r[1] - r[12] : 0; fused_closest_rays = fused_launches ? r[0] - r[13] : 0;
    closest_counts = ctl_traversal_counts{ r[2], r[3], r[4], r[5], r[6] };
    any_counts = ctl_traversal_counts{ r[7], r[8], r[9], r[10], r[11] };
}

// WavefrontPathTracer::DoRender (Integrators/PseudoRealtime/WavefrontPathTracer.cu:166-191): ray generation, then per
// bounce {intersect path rays, intersect the previous bounce's shadow rays, shade}.  Queue lengths stay in HBM, so
// the whole pass is enqueued without any host round trip (the reference synchronises 3x per bounce).
unsigned int WavefrontPathTracer::passBatch() const {
    const int v = m_sParameters.getValue("PassBatch");
    if (v > 0) return (unsigned int)v;
    if (w == 0xffffffffu) return 1;
    const uint64_t per_pass = shard_pixel_count(w, h, shard_rank, shard_world);
    if (per_pass == 0) return 1;   // more ranks than 64x64 tiles: this rank owns nothing and its passes are empty launches
    // Paths per launch.  Measured on MI355X (tools/passbatch_probe.py, synthetic-SM 1080p): 2 M paths per launch 1.04 Grays/s, 4 M 1.50, 8 M 1.88,
    // 33 M 2.33, 67 M 2.45, 134 M 2.51 — the persistent traversal kernels keep 524 k rays in flight and every launch pays a ramp and a
    // tail of a few hundred us, so small launches are mostly ramp and tail.  ~350 B of queue state per path: 64 M paths = 22 GB of the 288 GB.
    const uint64_t target = 64u << 20;
    return (unsigned int)std::min<uint64_t>(128, std::max<uint64_t>(1, (target + per_pass - 1) / per_pass));
}

void WavefrontPathTracer::DoRender(Image* I, const float* d_t1p, const float* d_t2p, unsigned int n_batch) {
    growBatch(n_batch, "DoRender");   // the queues grow to the largest batch asked for
    const int maxPathLength = m_sParameters.getValue("MaxPathLength"), rrStart = m_sParameters.getValue("RRStartDepth");
    const bool direct = m_sParameters.getValue("Direct") != 0;
    const size_t n_counts = (size_t)(maxPathLength + 2) * 4, n_work = (size_t)2 * (maxPathLength + 2);
    const size_t n_mat = (size_t)(maxPathLength + 2) * 32;
    if (counts_.n < n_counts) { counts_.alloc(n_counts); work_.alloc(n_work); mat_counts_.alloc(n_mat); }
    Q.counts = counts_.p; Q.work = work_.p; Q.mat_counts = mat_counts_.p;
    const dev_scene& S = m_pScene->S;
    const launch_ctx lc{ stream, grid_blocks, m_sParameters.getValue("AlphaTest") != 0 && S.alpha_maps != 0 };
    pass_params P{};
    P.t1 = d_t1p; P.t2 = (const float2*)d_t2p;
    P.batch = n_batch;
    P.width = w; P.height = h; P.tile_rank = shard_rank; P.tile_world = shard_world; P.n_local_pixels = n_local_pixels;
    P.direct = direct ? 1 : 0; P.max_path_length = maxPathLength; P.rr_start_depth = rrStart;
    P.sort_materials = (m_sParameters.getValue("SortMaterials") != 0 && S.shade_features != 0) ? 1 : 0;
    P.block_sort = m_sParameters.getValue("BlockSort") != 0 ? 1 : 0;
    P.sort_octants = m_sParameters.getValue("SortOctants") != 0 ? 1 : 0;
    P.block_counts = pass_block_counts_; P.max_block_count = pass_max_block_count_;
    P.wavefront_rules = m_sParameters.getValue("PathSemantics") == 1 ? 1 : 0; P.u16_bary = m_sParameters.getValue("U16Barycentrics") != 0 ? 1 : 0;
    P.depth_buffer = depth_buffer_; P.depth_w = depth_w_; P.depth_h = depth_h_; P.depth_near = m_pScene->near_depth; P.depth_far = m_pScene->far_depth;
    // ordered accumulation (kernels.h pass_params::stage): one staged sample per (pass of the batch, pixel), added to the frame in pass order after the last bounce
    // The stage covers THIS RANK'S tiles only (n_local_pixels slots per pass of the batch, 16 B each: 663 MB for 20 passes of a whole 1080p frame, an eighth of that on a rank
    // of eight).  It is an optimisation, not a requirement: past OrderedAccumulationMaxMB (default 4096) or when the allocation fails the paths fall back to Image::AddSample's
    // four float atomics — same sums, hardware order.
    P.stage = nullptr; P.stage_stride = (size_t)n_local_pixels;
    if (!pass_block_counts_) P.stage = ensureStage(n_batch);
    if (pass_block_counts_ && pass_paths_ > capacity) throw std::runtime_error("ray queue overflow: the block sampler asks for more samples in one pass than the queues hold (DoubleRayBuffer.h:86-89)");
    CTL_HIP(hipMemsetAsync(mat_counts_.p, 0, n_mat * sizeof(uint32_t), stream));
    CTL_HIP(hipMemsetAsync(counts_.p, 0, n_counts * sizeof(uint32_t), stream));
    CTL_HIP(hipMemsetAsync(work_.p, 0, n_work * sizeof(uint32_t), stream));
    timer.begin(stream, 0); launch_raygen(lc, S, Q, P); timer.end(stream);
    auto shadow_pass = [&](int d) {   // any-hit intersection of the shadow rays emitted at depth d
        timer.begin(stream, 3);
        if (counting) launch_intersect_count(lc, S, Q.sh_o[d & 1], Q.sh_d[d & 1], &Q.counts[d * 4 + 1], &Q.work[2 * d + 3], nullptr, nullptr, Q.sh_occ[d & 1], 1, Q.stats + 7);
        else launch_intersect_any(lc, S, Q.sh_o[d & 1], Q.sh_d[d & 1], &Q.counts[d * 4 + 1], &Q.work[2 * d + 3], Q.sh_occ[d & 1]);
        timer.end(stream);
        shadow_launches++;
    };
    const bool fuse = !counting && direct && m_sParameters.getValue("FuseTraversal") != 0;
    // closest-hit traversals of a flattened scene leave the BSDF model of every hit in Q.mat_key (device_scene.h hit_key_out) for the shade kernel's regrouping; the counting
    // kernels and the device-wide material sort (which writes its own keys there) do without
    P.model_classes = (S.shade_features != 0 && m_sParameters.getValue("ShadeByModelClass") != 0) ? 1 : 0;
    dev_scene Sk = S; Sk.hit_key_out = (S.flat_leaf_keys && (S.shade_features == 0 || P.model_classes) && !counting && !P.sort_materials && P.block_sort) ? Q.mat_key : nullptr;   // (k_shade_full keys its regrouping by model AND material index)
    P.key_from_traversal = Sk.hit_key_out ? 1 : 0;
    if (!P.key_from_traversal) P.model_classes = 0;
    if (P.model_classes && !class_order_.p) { class_order_.alloc((size_t)capacity * 5); for (int c = 0; c < 5; c++) Q.class_order[c] = class_order_.p + (size_t)c * capacity; }
    for (int depth = 1; depth <= maxPathLength; depth++) {
        const int cur = (depth - 1) & 1;
        if (fuse && depth > 1) {   // path rays of this bounce + shadow rays of the previous one in one launch
            const int d = depth - 1;
            timer.begin(stream, 4);
            launch_intersect_pair(lc, Sk, Q.path[cur].ray_o, Q.path[cur].ray_d, &Q.counts[(depth - 1) * 4 + 0], &Q.work[2 * depth], Q.hit, Q.hit_node,
                                  Q.sh_o[d & 1], Q.sh_d[d & 1], &Q.counts[d * 4 + 1], &Q.work[2 * d + 3], Q.sh_occ[d & 1]);
            timer.end(stream);
            fused_launches++;
            timer.begin(stream, 2);
            launch_finalize(lc, Q, P, depth - 1, I->device());
            launch_shade(lc, S, Q, P, depth, I->device());
            timer.end(stream);
            continue;
        }
        timer.begin(stream, 1);
        if (counting) launch_intersect_count(lc, S, Q.path[cur].ray_o, Q.path[cur].ray_d, &Q.counts[(depth - 1) * 4 + 0], &Q.work[2 * depth], Q.hit, Q.hit_node, nullptr, 0, Q.stats + 2);
        else launch_intersect_closest(lc, Sk, Q.path[cur].ray_o, Q.path[cur].ray_d, &Q.counts[(depth - 1) * 4 + 0], &Q.work[2 * depth], Q.hit, Q.hit_node);
        timer.end(stream);
        intersect_launches++;
        if (depth > 1 && direct) shadow_pass(depth - 1);
        timer.begin(stream, 2);
        if (depth > 1 && direct) launch_finalize(lc, Q, P, depth - 1, I->device());
        launch_shade(lc, S, Q, P, depth, I->device());
        timer.end(stream);
    }
    if (direct) {
        shadow_pass(maxPathLength);
        timer.begin(stream, 2); launch_finalize(lc, Q, P, maxPathLength, I->device()); timer.end(stream);
    }
    if (P.stage) { timer.begin(stream, 2); launch_resolve_stage(lc, P.stage, P.stage_stride, n_batch, I->device(), w, h, shard_rank, shard_world); timer.end(stream); }
    launch_accumulate_stats(lc, Q, maxPathLength);
}

} // namespace ctl
