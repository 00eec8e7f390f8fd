// scene_builder.h — host scene assembly (see scene_builder.cpp).
#pragma once
#include "../../include/ctl_amd.h"
#include "bvh_builder.h"
#include "ctl_math.h"
#include <vector>

namespace ctl {

void mat_inverse(const float* m16, float* out16);   // float4x4::inverse (Math/float4x4.h:132-193)
void mat_mul(const float* l16, const float* r16, float* out16);
void woop_set_data(ctl_woop_tri& w, f3 a, f3 b, f3 c);
void woop_get_data(const ctl_woop_tri& w, f3& v0, f3& v1, f3& v2);

struct mesh_rec { uint32_t tri_offset, n_tris, node_offset, n_nodes, woop_offset, n_woop, mat_offset, n_mat; aabb box; int max_depth; };

struct scene_builder {
    std::vector<ctl_triangle_data> tri;
    std::vector<ctl_woop_tri> woop;
    std::vector<ctl_woop_index> widx;
    std::vector<ctl_bvh_node> bvh, scene_bvh;
    std::vector<ctl_kernel_mesh> meshes;
    std::vector<mesh_rec> mesh_info;
    std::vector<ctl_material> mesh_materials;   // per-mesh standard materials (Mesh::m_sMatInfo)
    std::vector<ctl_material> mats;             // per-node copies (m_sMatData)
    std::vector<ctl_node> nodes;
    std::vector<ctl_float4x4> xf, ixf;
    std::vector<ctl_light> lights;
    std::vector<uint8_t> anim;
    std::vector<std::vector<uint32_t>> image_texels;   // level-0 texels of every registered KernelMIPMap
    std::vector<ctl_mipmap> images;
    uint32_t env_light = 0xffffffffu;
    std::vector<float> rt_trans[3], rt_diff[3];        // RoughTransmittanceManager's three tables
    ctl_rough_transmittance rt[3] = {};
    bool have_rt = false;
    ctl_sensor camera{};
    bool have_camera = false;

    // mesh BVH builder: CTL_BVH_SBVH = the reference's SplitBVHBuilder restated (sbvh_builder.cpp, identical arrays), CTL_BVH_BINNED = binned
    // SAH without spatial splits (bvh_builder.cpp, threaded), CTL_BVH_AUTO = SBVH up to kSbvhAutoLimit = 65536 triangles (its sweep sorts the
    // references three times per node: the reference quotes minutes for San Miguel), binned above.  Default from $CTL_BVH_MODE.
    static constexpr uint32_t kSbvhAutoLimit = 1u << 16;
    uint32_t bvh_mode = default_bvh_mode();
    static uint32_t default_bvh_mode();
    uint32_t finish_mesh(const mesh_rec& mr);   // registers a compiled (or cache-loaded) mesh, returns its index
    uint32_t add_mesh(const float* positions, uint32_t n_vert, const uint32_t* indices, uint32_t n_tri, const float* normals, const float* uvs,
                      const uint8_t* tri_material, const ctl_material* materials, uint32_t n_mat,
                      bool flip_normals = false, bool face_normals = false, float max_smooth_angle = 0.0f);   // Mesh::CompileMesh options (Mesh.cpp:199)
    uint32_t add_node(uint32_t mesh_index, const ctl_float4x4* to_world);
    void set_node_transform(uint32_t node_index, const ctl_float4x4& to_world);
    void set_node_bsdf(uint32_t node_index, uint32_t local_material, const ctl_material& m);
    uint32_t add_aux_material(const ctl_material& m);   // a material no triangle refers to: the nested BSDF of a coating / roughcoating / blend; returns its absolute index
    const ctl_material& node_material(uint32_t node_index, uint32_t local_material) const;
    aabb scene_box() const;
    uint32_t add_area_light(uint32_t node_index, uint32_t local_material, const float radiance[3], const ctl_texture* rad_texture = nullptr, bool orthogonal = false);
    uint32_t add_point_light(const float position[3], const float intensity[3]);
    uint32_t add_spot_light(const float position[3], const float target[3], const float intensity[3], float cutoff_deg, float beam_deg);
    uint32_t add_distant_light(const float direction[3], const float irradiance[3], float scene_radius);
    uint32_t add_image(const uint32_t* texels, uint32_t w, uint32_t h, uint32_t texel_type, uint32_t wrap, uint32_t filter);
    uint32_t set_environment_map(uint32_t image, const float scale[3], const ctl_float4x4* to_world);
    void set_rough_transmittance(uint32_t slot, const ctl_rough_transmittance& t);
    void load_rough_transmittance(uint32_t slot, const char* path);
    void set_camera_lookat(const float pos[3], const float target[3], const float up[3], float fov_degrees, uint32_t w, uint32_t h);
    void set_camera(const ctl_sensor& s);
    void finalize(ctl_scene_desc& out);
};

} // namespace ctl
