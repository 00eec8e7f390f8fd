// device_scene.h — the scene as it lives in HBM (re-laid-out once at ctl_scene_create from the reference-layout
// arrays of a ctl_scene_desc; results are unchanged, see DESIGN.md "Data layout in HBM").
#pragma once
#include "../../include/ctl_amd.h"
#include "ctl_math.h"

namespace ctl {

constexpr int kExitMarker = 0x76543211;   // traversal-stack marker: leave the current instance (> kSentinel as unsigned)
// feature bits of the shade kernel builds (CTL_SHADE_FEATURES in shading.h)
enum { kShadeMoreBsdfs = 1, kShadeRoughBsdfs = 2, kShadeImageTextures = 4, kShadeMoreLights = 8, kShadeNestingBsdfs = 16, kShadeSurfaceMaps = 32, kShadeMoreMicrofacet = 64 };
constexpr int kStackSize = 96;            // two-level: top depth + bottom depth + markers; 4-wide flat tree: 3 * depth + 1 — both checked at upload

// precomputed PerspectiveSensor state (SceneTypes/Sensor.cu:76-96)
struct dev_sensor {
    float s2c[16];        // m_sampleToCamera, row-major 4x4 (projective: TransformPoint divides by w)
    float to_world[12];   // rows 0..2 of toWorld
    float inv_res[2];
    float dx[3], dy[3];   // m_dx, m_dy: camera-space offsets of the neighbouring pixels' rays (Sensor.cu:86-89, sampleRayDifferential)
    uint32_t type;        // CTL_SENSOR_PERSPECTIVE / THINLENS / ORTHOGRAPHIC / TELECENTRIC
    float aperture_radius, focus_distance, screen_scale_x;   // ThinLens / Telecentric
};
// the levels behind level 0 of an image (KernelMIPMap::m_uLevels, m_sOffsets; Engine/MIPMap_device.h:57-69): texel offsets relative to the image's level 0
struct dev_mip_levels { uint32_t levels; uint32_t offsets[15]; };

struct dev_scene {
    const float4* top_nodes;     // scene BVH: 4 x float4 per node, addressed in float4 units (reference encoding)
    const float4* bot_nodes;     // all mesh BVHs, same encoding; a mesh's root is at its node offset
    const float4* leaf_tris;     // 4 x float4 per leaf entry: Woop rows a,b,c + {index bits, 0, 0, 0}  (64 B, one fetch group)
    const float4* inst;          // 4 x float4 per node: inverse-transform rows 0..2 + {w33, nodeOff4, leafOff, triOff} (bits)
    const float4* flat_nodes;    // optional single-level world-space BVH over all instanced triangles (flatten.cpp), else nullptr
    const float4* flat_leaves;   // 4 x float4 per leaf entry: the mesh's object-space Woop rows + {globalTri << 1 | last, node, 0, 0}
    int flat_root;
    int flat_compact;            // Q4: the child links are implied by the layout (flat4_node::links), a step loads 48 of the node's 64 B
    int inst_w_one;              // every node's inverse transform has w33 == 1.0f exactly (x / 1 == x: the traversal skips the division)
    int flat_format;             // flat_format of flatten.h: 0 Q4 (64-B quantised 4-wide), 1 F4 (128-B fp32 4-wide), 2 F2 (64-B fp32 2-wide)
    int flat_leaf_keys;          // the device copy of the leaf entries carries the BSDF model of the entry's material (bsdf_type & 15) in bits 28..31 of its index word (tracer.hip; host arrays and the cache do not)
    unsigned char* hit_key_out;  // per launch (the tracer's copy of this struct): where a closest-hit traversal leaves that model (CTL_BSDF_*, all >= 1) per ray (0 = miss), for the shade kernel's regrouping; else nullptr
    int flat_top_cached;         // Q4 with implied links: the first this-many nodes of flat_nodes (the top of the tree, stored breadth-first) are kept in LDS by every traversal workgroup (traverse_flat.h kTopCache)
    const float4* inst_fwd;      // 3 x float4 per node: forward-transform rows 0..2 (fillDG)
    const float2* normal_lut;    // 512 x {sin, cos}: the 8+8-bit normal codec's angles (ctl_math.h uchar2_to_normal_lut)
    const uint4* tri_data;       // 2 x uint4 per triangle (TriangleData, 32 B)
    const uint4* node_info;      // per node {material_offset, light0, light1, n_lights}
    const ctl_material* mats;
    const ctl_light* lights;
    const unsigned char* anim;
    const ctl_mipmap* images;    // level-0 KernelMIPMap descriptors with device texel pointers (every image's further levels follow its level 0 in the pool)
    const dev_mip_levels* mip_levels;   // per image: the pyramid behind level 0 (first-hit texture filtering of the PathTracer plugin)
    const float* mip_weight_lut; // KernelMIPMap::m_weightLut[64] (EWA)
    const ctl_rough_transmittance* rough_transmittance;   // 3 tables with device pointers, or nullptr
    const float* rt_reduced;     // per rough-plastic material with constant alpha: its transmittance table reduced to 1-D in cos(theta) + the diffuse value (tracer.hip)
    int start_node;
    uint32_t n_nodes;
    uint32_t num_lights;
    uint32_t env_map_index;
    uint32_t n_lights_buf, n_anim_bytes, n_materials_probe;   // sizes of lights[] and the anim blob (shade kernels keep small ones in LDS, shade_kernel.inc)
    uint32_t alpha_maps;         // KernelDynamicScene::doAlphaMapping: some material carries an alpha map
    uint32_t shade_features;     // kShade* bits the scene needs (selects the shade-kernel build, kernels.hip launch_shade)
    uint32_t shade_models;       // bit m = some material of the scene has the BSDF model CTL_BSDF_* == m (which model-class launches a depth needs, kernels.hip launch_shade)
    float eps;                   // m_rayTraceEps
    uint32_t light_indices[CTL_MAX_NUM_LIGHTS];
    float light_cdf[CTL_MAX_NUM_LIGHTS];
    dev_sensor cam;
};

} // namespace ctl
