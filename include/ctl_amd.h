/*
 * ctl_amd.h — C-ABI drop-in boundary of the MI355X-native wavefront path tracer.
 *
 * The reference (hhergeth/CudaTracerLib) exposes its tracer plugins as C++ classes
 * (`Tracer<true>` subclasses, Kernel/Tracer.h:193-294) fed by a `KernelDynamicScene`
 * POD of device arrays (Engine/KernelDynamicScene.h:28-109).  This header is the
 * plain-C restatement of exactly that boundary for the wavefront path-tracing hot
 * path: the same arrays (bit-compatible element layouts, cited per struct), the same
 * tracer life-cycle (Resize / InitializeScene / DoPass / counters / parameters) and
 * the same `Image` accumulator (`PixelData`, Engine/Image.h:10-29).
 *
 * All pointers are plain host or device pointers, all sizes are element counts,
 * no C++ or torch types appear.  Every function returns 0 on success or a negative
 * ctl_status; ctl_last_error() gives the message the reference would have thrown
 * as std::runtime_error (Defines.cpp:15-29).
 *
 * The library is libctl_amd.so (cudatracerlib_amd/csrc).  It needs a gfx950 device
 * for every entry point that renders or intersects; those fail with CTL_ERR_NO_DEVICE
 * instead of falling back to any CPU path.
 */
#ifndef CTL_AMD_H
#define CTL_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum {
    CTL_OK = 0,
    CTL_ERR_INVALID = -1,      /* bad argument / bad handle                       */
    CTL_ERR_NO_DEVICE = -2,    /* no HIP device (product path never falls back)   */
    CTL_ERR_HIP = -3,          /* hipError_t != hipSuccess, message in last_error */
    CTL_ERR_OVERFLOW = -4,     /* ray queue overflow (DoubleRayBuffer.h:86-89)    */
    CTL_ERR_UNSUPPORTED = -5,  /* e.g. participating media, unknown plugin name   */
    CTL_ERR_IO = -6            /* scene loader: file / parse errors               */
} ctl_status;

const char* ctl_last_error(void);
const char* ctl_version(void);
/* number of visible HIP devices (0 on a CPU-only box; never an error) */
int ctl_device_count(void);

/* ------------------------------------------------ reference-layout elements */
/* Engine/TriIntersectorData.h:42-117 — Aila–Laine node, 4 x float4 = 64 B.
 * a = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y), b = (c1...), c = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
 * child >= 0: float4 index of an inner node (nodeIdx*4); child < 0: ~firstLeafEntry;
 * 0x76543210: no child (SplitBVHBuilder.cpp:163-203). */
typedef struct { float a[4], b[4], c[4]; int32_t child0, child1; uint32_t parent; uint32_t pad; } ctl_bvh_node;
/* Engine/TriIntersectorData.h:30-40 — Woop unit-triangle rows, 48 B. */
typedef struct { float a[4], b[4], c[4]; } ctl_woop_tri;
/* Engine/TriIntersectorData.h:8-28 — (triangleIndex << 1) | lastInLeaf. */
typedef struct { uint32_t index; } ctl_woop_index;
/* Engine/TriangleData.h:10-54 (EXT_TRI, NUM_UV_SETS 1) — 32 B shading triangle. */
typedef struct { uint32_t nor_mat_extra[2]; uint32_t dpdu_dpdv[3]; uint32_t uv[3]; } ctl_triangle_data;
/* Engine/Mesh.h:12-19 */
typedef struct { uint32_t tri_offset, bvh_node_offset, bvh_tri_offset, bvh_index_offset, std_material_offset; } ctl_kernel_mesh;
/* SceneTypes/Node.h:13-24 (FixedSizeArray<unsigned,2>: buffer then length) — 24 B */
typedef struct { uint32_t mesh_index, material_offset, instanciated_material; uint32_t lights[2]; uint32_t n_lights; } ctl_node;
/* Math/float4x4.h:12-18 — row-major, column-vector convention */
typedef struct { float m[16]; } ctl_float4x4;
/* Kernel/TraceHelper.h:55-59 — a = (origin, tmin), b = (direction, tmax) */
typedef struct { float a[4]; float b[4]; } ctl_ray;
/* Kernel/TraceHelper.h:61-69 — dist, node, triangle, barycentrics.
 * The reference packs the barycentrics into 2 x u16 (TraceHelper.cu:728-729); this build
 * keeps them as full floats (the precision of the reference's single-ray traceRay,
 * TraceHelper.cu:159) and so the record is 20 B wide on the C-ABI. */
typedef struct { float dist; int32_t node_idx; int32_t tri_idx; float u, v; } ctl_hit;
/* Engine/Image.h:10-29 — 28 B accumulator */
typedef struct { float rgb[3]; float rgb_splat[3]; float weight_sum; } ctl_pixel_data;
/* Engine/ShapeSet.h:19-30 — area-light triangle, 64 B, lives in the anim blob */
typedef struct { float p[3][3]; float n[3]; float area; uint32_t i_dat; uint32_t t_dat; uint32_t pad; } ctl_shape_tri;   /* CUDA_ALIGN(16): 64 B */

/* --------------------------------------------- compact scene-type descriptors */
/* The reference stores BSDFs / lights / sensors as tagged C++ unions (Material 3344 B,
 * Light 592 B, Sensor 320 B) whose byte layout depends on the host compiler.  The
 * boundary carries the same parameters as flat descriptors; INTEGRATION.md shows the
 * conversion a maintainer adds next to DynamicScene::getKernelSceneData(). */
/* ids = TYPE_FUNC ids of SceneTypes/Texture.h:107,127,159 */
enum { CTL_TEX_CONSTANT = 2, CTL_TEX_CHECKER = 3, CTL_TEX_IMAGE = 4 };
typedef struct {
    uint32_t type;       /* CTL_TEX_*                                          */
    float value[3];      /* constant colour / checker colour 0                  */
    float value1[3];     /* checker colour 1                                    */
    float uv_scale[2], uv_offset[2];
    uint32_t image;      /* index into the scene's image table (CTL_TEX_IMAGE)  */
} ctl_texture;           /* 48 B */

/* CTL_TEX_IMAGE (ImageTexture, SceneTypes/Texture.h:159-183, Texture.cu:6-13): value = m_scale, uv_scale/uv_offset = the
 * diagonal TextureMapping2D (m11, m22, m13, m23), image = index into ctl_scene_desc::images. */

/* Engine/MIPMap_device.h:11-32,57-69 — level 0 of a KernelMIPMap.  The path samples level 0 only: ImageTexture::Evaluate
 * without uv partials -> Sample(uv) = Texel(0,uv) / triangle(0,uv) (MIPMap.cu:116-121), InfiniteLight -> Sample(uv, 0) =
 * triangle(0,uv) and Sample(0,x,y) (MIPMap.cu:139-165). */
enum { CTL_WRAP_REPEAT = 0, CTL_WRAP_CLAMP = 1, CTL_WRAP_MIRROR = 2, CTL_WRAP_BLACK = 3 };
enum { CTL_FILTER_POINT = 0, CTL_FILTER_BILINEAR = 1, CTL_FILTER_ANISOTROPIC = 2, CTL_FILTER_TRILINEAR = 3 };
enum { CTL_TEXEL_RGBE = 0, CTL_TEXEL_RGBCOL = 1 };   /* Texture_DataType; both are uchar4 (Math/Spectrum.h:323-324,528-565) */
typedef struct {
    const uint32_t* texels;    /* width*height level-0 texels, row-major, byte 0 = r (x), 1 = g, 2 = b, 3 = e / alpha */
    uint32_t width, height;
    uint32_t texel_type, wrap_mode, filter_mode;
} ctl_mipmap;

/* Engine/RoughTransmittance.h:9-27 — one table of Mitsuba's data/microfacet/{beckmann,phong,ggx}.dat (roughplastic).
 * trans[2*eta_samples][alpha_samples][theta_samples], diff_trans[2*eta_samples][alpha_samples] (the second half of each is
 * the eta < 1 block).  ctl_scene_desc::rough_transmittance points at THREE of them, indexed by the microfacet distribution
 * type exactly as RoughTransmittanceManager does (RoughTransmittance.cu:124-157: slot 0 beckmann.dat, 1 phong.dat, 2 ggx.dat). */
typedef struct {
    const float* trans; const float* diff_trans;
    uint32_t eta_samples, alpha_samples, theta_samples;
    float eta_min, eta_max, alpha_min, alpha_max;
} ctl_rough_transmittance;

/* ids = the reference's TYPE_FUNC ids (BSDF_Simple.h:6-401, BSDF_Complex.h:9-182) */
enum { CTL_BSDF_DIFFUSE = 1, CTL_BSDF_ROUGHDIFFUSE = 2, CTL_BSDF_DIELECTRIC = 3, CTL_BSDF_THINDIELECTRIC = 4,
       CTL_BSDF_ROUGHDIELECTRIC = 5, CTL_BSDF_CONDUCTOR = 6, CTL_BSDF_ROUGHCONDUCTOR = 7, CTL_BSDF_PLASTIC = 8,
       CTL_BSDF_ROUGHPLASTIC = 9, CTL_BSDF_PHONG = 10, CTL_BSDF_WARD = 11, CTL_BSDF_HK = 12,
       CTL_BSDF_COATING = 13, CTL_BSDF_ROUGHCOATING = 14, CTL_BSDF_BLEND = 15 };
/* SceneTypes/Samples.h:31-95 lobe flags */
enum { CTL_ENull = 0x1, CTL_EDiffuseReflection = 0x2, CTL_EDiffuseTransmission = 0x4, CTL_EGlossyReflection = 0x8,
       CTL_EGlossyTransmission = 0x10, CTL_EDeltaReflection = 0x20, CTL_EDeltaTransmission = 0x40,
       CTL_EDelta1DReflection = 0x80, CTL_EDelta1DTransmission = 0x100 };
/* Engine/MicrofacetDistribution.h:14-21 */
enum { CTL_MF_BECKMANN = 0, CTL_MF_GGX = 1, CTL_MF_PHONG = 2 };

/* Parameter slots by bsdf_type:
 *  diffuse        tex0 reflectance
 *  roughdiffuse   tex0 reflectance, tex1 alpha, u0 useFastApprox
 *  dielectric     tex0 specularTransmittance, tex1 specularReflectance, f0 Cauchy B (= eta), f1 Cauchy C
 *  thindielectric tex0 specularTransmittance, tex1 specularReflectance, f0 eta
 *  roughdielectric tex0 specT, tex1 specR, tex2 alphaU, tex3 alphaV, f0 eta, f1 invEta, u0 distribution, u1 sampleVisible
 *  conductor      tex0 specularReflectance, f0..2 eta rgb, f3..5 k rgb
 *  roughconductor tex0 specR, tex1 alphaU, tex2 alphaV, f0..2 eta, f3..5 k, u0 distribution, u1 sampleVisible
 *  plastic        tex0 diffuseReflectance, tex1 specularReflectance, f0 fdrInt, f1 fdrExt, f2 eta, f3 invEta2,
 *                 f4 specularSamplingWeight, u0 nonlinear
 *  roughplastic   tex0 diffuse, tex1 specular, tex2 alpha, f0 eta, f1 invEta2, f2 specularSamplingWeight,
 *                 u0 nonlinear, u1 sampleVisible, u2 distribution
 *  phong          tex0 diffuse, tex1 specular, tex2 exponent, f0 specularSamplingWeight
 *  coating        tex0 sigmaA, tex1 specularReflectance, f0 eta, f1 invEta, f2 thickness, f3 specularSamplingWeight, u2 nested material
 *  roughcoating   tex0 sigmaA, tex1 specularReflectance, tex2 alpha, f0 eta, f1 invEta, f2 thickness, f3 specularSamplingWeight,
 *                 u0 distribution, u1 sampleVisible, u2 nested material
 *  blend          tex0 weight, u2 / u3 nested materials (absolute indices, ctl_builder_add_material); combined_type of the three =
 *                 own lobes | the nested BSDFs' combined_type
 *  ward           tex0 diffuse, tex1 specular, tex2 alphaU, tex3 alphaV, f0 specularSamplingWeight, u0 variant (0 Ward, 1 Ward-Duer, 2 balanced) */
typedef struct {
    uint32_t bsdf_type;        /* CTL_BSDF_*                                       */
    uint32_t combined_type;    /* BSDF::m_combinedType (SceneTypes/BSDF.h:24)      */
    uint32_t two_sided;        /* BSDF::m_enableTwoSided (SceneTypes/BSDF.h:26)    */
    uint32_t node_light_index; /* Material::NodeLightIndex, UINT32_MAX = none      */
    ctl_texture tex[4];
    float f[8];
    uint32_t u[4];
    /* Material::NormalMap / HeightMap (Engine/Material.h:58-59,93-106: at most one of them) and Material::AlphaMap
     * (Material.h:13-36,60,107-111).  The normal / height map perturbs the shading frame in TraceResult::getBsdfSample
     * (Material::SampleNormalMap, Material.cu:96-138; parallax occlusion is never enabled by the reference and is not carried).
     * The alpha test runs inside single-ray traversal (TraceHelper.cu:135-153) — i.e. for the megakernel PathTracer; the
     * reference's wavefront intersectKernel has no alpha test, see the AlphaTest tracer parameter. */
    uint32_t map_kind;         /* CTL_MAP_NONE / CTL_MAP_NORMAL / CTL_MAP_HEIGHT                 */
    uint32_t alpha_state;      /* AlphaBlendState, CTL_ALPHA_*                                   */
    float alpha_test_scalar;   /* AlphaBlendData::test_val_scalar                                */
    float alpha_test_color[3]; /* AlphaBlendData::test_val_color                                 */
    uint32_t reserved_[2];
    ctl_texture map_tex;       /* the normal or height map                                      */
    ctl_texture alpha_tex;     /* AlphaBlendData::tex                                            */
} ctl_material;                /* 384 B */
enum { CTL_MAP_NONE = 0, CTL_MAP_NORMAL = 1, CTL_MAP_HEIGHT = 2 };
/* Material.h:13-22: low two bits = test (1 luminance >= scalar, 2 alpha channel >= scalar, 3 |colour - test colour| <= scalar),
 * bit 2 = take the BSDF's first texture instead of alpha_tex */
enum { CTL_ALPHA_DISABLED = 0, CTL_ALPHA_MAP_LUMINANCE = 1, CTL_ALPHA_MAP_ALPHA = 2, CTL_ALPHA_MAP_COLOR = 3,
       CTL_ALPHA_REFLECTANCE_LUMINANCE = 5, CTL_ALPHA_REFLECTANCE_ALPHA = 6, CTL_ALPHA_REFLECTANCE_COLOR = 7 };

/* ids = TYPE_FUNC ids of SceneTypes/Light.h:36,98,147,228,296 */
enum { CTL_LIGHT_POINT = 1, CTL_LIGHT_DIFFUSE = 2, CTL_LIGHT_DISTANT = 3, CTL_LIGHT_SPOT = 4, CTL_LIGHT_INFINITE = 5 };
typedef struct {
    uint32_t type;                 /* CTL_LIGHT_*                                              */
    float radiance[3];             /* DiffuseLight::m_rad_texture (constant) / intensity       */
    /* DiffuseLight: ShapeSet (Engine/ShapeSet.h:61-66) — byte offsets into the anim blob */
    uint32_t area_dist_index;      /* float[count+1] normalised area CDF                       */
    uint32_t triangles_index;      /* ctl_shape_tri[count]                                     */
    float sum_area;
    uint32_t count;
    uint32_t orthogonal;           /* DiffuseLight::m_bOrthogonal                              */
    uint32_t node_idx;             /* DiffuseLight::m_uNodeIdx                                 */
    /* Point / Spot / Distant */
    float position[3];
    float direction[3];
    float cutoff_angle, beam_width, cos_cutoff_angle, cos_beam_width, inv_transition_width;
    float to_world[16];            /* Spot / Distant: Frame ToWorld as rows s = [0..2], t = [4..6], n = [8..10];
                                      Infinite: m_worldTransform, row-major 4x4                */
    /* InfiniteLight (SceneTypes/Light.h:293-311, Light.cpp:10-61); the three tables are byte offsets into the anim blob */
    uint32_t env_image;            /* radianceMap: index into ctl_scene_desc::images           */
    float env_scale[3];            /* m_scale                                                  */
    float bsphere_center[3], bsphere_radius;   /* m_SceneCenter, m_SceneRadius; DistantLight::radius */
    uint32_t cdf_rows_index;       /* float[height + 1]                                        */
    uint32_t cdf_cols_index;       /* float[height * (width + 1)]                              */
    uint32_t row_weights_index;    /* float[height]                                            */
    float normalization;           /* m_normalization                                          */
    /* DiffuseLight::m_rad_texture when it is not a ConstantTexture (SceneTypes/Light.cu:50-53 needsUVSample): CTL_TEX_CHECKER / CTL_TEX_IMAGE;
     * any other type (0, CTL_TEX_CONSTANT) = the constant `radiance` above */
    ctl_texture rad_texture;
} ctl_light;

/* ids = TYPE_FUNC ids of SceneTypes/Sensor.h:107,191,272,364,445 */
enum { CTL_SENSOR_SPHERICAL = 1, CTL_SENSOR_PERSPECTIVE = 2, CTL_SENSOR_THINLENS = 3, CTL_SENSOR_ORTHOGRAPHIC = 4, CTL_SENSOR_TELECENTRIC = 5 };
typedef struct {
    uint32_t type;
    float to_world[16];            /* SensorBase::toWorld (row-major)                          */
    float fov;                     /* radians, horizontal (SceneTypes/Sensor.h:72-76)          */
    float near_depth, far_depth;   /* SensorBase::m_fNearFarDepths                             */
    float resolution[2];           /* film size in pixels                                      */
    float aperture_radius, focus_distance;   /* ThinLens / Telecentric: SensorBase::m_apertureRadius, m_focusDistance */
    float screen_scale[2];         /* Telecentric: screenScale (its aperture is aperture_radius / screen_scale[0]); 0 = 1 */
} ctl_sensor;

#define CTL_MAX_NUM_LIGHTS 16      /* Engine/KernelDynamicScene.h:26 */

/* Engine/KernelDynamicScene.h:28-109 restated with plain pointers (HOST memory). */
typedef struct {
    const ctl_triangle_data* tri_data;   uint32_t n_tri_data;     /* m_sTriData       */
    const ctl_woop_tri* woop;            uint32_t n_woop;         /* m_sBVHIntData    */
    const ctl_woop_index* woop_index;                             /* m_sBVHIndexData (n_woop) */
    const ctl_bvh_node* bvh_nodes;       uint32_t n_bvh_nodes;    /* m_sBVHNodeData   */
    const ctl_kernel_mesh* meshes;       uint32_t n_meshes;       /* m_sMeshData      */
    const ctl_node* nodes;               uint32_t n_nodes;        /* m_sNodeData      */
    const ctl_material* materials;       uint32_t n_materials;    /* m_sMatData       */
    const ctl_light* lights;             uint32_t n_lights_buf;   /* m_sLightBuf      */
    const uint8_t* anim;                 uint32_t n_anim_bytes;   /* m_sAnimData      */
    /* KernelSceneBVH (Engine/SceneBVH_device.h:8-15) */
    int32_t scene_start_node;
    const ctl_bvh_node* scene_bvh_nodes; uint32_t n_scene_bvh_nodes;
    const ctl_float4x4* node_transforms;      /* n_nodes */
    const ctl_float4x4* node_inv_transforms;  /* n_nodes */
    uint32_t env_map_index;                   /* UINT32_MAX = none */
    float box_min[3], box_max[3];             /* m_sBox */
    ctl_sensor camera;                        /* m_Camera */
    uint32_t num_lights;                      /* m_numLights */
    uint32_t light_indices[CTL_MAX_NUM_LIGHTS];
    float light_cdf[CTL_MAX_NUM_LIGHTS];
    float ray_trace_eps;                      /* m_rayTraceEps = 1e-4 * |box diagonal| (DynamicScene.cpp:587) */
    const ctl_mipmap* images;            uint32_t n_images;       /* m_sTexData (level 0) */
    const ctl_rough_transmittance* rough_transmittance;           /* [3] or NULL (needed by roughplastic only) */
} ctl_scene_desc;

/* BSDF::Update() (SceneTypes/BSDF_Simple.h:255-264 plastic, :298-304 roughplastic, :332-337 phong, :371-376 ward; BSDF_Complex.h:37-44 coating, :117-125
 * roughcoating): recomputes a material's DERIVED fields from its primary ones — fdrInt / fdrExt by the reference's adaptive Gauss-Lobatto quadrature
 * (FresnelHelper::fresnelDiffuseReflectance, Math/FresnelHelper.cu:57-60), invEta / invEta2, the specular sampling weights from the textures' average
 * luminance (constant and checkerboard textures; an image texture's average is the caller's, as ImageTexture::Average reads the coarsest MIP level).
 * Host only.  A caller converting the reference's own objects copies their fields instead and does not need this. */
int ctl_material_update(ctl_material* m);
/* FresnelHelper::fresnelDiffuseReflectance(eta, false) as above. */
float ctl_fresnel_diffuse_reflectance(float eta);

/* ------------------------------------------------------------- scene builder */
/* Host-side mirror of the subset of DynamicScene the loader drives
 * (Engine/DynamicScene.h:70-187; SURVEY §8b "Loader boundary"). */
typedef struct ctl_builder ctl_builder;
int ctl_builder_create(ctl_builder** out);
void ctl_builder_destroy(ctl_builder* b);
/* Mesh::CompileMesh (Engine/Mesh.cpp:199-290): positions[3*n_vert], indices[3*n_tri] (NULL = soup),
 * normals[3*n_vert] or NULL (computed as Mesh.cpp:151-190), uvs[2*n_vert] or NULL, tri_material[n_tri] local
 * material index or NULL (all 0), materials[n_mat]. Builds the mesh BVH (SAH, max leaf 8 as
 * BVHBuilderHelper.cpp:119).  Returns the mesh index. */
int ctl_builder_add_mesh(ctl_builder* b, const float* positions, uint32_t n_vert, const uint32_t* indices, uint32_t n_tri,
                         const float* normals, const float* uvs, const uint8_t* tri_material,
                         const ctl_material* materials, uint32_t n_mat, uint32_t* mesh_index_out);
/* Which builder ctl_builder_add_mesh uses from now on.  CTL_BVH_SBVH: the reference's SplitBVHBuilder restated
 * (Engine/SpatialStructures/BVH/SplitBVHBuilder.cpp:219-640 through ConstructBVH) — the same node / Woop / index arrays element for
 * element.  CTL_BVH_BINNED: binned SAH, object splits only, threaded.  CTL_BVH_AUTO (default; or $CTL_BVH_MODE = sbvh | binned):
 * SBVH for meshes of up to 65536 triangles, binned above. */
enum { CTL_BVH_AUTO = 0, CTL_BVH_SBVH = 1, CTL_BVH_BINNED = 2 };
int ctl_builder_set_bvh_mode(ctl_builder* b, uint32_t mode);
/* DynamicScene::CreateNode + SetNodeTransform (DynamicScene.cpp:269-346): one instance of a mesh. */
int ctl_builder_add_node(ctl_builder* b, uint32_t mesh_index, const ctl_float4x4* to_world, uint32_t* node_index_out);
/* DynamicScene::CreateLight(node, matName, L) (DynamicScene.cpp:689-711): all triangles of the node whose local
 * material index is `local_material` become one DiffuseLight. */
int ctl_builder_add_area_light(ctl_builder* b, uint32_t node_index, uint32_t local_material, const float radiance[3]);
/* The same with the two DiffuseLight members an application sets on the light afterwards (SceneTypes/Light.h:100-101): a radiance texture that is
 * not constant (NULL = constant `radiance`) and m_bOrthogonal (the light emits along its surface normal only). */
int ctl_builder_add_area_light_ex(ctl_builder* b, uint32_t node_index, uint32_t local_material, const float radiance[3], const ctl_texture* rad_texture, int32_t orthogonal);
/* DynamicScene::CreateLight(Light) for point lights (SceneTypes/Light.h:31-94) */
int ctl_builder_add_point_light(ctl_builder* b, const float position[3], const float intensity[3]);
/* SpotLight(p, t, L, width, fall) (SceneTypes/Light.cu:268-277): cutoff/beam angles in degrees as the loader passes them
 * (cutoffAngle, beamWidth). */
int ctl_builder_add_spot_light(ctl_builder* b, const float position[3], const float target[3], const float intensity[3],
                               float cutoff_angle_degrees, float beam_width_degrees);
/* DistantLight(L, d, r) (SceneTypes/Light.h:155-163): d = ToWorld.n (sampleDirect returns dRec.d = -d); r = radius of the
 * scene's bounding sphere as the caller sees it (the light stores 1.1 r; the Mitsuba loader passes r = 1, ObjectParser.h:530). */
int ctl_builder_add_distant_light(ctl_builder* b, const float direction[3], const float irradiance[3], float scene_radius);
/* MIPMap level 0 handed over decoded (the reference decodes files with FreeImage, Engine/MIPMap.cpp): returns the image index */
int ctl_builder_add_image(ctl_builder* b, const uint32_t* texels, uint32_t width, uint32_t height, uint32_t texel_type,
                          uint32_t wrap_mode, uint32_t filter_mode, uint32_t* image_index_out);
/* DynamicScene::setEnvironementMap(scale, file) (Engine/DynamicScene.cpp:846-859) + InfiniteLight ctor (Light.cpp:10-61):
 * builds the row / column CDFs; to_world (row-major 4x4, orthogonal) may be NULL = identity. */
int ctl_builder_set_environment_map(ctl_builder* b, uint32_t image_index, const float scale[3], const ctl_float4x4* to_world);
/* Registers a BSDF that no triangle refers to directly: the nested BSDF (BSDFFirst, SceneTypes/BSDF.h:102) of a coating,
 * roughcoating or blend.  Returns its absolute index in ctl_scene_desc::materials, to be stored in the parent's u[2] (/u[3]). */
int ctl_builder_add_material(ctl_builder* b, const ctl_material* material, uint32_t* index_out);
/* RoughTransmittanceManager::StaticInitialize (Engine/RoughTransmittance.cu:124-131): install the table of one slot (0..2);
 * the arrays are copied.  ctl_builder_load_rough_transmittance parses a Mitsuba "MTS_TRANSMITTANCE" .dat file (:8-45). */
int ctl_builder_set_rough_transmittance(ctl_builder* b, uint32_t slot, const ctl_rough_transmittance* table);
int ctl_builder_load_rough_transmittance(ctl_builder* b, uint32_t slot, const char* dat_path);
/* DynamicScene::setCamera — perspective sensor as built by the Mitsuba loader (ObjectParser.h:292-297): */
int ctl_builder_set_camera_lookat(ctl_builder* b, const float pos[3], const float target[3], const float up[3],
                                  float fov_degrees, uint32_t width, uint32_t height);
int ctl_builder_set_camera(ctl_builder* b, const ctl_sensor* sensor);
/* DynamicScene::UpdateScene + getKernelSceneData (DynamicScene.cpp:480-589): builds the scene BVH and fills
 * `out` with pointers that stay valid until the builder is destroyed or finalized again. */
int ctl_builder_finalize(ctl_builder* b, ctl_scene_desc* out);

/* ------------------------------------------------------------------- scenes */
typedef struct ctl_scene ctl_scene;
/* UpdateKernel(scene) (Kernel/TraceHelper.cu:182-217): uploads + re-lays-out the arrays in HBM. */
int ctl_scene_create(const ctl_scene_desc* desc, ctl_scene** out);
/* flags: CTL_SCENE_FLATTEN = additionally build ONE world-space BVH over all instanced triangles (128 B of HBM per instanced
 * triangle) and traverse that.  The flattened tree only culls: every leaf entry is evaluated with the reference's instance
 * transform + object-space Woop arithmetic (Kernel/TraceHelper.cu:526-560,646-682), so t,u,v,triangle,node equal the two-level
 * traversal bit for bit.  A triangle much longer than its neighbours is entered under several leaf entries (early split
 * clipping, DESIGN.md §3: the same 128 B under the boxes of its parts; a hit is the whole triangle's), so the tree may hold more entries
 * than the scene has instanced triangles.  CTL_SCENE_FLAT_FORMAT(f) picks the node format (measurement; default Q4). */
enum { CTL_SCENE_FLATTEN = 1,
       /* opt-in: a rough plastic with a constant roughness looks RoughTransmittanceManager's table up through a per-material 1-D reduction (4 taps instead of 64; synthetic-bathroom
        * + 3 % rays/s).  Equal to the reference's 3-D lookup (Engine/RoughTransmittance.cu:55-88) up to fp32 rounding only: frames stay within the per-pixel tolerance in most
        * scenes but are no longer equal to the bit, and textured scenes can exceed it at texture boundaries.  Without the flag every lookup is the reference's own arithmetic. */
       CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE = 2 };
enum { CTL_FLAT_Q4 = 0,    /* 4-wide, 64-B nodes, 8-bit child boxes (default) */
       CTL_FLAT_F4 = 1,    /* 4-wide, 128-B nodes, fp32 child boxes           */
       CTL_FLAT_F2 = 2,    /* 2-wide, 64-B nodes in the reference's BVHNodeData layout */
       CTL_FLAT_Q8 = 3 };  /* 8-wide, 128-B nodes, 8-bit child boxes, octant-ordered slots, one-triangle leaf slots (csrc/flat8.h) */
#define CTL_SCENE_FLAT_FORMAT(f) ((((uint32_t)(f)) + 1u) << 8)
int ctl_scene_create_ex(const ctl_scene_desc* desc, uint32_t flags, ctl_scene** out);
void ctl_scene_destroy(ctl_scene* s);
/* On-disk cache of compiled geometry — the role of the reference's .xmsh files (Engine/Mesh.cpp:46-98,199-290; DynamicScene::CreateNode
 * compiles a mesh only when its .xmsh is missing).  With a directory set, ctl_builder_add_mesh stores / reloads a compiled mesh
 * (TriangleData, BVH nodes, Woop rows) and CTL_SCENE_FLATTEN stores / reloads the flattened BVH, both keyed by a hash of their
 * inputs.  NULL or "" disables; the default comes from the environment variable CTL_CACHE_DIR.  Host only. */
int ctl_set_cache_dir(const char* dir);
/* The host half of CTL_SCENE_FLATTEN without a device (tests, cache warming): builds (or loads) the flattened BVH of `desc` in
 * node format `format` (CTL_FLAT_*) and reports out4 = { inner nodes, leaf entries, depth of the tree, low 64 bits of a hash of the arrays }. */
int ctl_flatten_probe(const ctl_scene_desc* desc, uint32_t format, uint64_t* out4);
/* The same, handing out the arrays (host memory owned by the handle): what ctl_scene_create_ex uploads.  The test oracle
 * traverses these very arrays on the CPU (counts "with the same BVH").  Layouts: cudatracerlib_amd/csrc/flatten.h. */
typedef struct ctl_flat_bvh ctl_flat_bvh;
typedef struct {
    uint32_t format, max_depth;      /* CTL_FLAT_*, depth of the stored tree                                      */
    const void* nodes; uint64_t n_nodes; uint32_t node_bytes;   /* node 0 is the root; child >= 0: node index * node_bytes / 16 */
    const void* leaves; uint64_t n_leaves;   /* 128 B each: object-space Woop rows a,b,c, {globalTri << 1 | last, node, 0, 0}, rows 0..2 of the node's inverse transform, {w33,0,0,0} */
    const int32_t* child_links;      /* CTL_FLAT_Q4: 4 explicit links per node (>= 0: node index * 4, < 0: ~first leaf entry, 0x76543210: none); CTL_FLAT_Q8: 8 per node, slot order (>= 0: node index); else NULL */
    uint32_t compact;                /* CTL_FLAT_Q4: 1 = the kernels derive the links from the layout (flat4_node::links) and the nodes' last 16 B hold oriented slabs; CTL_FLAT_Q8: always 1 */
    uint32_t root_slab;              /* the root node carries a slab (Q4: bit 0 of the link a traversal starts with; Q8: the root's q5 is loaded)            */
    uint64_t n_slab_nodes;           /* nodes that carry an oriented slab (flat_slab.h)                                                  */
} ctl_flat_bvh_desc;
int ctl_flat_bvh_build(const ctl_scene_desc* desc, uint32_t format, ctl_flat_bvh** out);
int ctl_flat_bvh_arrays(const ctl_flat_bvh* h, ctl_flat_bvh_desc* out);
void ctl_flat_bvh_destroy(ctl_flat_bvh* h);
/* ParseMitsubaScene (Engine/SceneLoader/Mitsuba/MitsubaLoader.h:13): fills a builder from a Mitsuba-0.5 XML file. */
int ctl_parse_mitsuba_scene(ctl_builder* b, const char* xml_path, int32_t* width_inout, int32_t* height_inout);
/* The bitmap reader behind the loader's textures and environment maps (the reference goes through FreeImage, Engine/MIPMap.cu:542-592): PNG, JPEG
 * (sequential and progressive), BMP, TGA, PNM, PFM, Radiance HDR, OpenEXR (scanline; NONE / RLE / ZIPS / ZIP).  Rows top-down.  *is_float = 1: `rgb`
 * receives 3 floats per pixel; 0: `rgba8` receives 4 bytes per pixel.  With both buffers NULL only the size and kind are returned. */
int ctl_decode_image_file(const char* path, uint32_t* width, uint32_t* height, int32_t* is_float, float* rgb, uint8_t* rgba8);

/* ------------------------------------------------------------------ sampler */
/* SequenceSamplerData(4096, 30) (Kernel/Sampler_device.h:11-57). tables_1d[30*4096], tables_2d[30*4096*2]:
 * element (seq, e) at e*4096+seq. */
#define CTL_SAMPLER_NUM_SEQUENCES 4096
#define CTL_SAMPLER_SEQUENCE_LENGTH 30
typedef struct ctl_sequence_generator ctl_sequence_generator;
/* IndependantSamplingSequenceGenerator (Kernel/Sampler.h:57-85): XORWOW curand_init(1234, 7539414, 0). */
int ctl_sequence_generator_create(ctl_sequence_generator** out);
void ctl_sequence_generator_destroy(ctl_sequence_generator* g);
/* SamplingSequenceGeneratorHost::Compute (Kernel/Sampler.h:36-55): next pass's tables into host buffers. */
int ctl_sequence_generator_compute(ctl_sequence_generator* g, float* tables_1d, float* tables_2d);
/* the tables of the next n_passes passes (consecutive [30*4096] / [30*4096*2] blocks), generated by up to `threads` host
 * threads; same values as n_passes calls of ctl_sequence_generator_compute (XORWOW skip-ahead) */
int ctl_sequence_generator_compute_many(ctl_sequence_generator* g, uint32_t n_passes, float* tables_1d, float* tables_2d, uint32_t threads);
/* The same tables written in HBM by the kernel the tracers use (k_sequence_fill: the host only advances the stream) and copied back: bit-identical to
 * ctl_sequence_generator_compute_many.  Needs a HIP device. */
int ctl_sequence_generator_compute_many_device(ctl_sequence_generator* g, uint32_t n_passes, float* tables_1d, float* tables_2d);

/* -------------------------------------------------------------------- image */
typedef struct ctl_image ctl_image;
int ctl_image_create(uint32_t width, uint32_t height, ctl_image** out);   /* Image::Image (Engine/Image.h:34) */
void ctl_image_destroy(ctl_image* img);
int ctl_image_clear(ctl_image* img);                                        /* Image::Clear                    */
int ctl_image_read_pixels(ctl_image* img, ctl_pixel_data* host_out);        /* D2H of the PixelData array      */
int ctl_image_write_pixels(ctl_image* img, const ctl_pixel_data* host_in);
/* Image::AddSample (Engine/Image.h:56, Image.cu:22-44) for `n` samples {sx, sy, r, g, b} in host memory: negatives clamped, a NaN / infinite radiance or a position outside the
 * film dropped, the pixel floor(sx), floor(sy) receives rgb += L, weightSum += 1 (float atomics, as the reference's device branch).  What a plugin that does not use the
 * tracer's own accumulation calls to deposit its radiance. */
int ctl_image_add_samples(ctl_image* img, uint32_t n, const float* host_samples5);
void* ctl_image_device_ptr(ctl_image* img);                                 /* PixelData* in HBM (RCCL gather) */
/* copySamplesToOutput (Kernel/ImagePipeline/ImagePipeline.cu:14-30): rgb/weight + splat*scale -> linear RGB float */
int ctl_image_resolve_rgb(ctl_image* img, float splat_scale, float* host_rgb_out);
/* applyImagePipeline(tracer, img, filter = 0, process = 0) (Kernel/ImagePipeline/ImagePipeline.cu:54-63): the display image,
 * sRGB transfer curve + 8-bit RGBCOL per pixel (byte 0 = r, alpha = 255).  splat_scale = TracerBase::getSplatScale() = 1/passes. */
int ctl_image_apply_pipeline(ctl_image* img, float splat_scale, uint32_t* host_rgbcol_out);
/* The other three branches of applyImagePipeline (ImagePipeline.cu:64-81): an ImageSamplesFilter and / or a PostProcess.
 * filter  = CanonicalFilter over a reconstruction Filter (Kernel/ImagePipeline/Filter/CanonicalFilter.cu:6-44, SceneTypes/Filter.h:
 *           ids = TYPE_FUNC ids; p0 / p1 = Gaussian alpha | Mitchell B, C | Lanczos tau); the filtered image is kept as RGBE.
 * process = ToneMapPostProcess, Reinhard et al. (PostProcess/ToneMapPostProcess.cu:6-42) driven by Image::ComputeLuminanceInfo
 *           (Engine/Image.cu:88-168) of the RGBE image; the result is quantised to RGBCOL and then gamma-corrected, as there.
 * Either may be NULL (both NULL = ctl_image_apply_pipeline).  The NonLocalMeans filter is not part of this build. */
enum { CTL_RFILTER_BOX = 1, CTL_RFILTER_GAUSSIAN = 2, CTL_RFILTER_MITCHELL = 3, CTL_RFILTER_LANCZOS = 4, CTL_RFILTER_TRIANGLE = 5 };
typedef struct { uint32_t type; float x_width, y_width, p0, p1; } ctl_reconstruction_filter;
typedef struct { float key, burn; } ctl_tonemap;           /* defaults of the reference: key 0.18, burn 0 */
int ctl_image_apply_pipeline_ex(ctl_image* img, float splat_scale, const ctl_reconstruction_filter* filter, const ctl_tonemap* process,
                                uint32_t* host_rgbcol_out);
/* Image::WriteDisplayImage (Engine/Image.cpp:67-75): .png writes the display image, .hdr / .pfm the linear float image.
 * Other formats (the reference saves through FreeImage) -> CTL_ERR_UNSUPPORTED. */
int ctl_image_write_file(ctl_image* img, float splat_scale, const char* path);

/* ---------------------------------------------------------------- multi-GPU */
/* The one exchange step of a multi-GPU render (the reference is single-device): rank r renders the 64x64 image tiles t with
 * t % world == r (ctl_tracer_set_tile_shard) into a cleared full-size Image.  ctl_image_gather brings every rank's own tiles to `root`'s image
 * with ONE ncclGather over RCCL / xGMI; ctl_image_reduce (the fallback) sums the whole PixelData frames with ONE ncclReduce — the tiles are
 * disjoint, so the sum is the gather, at 8x the bytes.  One rank per process and GPU: call ctl_set_device first.  ctl_comm_get_unique_id on one rank, hand the 128 bytes to the others by any means (MPI_Bcast, a file,
 * torch.distributed), then ctl_comm_create on every rank (collective).  RCCL is loaded on first use. */
typedef struct ctl_comm ctl_comm;
int ctl_comm_get_unique_id(uint8_t out128[128]);
int ctl_comm_create(const uint8_t id128[128], int32_t rank, int32_t world, ctl_comm** out);
void ctl_comm_destroy(ctl_comm* c);
/* ctl_comm_create with a deadline: ncclCommInitRank is collective and waits for ever for a rank that never arrives; after timeout_ms (<= 0: $CTL_COMM_TIMEOUT_MS, else
 * 120 000) the call fails with CTL_ERR_INVALID and a message naming the rank, so that the host can fall back or stop the job.  ctl_comm_create uses the default. */
int ctl_comm_create_timeout(const uint8_t id128[128], int32_t rank, int32_t world, int32_t timeout_ms, ctl_comm** out);
/* In place: the root's image becomes the sum over the ranks.  ONE call per render: with more than one rank a second call on an image that already went through an in-place
 * exchange is refused with CTL_ERR_INVALID on EVERY rank, before any of them enters the collective (it would add the other ranks' cumulative tiles onto sums that contain
 * them), until the image is cleared or rewritten. */
int ctl_image_reduce(ctl_image* img, ctl_comm* comm, int32_t root);
/* Out of place — the per-pass gather of a progressive display (the reference shows the frame after every DoPass, main.cpp:164-172): dst on the root receives the sum over the
 * ranks of `src`; every rank's src (its own cumulative tile frame) is left as it is, so the call can be repeated after every pass.  dst may be NULL on the other ranks.
 * K per-pass gathers end with the frame one end-of-render ctl_image_reduce gives, bit for bit (same ncclReduce over the same inputs). */
int ctl_image_reduce_to(ctl_image* src, ctl_image* dst, ctl_comm* comm, int32_t root);
/* The gather of BASELINE's north_star.  Every rank packs the tiles it owns into ceil(tiles / world) slots of 65 x 65 x 28 B: [slot k = tile k * world + rank][row-major
 * pixel][7 floats]; rows / columns 0..63 are the tile, row 64 / column 64 its HALO — the frame's pixels just right of and below the tile where they belong to another rank
 * (a sample's film position pixel + jitter rounds into the next pixel once in ~10^4 samples, as Image::AddSample's floor does in the reference, and at a tile edge that
 * pixel is another rank's: the rank accumulated it in its own full-size frame).  The clipped part of a border tile and a missing last slot are zero.  7.6 MB per rank at
 * 1920x1080 / 8 against the reduce's 58 MB.  ONE ncclGather moves the slots to the root; a streaming kernel there copies every tile into the frame, a second one adds the
 * halos.  Result: weights equal to the reduce's / the one-rank frame's exactly, colours to float rounding in the ~10^-5 of pixels that received a halo sample (bit-equal
 * elsewhere).  _to writes all of `dst` (root only; NULL elsewhere), leaves every rank's `src` alone and can be repeated — the per-pass progressive exchange; the in-place
 * form is ONE call per render like ctl_image_reduce (refused on every rank on a repeat).  Same deadline as the reduce: after the communicator's time-out the communicator
 * is aborted (ncclCommAbort), the call fails with CTL_ERR_INVALID and every later call on that communicator is refused. */
int ctl_image_gather(ctl_image* img, ctl_comm* comm, int32_t root);
int ctl_image_gather_to(ctl_image* src, ctl_image* dst, ctl_comm* comm, int32_t root);
/* The same packed slots for a host that moves them itself (MPI_Gather, a socket; bench.py's gloo fallback): bytes of one rank's buffer; D2H of `rank`'s slots of img;
 * H2D of ALL ranks' buffers, rank-major (MPI_Gather's receive buffer), into img on the root: every tile is copied, then every halo added. */
int ctl_image_packed_tile_bytes(uint32_t width, uint32_t height, uint32_t world, uint64_t* out_bytes);
int ctl_image_pack_tiles(ctl_image* img, uint32_t rank, uint32_t world, void* host_out);
int ctl_image_unpack_tiles(ctl_image* img, uint32_t world, const void* host_in_all_ranks);

/* ------------------------------------------------------------------- tracer */
typedef struct ctl_tracer ctl_tracer;
/* plugin names: "WavefrontPathTracer" (Integrators/PseudoRealtime/WavefrontPathTracer.h:24). */
int ctl_tracer_create(const char* plugin, ctl_tracer** out);
void ctl_tracer_destroy(ctl_tracer* t);
/* TracerParameterCollection (Kernel/TracerSettings.h:221-350): keys Direct(bool), MaxPathLength(int>=1),
 * RRStartDepth(int>=1) (WavefrontPathTracer.h:29-39). Out-of-interval values -> CTL_ERR_INVALID. */
int ctl_tracer_set_param_bool(ctl_tracer* t, const char* key, int value);
int ctl_tracer_set_param_int(ctl_tracer* t, const char* key, int value);
int ctl_tracer_get_param_int(ctl_tracer* t, const char* key, int* value_out);   /* bool, int and enum (its index) parameters */
/* float intervals and enumerations (TracerParameter<float>, TracerParameter<enum> with its string table, Kernel/TracerSettings.h:14-195):
 * an enum is set by the NAME of the value ("Uniform", "Variance", ...) or, through ctl_tracer_set_param_int, by its index */
int ctl_tracer_set_param_float(ctl_tracer* t, const char* key, float value);
int ctl_tracer_get_param_float(ctl_tracer* t, const char* key, float* value_out);
int ctl_tracer_set_param_enum(ctl_tracer* t, const char* key, const char* value_name);
int ctl_tracer_resize(ctl_tracer* t, uint32_t width, uint32_t height);       /* Tracer::Resize (Tracer.h:196-208)          */
int ctl_tracer_initialize_scene(ctl_tracer* t, ctl_scene* s);                /* TracerBase::InitializeScene (Tracer.h:102) */
/* image-tile sharding for multi-GPU: this tracer renders the 64x64 tiles t with t % world == rank. */
int ctl_tracer_set_tile_shard(ctl_tracer* t, uint32_t rank, uint32_t world);
/* Use caller-provided sampler tables for the next pass instead of the tracer's own generator (tests). */
int ctl_tracer_set_sampler_tables(ctl_tracer* t, const float* tables_1d, const float* tables_2d);
/* Tracer<true>::DoPass (Tracer.h:209-248). */
int ctl_tracer_do_pass(ctl_tracer* t, ctl_image* img, int new_trace);
/* n passes back-to-back without host synchronisation in between (throughput mode). */
int ctl_tracer_do_passes(ctl_tracer* t, ctl_image* img, int new_trace, uint32_t n_passes);
/* TracerBase::Debug(Image*, Vec2i) (Kernel/Tracer.h:119-123): UpdateKernel(scene, generator) draws the NEXT set of sampling tables from the tracer's stream —
 * the pass that follows uses the set after it — then DebugInternal follows ONE path for the pixel: PathTracer::DebugInternal (Integrators/PathTracer.cu:172-180)
 * = PathTrace<true> from the pixel's own position (no jitter) with that set; the WavefrontPathTracer has no DebugInternal of its own (nothing is traced).
 * rgb_out (3 floats, may be NULL) receives the path's radiance, which the reference computes and drops (it is looked at in a debugger). */
int ctl_tracer_debug_pixel(ctl_tracer* t, ctl_image* img, uint32_t x, uint32_t y, float* rgb_out);
/* IDepthTracer::setDepthBuffer(DeviceDepthImage{m_pData, w, h}) (Kernel/Tracer.h:16-57; WavefrontPathTracer : IDepthTracer): device_depth = width*height floats in
 * DEVICE memory (ctl_device_malloc); every pass stores DeviceDepthImage::NormalizeDepthD3D of the primary hit distance of pixel (x, y) — clamped to the camera's
 * [near, far], 1 for a miss — as pathIterateKernel does at pathDepth 0 (WavefrontPathTracer.cu:76-77).  NULL, 0, 0 removes it.  Wavefront plugin only. */
int ctl_tracer_set_depth_buffer(ctl_tracer* t, float* device_depth, uint32_t width, uint32_t height);
/* traversal statistics for the roofline: sums over rays of inner-node visits, triangle tests and instance entries
 * (B_ray = 32 + 16 + 64*N_inner + 52*N_tri + 108*N_inst). */
typedef struct { uint64_t n_inner, n_tri, n_inst;
                 uint64_t wave_inner_iters, wave_tri_iters;   /* wave-level loop iterations: n_inner / (64 * wave_inner_iters) = lane utilisation */
} ctl_traversal_counts;
typedef struct {
    uint64_t rays_last_pass;       /* getRaysInLastPass (64-bit; the reference wraps at 2^32)            */
    uint64_t rays_total;           /* getAccRays                                                         */
    double seconds_last_pass;      /* getLastTimeSpentRenderingSec                                       */
    double seconds_total;          /* getAccTimeSpentRenderingSec                                        */
    uint32_t passes_done;          /* getNumPassesDone                                                   */
    /* per-kernel HIP-event timing of the last do_pass/do_passes call, milliseconds (events on the tracer's stream) */
    double ms_intersect;           /* closest-hit intersect kernel (path rays)                           */
    double ms_shade, ms_raygen;
    double ms_intersect_any;       /* any-hit intersect kernel (NEE shadow rays)                         */
    uint64_t intersect_rays;       /* path rays through the closest-hit kernel in that call              */
    uint64_t intersect_launches;
    uint64_t shadow_rays;          /* shadow rays through the any-hit kernel in that call                */
    uint64_t shadow_launches;
    /* traversal statistics, filled only while ctl_tracer_set_counting(t, 1): sums over all rays of the call */
    ctl_traversal_counts closest_counts, any_counts;
    /* FuseTraversal (default): the path rays of bounce d >= 2 and the shadow rays of bounce d - 1 are traced by ONE persistent launch (k_intersect_pair).
     * Those launches, the rays they carried and their time are reported here; ms_intersect / intersect_launches and ms_intersect_any / shadow_launches
     * then cover only the launches that stayed separate (the first bounce's path rays, the last bounce's shadow rays).  intersect_rays and shadow_rays
     * remain the totals of the call. */
    uint64_t fused_launches, fused_shadow_rays, fused_closest_rays;
    double ms_fused;
} ctl_tracer_stats;
int ctl_tracer_get_stats(ctl_tracer* t, ctl_tracer_stats* out);
/* Block samplers of Tracer<true> (Kernel/BlockSampler/, Kernel/Tracer.h:209-248; wavefront plugin only).  The int parameter
 * "BlockSamplerType" selects Uniform (0, default), Variance (1), Difference (2) or Select (3); "FractionDeterministic" (2) and
 * "FractionWeighted" (4) are the mixed-sampling settings of the Variance / Difference samplers.  Blocks are 64 x 64 pixels, indexed
 * (block_x, block_y); ctl_tracer_set_block_weight = IUserPreferenceSampler::setWeight (call after ctl_tracer_resize).
 * ctl_tracer_get_block_counts returns the samples per block (row-major, blocks_x = ceil(width / 64)) of the last rendered pass, all ones
 * while the sampler takes every block once. */
/* Build-specific: allocate the ray queues now for a ctl_tracer_do_passes(n) to come; without it they grow inside that call. */
int ctl_tracer_reserve_passes(ctl_tracer* t, uint32_t n_passes);
int ctl_tracer_set_block_weight(ctl_tracer* t, uint32_t block_x, uint32_t block_y, float weight);
int ctl_tracer_get_block_counts(ctl_tracer* t, uint8_t* counts_out, uint32_t n_blocks);
/* run the intersect kernels in counting mode (N_inner / N_tri / N_inst of SURVEY §8d); slower, for measurement only */
int ctl_tracer_set_counting(ctl_tracer* t, int on);

/* ---------------------------------------------------- intersect (row a7 alone) */
/* __internal__IntersectBuffers (Kernel/TraceHelper.cu:736-746): n rays -> n hits; host pointers.
 * any_hit = 1 selects intersectKernel<true>. */
int ctl_intersect(ctl_scene* s, const ctl_ray* rays, uint32_t n, ctl_hit* hits, int any_hit);
/* TracerBase::TraceSingleRay(Ray, DynamicScene*) (Kernel/Tracer.cu:74-78): the closest hit of one ray (tmin = the scene's ray epsilon as the caller sets it in `ray`). */
int ctl_trace_single_ray(ctl_scene* s, const ctl_ray* ray, ctl_hit* hit_out);
/* device-pointer variant (the layout the tracer itself uses): d_ray_o[n], d_ray_d[n] = float4 (origin,tmin) / (direction,tmax);
 * d_hit4[n] = float4 (t, u, v, triangle index bits, -1 = miss); d_hit_node[n] = int32.  Synchronous; ms_out (may be NULL) =
 * HIP-event time of the kernel launch. */
int ctl_intersect_device(ctl_scene* s, const void* d_ray_o, const void* d_ray_d, uint32_t n, void* d_hit4, void* d_hit_node, int any_hit, float* ms_out);
/* traversal statistics for the roofline: sums over the n rays of inner-node visits, triangle tests and
 * instance entries (B_ray = 32 + 16 + 64*N_inner + 52*N_tri + 108*N_inst). */
int ctl_intersect_count(ctl_scene* s, const ctl_ray* rays, uint32_t n, int any_hit, ctl_traversal_counts* out);

/* Measurement: rays of all COUNTING traversals so far (ctl_intersect_count, ctl_tracer_set_counting) over the flattened BVH, by the deepest traversal-stack entry they
 * used; bin n_bins - 1 collects everything deeper.  The kernels keep the first 19 entries of a lane in LDS and deeper ones in scratch (csrc/traverse_flat.h): the
 * histogram says how often that happens.  reset != 0 clears it. */
int ctl_traversal_stack_histogram(uint64_t* out, uint32_t n_bins, int reset);

/* The shared fp32 transcendental functions of the shading code (cudatracerlib_amd/csrc/ctl_fmath.h; the oracle's -DORC_SHARED_MATH build runs the same source):
 * which = 0 sin, 1 cos, 2 tan, 3 acos, 4 atan, 5 atan2(x, y), 6 exp, 7 log, 8 log2, 9 pow(x, y); on_device = 0 evaluates on the host, 1 in a kernel — the two are
 * bit-identical (tests/test_fmath.py).  No reference counterpart: the reference calls the CUDA / C library's functions. */
int ctl_shared_math_eval(int32_t which, uint32_t n, const float* x, const float* y, float* out, int32_t on_device);

/* device memory helpers so that Python callers need no HIP binding */
int ctl_device_malloc(size_t bytes, void** out);
int ctl_device_free(void* p);
int ctl_memcpy_h2d(void* dst, const void* src, size_t bytes);
int ctl_memcpy_d2h(void* dst, const void* src, size_t bytes);
int ctl_memcpy_d2d(void* dst, const void* src, size_t bytes);
int ctl_device_synchronize(void);
int ctl_set_device(int ordinal);

#ifdef __cplusplus
}
#endif
#endif /* CTL_AMD_H */
