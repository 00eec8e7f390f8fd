// kernels.h — per-bounce HIP kernels of the wavefront path tracer and their launch wrappers (kernels.hip).
#pragma once
#include "device_scene.h"
#include <hip/hip_runtime.h>

namespace ctl {

// One path per slot, structure-of-arrays so that a wave's 64 lanes read/write 1 KiB contiguous per array.
struct path_soa {
    float4* ray_o;    // origin.xyz, tmin                                  (traversalRay::a, Kernel/TraceHelper.h:55-59)
    float4* ray_d;    // direction.xyz, tmax                               (traversalRay::b)
    float4* thr;      // throughput cf.rgb, last bsdf pdf                  (WavefrontPTRayData::throughput, bsdf_pdf)
    float4* rad;      // accumulated radiance cl.rgb, pixel index (bits)   (WavefrontPTRayData::L, x/y)
    float4* nor;      // normal of the previous vertex xyz, packed {d1:8, d2:8, pass-in-batch:8, flags:8}
    float4* pend;     // pending NEE contribution directF.rgb, shadow-ray index (bits)   (directF, dIdx)
    float2* px;       // film sample position pX
};
enum { kFlagSpecular = 1 };
constexpr uint32_t kNoShadow = 0xffffffffu;

struct final_soa {    // terminated paths that still wait for a shadow ray
    float4* rad;      // cl.rgb, shadow-ray index
    float4* dir;      // directF.rgb, -
    float4* px;       // film sample position pX.xy, pixel index (bits), pass-in-batch (bits)
};

struct wave_queues {
    path_soa path[2];          // ping-pong per bounce
    float4* hit;               // t, u, v, triangle (bits; -1 = miss)
    int* hit_node;
    float4* sh_o[2]; float4* sh_d[2]; uint32_t* sh_occ[2];   // shadow rays of bounce d (write) / d-1 (read)
    final_soa fin;
    uint32_t* counts;          // [ (depth+1)*4 + {0: n_paths, 1: n_shadow, 2: n_final} ]
    uint32_t* work;            // dynamic-fetch cursors, one per intersect launch of a pass
    unsigned long long* stats; // [0] rays traced (primary + continuation + shadow)
    uint32_t capacity;
    // material sort of the shading queue (scenes that need the full shade kernel): path slots grouped by BSDF model, so that the 64 lanes of
    // a wave run one model's code instead of all of them
    uint32_t* order;           // [capacity] path slots in shading order
    unsigned char* mat_key;    // [capacity] BSDF model of the hit (0 = miss)
    uint32_t* mat_counts;      // [depth * 32 + k]: k < 16 paths per model, 16 + k scatter cursors, 24 + c vertices of model class c (k_class_partition)
    // model-class shading (pass_params::model_classes): k_class_partition splits the slots of a depth into one list per model class, class_order[c][0 .. mat_counts[depth * 32 + 24 + c])
    uint32_t* class_order[5];  // [capacity] each: classes a, b, c, p, g
};
// the model classes of shade_class_a/b/c/p.hip: which traversal keys (Q.mat_key: CTL_BSDF_* of the hit, 0 = miss) each launch shades
#define CTL_CLASS_A_KEYS 0x004Bu   // miss, diffuse, dielectric, conductor
#define CTL_CLASS_G_KEYS 0x0080u   // rough conductor alone (the heaviest model of the basic set: with it class a spills 59 registers, without 15)
#define CTL_CLASS_B_KEYS 0x1D34u   // rough diffuse, thin dielectric, rough dielectric, plastic, Phong, Ward, Hanrahan-Krueger
#define CTL_CLASS_P_KEYS 0x0200u   // rough plastic alone: the commonest model of interiors and the heaviest single-layer one — by itself its BSDF record stays in registers (144 B of scratch; together with class b's models: 400 B)
#define CTL_CLASS_C_KEYS 0xE000u   // coating, rough coating, blend (the nesting models)

struct pass_params {
    const float* t1; const float2* t2;   // sampler tables of the first pass of this batch; pass b of the batch at + b * 4096*30
    uint32_t batch;                      // passes rendered together in this wavefront (paths carry their pass id)
    uint32_t width, height;              // full film
    uint32_t tile_rank, tile_world;      // image-tile shard: tiles t with t % world == rank
    uint32_t n_local_pixels;             // pixels rendered by this rank
    int direct, max_path_length, rr_start_depth;
    int sort_materials;                  // shade in wave_queues::order
    int key_from_traversal;              // Q.mat_key[i] = BSDF model (CTL_BSDF_*, all >= 1) of path i's hit (0 = miss), left there by the closest-hit traversal (dev_scene::hit_key_out)
    int model_classes;                   // the full feature set shaded by one launch per model class present in the scene (shade_class_*.hip) instead of the one k_shade_full; needs key_from_traversal
    int block_sort;                      // full shade kernel: regroup the path slots of a workgroup by BSDF model (shade_kernel.inc)
    int sort_octants;                    // append the new rays of a workgroup grouped by direction octant (compaction.h)
    const unsigned char* block_counts;   // samples per 64x64 film block in this pass (a block sampler's decision), nullptr = one everywhere
    uint32_t max_block_count;            // largest entry of block_counts
    int wavefront_rules;                 // pathIterateKernel's own path rules (PathSemantics = Wavefront): selects the *_wf shade kernels
    int u16_bary;                        // hit barycentrics through the traversal result's 16-bit pair (U16Barycentrics)
    float* debug_out; uint32_t debug_x, debug_y;   // TracerBase::Debug (Kernel/Tracer.h:119-123): k_path_trace follows ONE path from the centre-less pixel position (x, y) and writes its radiance here
    float* depth_buffer; uint32_t depth_w, depth_h; float depth_near, depth_far;   // IDepthTracer::setDepthBuffer (Kernel/Tracer.h:16-57), nullptr = none    // Ordered accumulation.  Image::AddSample is four float atomics per finished path; a batch holds B passes of every pixel, so they collide on the same 28 bytes and
    // cost 3 % of the whole job — and they add in whatever order the hardware serves them.  With `stage` set, a finished path stores its sample (rgb, weight 1) at
    // stage[pass-in-batch][pixel] with one plain store (one path per pixel and pass: no two writers), and k_resolve_stage adds the batch to the frame pass by pass, in pass order — the order in which
    // the reference's one-pass-at-a-time loop adds them.  nullptr: atomics (block samplers hand out 0 / 1 / 2 samples per pixel; the megakernel plugin).
    float4* stage; size_t stage_stride;
};

struct launch_ctx { hipStream_t stream; int grid_blocks; bool alpha_test = false; };   // alpha_test: intersect kernels run Material::AlphaTest on candidate hits

// sampling-sequence tables of n_passes passes written to (t1, t2) in HBM; jumps = sequence_generator::chunk_jump_matrices(), starts = n_passes x sequence_generator::pass_start
void launch_sequence_fill(hipStream_t stream, const uint32_t* jumps, const void* starts, uint32_t n_passes, float* t1, float* t2);

// measurement knobs (environment: CTL_REFILL_IDLE), applied once per process
void apply_tuning_from_env();
void launch_raygen(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P);
// intersect `n = counts[count_slot]` rays (device-side count) from (ro, rd) into (hit, hit_node) or into occ (any-hit)
void launch_intersect_closest(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node);
void launch_intersect_any(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, uint32_t* occ, float4* hit = nullptr, int* hit_node = nullptr);
// the two above in one persistent launch: closest hits of (ro, rd), then occlusion of (sro, srd)
void launch_intersect_pair(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node,
                           const float4* sro, const float4* srd, const uint32_t* sn_ptr, uint32_t* swork, uint32_t* occ);
void launch_intersect_count(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node,
                            uint32_t* occ, int any_hit, unsigned long long* counts3);
void launch_shade(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void read_stack_histogram(unsigned long long* h, bool reset);   // kernels.hip: rays of the counting traversals by deepest stack entry
void launch_shade_basic(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_full(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_class_partition(const launch_ctx& lc, const wave_queues& Q, int depth);
void launch_shade_class_a(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_b(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_c(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_g(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_g_wf(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_p(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_p_wf(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_a_wf(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_b_wf(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_class_c_wf(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_basic_wf(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_shade_full_wf(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_finalize(const launch_ctx& lc, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image);
void launch_resolve_stage(const launch_ctx& lc, float4* stage, size_t stride, uint32_t n_passes, ctl_pixel_data* image, uint32_t W, uint32_t H, uint32_t tile_rank, uint32_t tile_world);   // frame += the staged samples of a batch, pass by pass; clears the stage
int flat_top_cache_nodes();   // nodes at the head of the flattened node array that the traversal workgroups keep in LDS (traverse_flat.h kTopCache)
void launch_accumulate_stats(const launch_ctx& lc, const wave_queues& Q, int max_depth);
void launch_apply_pipeline(const launch_ctx& lc, const ctl_pixel_data* image, uint32_t n, float splat_scale, uint32_t* rgbcol_out);
void launch_resolve_rgb(const launch_ctx& lc, const ctl_pixel_data* image, uint32_t n, float splat_scale, float* rgb_out);
void launch_add_samples(const launch_ctx& lc, ctl_pixel_data* image, uint32_t W, uint32_t H, uint32_t n, const float* samples5);   // Image::AddSample for n samples {sx, sy, r, g, b}

// local pixel index -> film pixel for a tile shard (64x64 tiles, 8x8 micro-tiles inside = one wave)
__host__ __device__ inline uint32_t shard_pixel_count(uint32_t W, uint32_t H, uint32_t rank, uint32_t world) {
    const uint32_t tx = (W + 63) / 64, ty = (H + 63) / 64, nt = tx * ty;
    uint32_t mine = nt / world + ((rank < nt % world) ? 1u : 0u);
    return mine * 64u * 64u;   // upper bound incl. clipped pixels; clipped lanes generate no path
}

} // namespace ctl
