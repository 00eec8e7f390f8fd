// kernels.hip — gfx950 kernels of the wavefront path tracer: ray generation, BVH traversal + Woop intersection,
// shading (BSDF sample/eval + next-event estimation + Russian roulette) with wave-level stream compaction, and
// framebuffer accumulation.  The per-path semantics are those of the reference's PathTrace<DIRECT>
// (Integrators/PathTracer.cu:10-113) re-cut at its two trace points; see DESIGN.md "Kernels".
#include "kernels.h"
#include "traverse.h"
#ifdef CTL_FLAT_EXPERIMENTS
#include "experiments/traverse_flat_variants.h"   // round 2 / 3's variant arms (F4 / F2 node formats, quad fetch, top cache, stack distances, ...) instead of the product header; no 8-wide format in such a build
#else
#include "traverse_flat.h"
#include "traverse_flat8.h"
#endif
#ifndef CTL_LEAF_QUEUE
#define CTL_LEAF_QUEUE 0   // 1: the 4-wide flattened traversal runs its entry tests from a wave-wide queue (experiments/traverse_flat_wq.h: measured, 38-47 % slower)
#endif
#if CTL_LEAF_QUEUE
#include "experiments/traverse_flat_wq.h"
#endif
#include "knobs.h"
#include <stdexcept>
#include "shading.h"
#include "compaction.h"
#include <cstdlib>

namespace ctl {

// ------------------------------------------------------------------------------------------------ ray generation
// pathCreateKernelWPT (Integrators/PseudoRealtime/WavefrontPathTracer.cu:17-49) with the megakernel's sampler
// indexing (sampler index = film pixel index, Integrators/PathTracer.cu:185-190).
__global__ __launch_bounds__(kWideBlock) void k_raygen(dev_scene S, wave_queues Q, pass_params P) {
    __shared__ uint32_t s_cnt[3][kWideBlock / 64]; __shared__ uint32_t s_base[3];
    const uint32_t tiles_x = (P.width + 63) / 64;
    const uint32_t n_total = P.n_local_pixels * P.batch;   // n_local_pixels is a multiple of 4096: waves never straddle passes
    const uint32_t n1 = CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH;
    for (uint32_t gi = blockIdx.x * kWideBlock + threadIdx.x; gi < n_total; gi += gridDim.x * kWideBlock) {   // n_total is a multiple of 4096
        const uint32_t pass_b = gi / P.n_local_pixels, li = gi - pass_b * P.n_local_pixels;
        const uint32_t tile = P.tile_rank + (li >> 12) * P.tile_world, p = li & 4095u, micro = p >> 6, lane = p & 63u;
        const uint32_t x = (tile % tiles_x) * 64 + (micro & 7u) * 8 + (lane & 7u), y = (tile / tiles_x) * 64 + (micro >> 3) * 8 + (lane >> 3);
        // BlockSamplerBuffer::getNumSamplesPerPixel (WavefrontPathTracer.cu:31-36): 0, 1 or more samples, all drawn from the pixel's one sampler
        const uint32_t n_smp = P.block_counts ? P.block_counts[tile] : 1u, max_smp = P.block_counts ? P.max_block_count : 1u;
        for (uint32_t smp = 0; smp < max_smp; smp++) {
        const bool valid = x < P.width && y < P.height && smp < n_smp;
        const uint32_t slot = block_append3(&Q.counts[0], valid, nullptr, false, nullptr, false, s_cnt, s_base).s[0];
        if (!valid) continue;
        const uint32_t pixel = y * P.width + x;
        sampler rng{ P.t1 + pass_b * n1, P.t2 + pass_b * n1, pixel, 0, 2 * smp };
        const f2 j = rng.next2();
        const f2 pX{ (float)x + j.x, (float)y + j.y };
        const f2 ap = rng.next2();   // aperture sample (PathTracer.cu:190; thin-lens and telecentric sensors use it)
        f3 o, d; sensor_sample_ray(S.cam, pX, ap, o, d);
        const path_soa& A = Q.path[0];
        A.ray_o[slot] = make_float4(o.x, o.y, o.z, S.eps);          // DoubleRayBuffer::convert (Kernel/DoubleRayBuffer.h:224-230)
        A.ray_d[slot] = make_float4(d.x, d.y, d.z, 3.402823466e+38f);
        A.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        A.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(pixel));
        A.nor[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(rng.d1 | (rng.d2 << 8) | (pass_b << 16) | (0u << 24)));
        A.pend[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kNoShadow));
        A.px[slot] = make_float2(pX.x, pX.y);
        }
    }
}

// ------------------------------------------------------------------------------------------------ intersection
// Persistent waves with lane refill and an LDS traversal stack — see traverse.h (the reference's g_warpCounter pool,
// Kernel/TraceHelper.cu:386-399, re-derived for 64-wide waves).
#ifndef CTL_INTERSECT_MIN_WAVES
#define CTL_INTERSECT_MIN_WAVES 7   // waves per SIMD the register allocation of the FLATTENED traversal kernels leaves room for (72 VGPRs, no spills; the two-level kernels stay at 6: 24 KiB of LDS stack).  Measured with the slab build (profiles/r03_occupancy.log): 6 (80 VGPRs) 18.71 ms per fused launch, 7: 18.25, 8 (64 VGPRs, 11 spilled) 20.98; fewer resident workgroups (LDS padding): 5: 20.1, 4: 22.7, 3: 27.4
#endif
// ints of LDS a traversal workgroup keeps for its lanes' stacks, and the body for a flattened layout (LAYOUT = 1 + flat_format)
#ifdef CTL_FLAT_EXPERIMENTS
constexpr int lds_stack_ints(int layout) { return (layout ? (kFlatLdsRows + 1) * kFlatStackInts : kLdsStack) * kBlock; }
template <bool ANY_HIT, bool COUNT, int LAYOUT, bool ALPHA>
__device__ __forceinline__ void intersect_flat_layout(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work, float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_stack, trav_counts& tc) {
    __shared__ uint16_t lds_dist[CTL_STACK_DIST >= 2 ? (kFlatLdsRows + 1) * kBlock : 1];
    __shared__ __attribute__((aligned(16))) float lds_top[kTopCacheFloats];
    fill_top_cache(S, lds_top);
    intersect_flat<ANY_HIT, COUNT, ALPHA, LAYOUT - 1>(S, ro, rd, n, work, hit, hit_node, occ, lds_stack, lds_top, tc, lds_dist);
}
#else
#if CTL_LEAF_QUEUE
constexpr int lds_stack_ints(int layout) { return layout == 1 + kFmtQ8 ? (kQ8LdsRows + 1) * 2 * kBlock : layout == 1 + kFmtQ4 ? kWqLdsInts : kLdsStack * kBlock; }
#else
constexpr int lds_stack_ints(int layout) { return layout == 1 + kFmtQ8 ? (kQ8LdsRows + 1) * 2 * kBlock : (layout ? kFlatLdsRows + 1 : kLdsStack) * kBlock; }
#endif
template <bool ANY_HIT, bool COUNT, int LAYOUT, bool ALPHA>
__device__ __forceinline__ void intersect_flat_layout(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work, float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_stack, trav_counts& tc) {
    if (LAYOUT == 1 + kFmtQ8) intersect_flat8<ANY_HIT, COUNT, ALPHA>(S, ro, rd, n, work, hit, hit_node, occ, (unsigned long long*)lds_stack, tc);
#if CTL_LEAF_QUEUE
    else intersect_flat_wq<ANY_HIT, COUNT, ALPHA>(S, ro, rd, n, work, hit, hit_node, occ, lds_stack, tc);
#else
    else intersect_flat<ANY_HIT, COUNT, ALPHA>(S, ro, rd, n, work, hit, hit_node, occ, lds_stack, tc);
#endif
}
#endif
template <bool ANY_HIT, bool COUNT, int LAYOUT, bool ALPHA>   // LAYOUT: 0 two-level, 1 + flat_format for the flattened structure; ALPHA: alpha-test candidate hits
#ifdef CTL_INTERSECT_EXACT_WAVES   // measurement builds: hold the traversal kernels to exactly this many waves per SIMD
#define CTL_INTERSECT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(CTL_INTERSECT_EXACT_WAVES, CTL_INTERSECT_EXACT_WAVES)))
#else
#define CTL_INTERSECT_WAVES_ATTR
#endif
__global__ __launch_bounds__(kBlock, ((LAYOUT && !ALPHA) ? CTL_INTERSECT_MIN_WAVES : 6)) CTL_INTERSECT_WAVES_ATTR void k_intersect(dev_scene S, const float4* __restrict__ ro, const float4* __restrict__ rd, const uint32_t* __restrict__ n_ptr,
                                                       uint32_t* __restrict__ work, float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ,
                                                       unsigned long long* __restrict__ counts3) {
    __shared__ __attribute__((aligned(8))) int lds_stack[lds_stack_ints(LAYOUT)];   // flat: + one spare row that absorbs unused push slots
    const uint32_t n = *n_ptr;
    trav_counts tc{ 0, 0, 0, 0, 0 };
    if (LAYOUT) intersect_flat_layout<ANY_HIT, COUNT, LAYOUT, ALPHA>(S, ro, rd, n, work, hit, hit_node, occ, lds_stack, tc);
    else intersect_persistent<ANY_HIT, COUNT, ALPHA>(S, ro, rd, n, work, hit, hit_node, occ, lds_stack, tc);
    if (COUNT) {
        atomicAdd(&counts3[0], (unsigned long long)tc.n_inner); atomicAdd(&counts3[1], (unsigned long long)tc.n_tri); atomicAdd(&counts3[2], (unsigned long long)tc.n_inst);
        atomicAdd(&counts3[3], (unsigned long long)tc.w_inner); atomicAdd(&counts3[4], (unsigned long long)tc.w_tri);
    }
}

// Closest-hit traversal of one bounce's path rays, then any-hit traversal of the previous bounce's shadow rays, in ONE persistent launch: a wave that finds
// the first queue exhausted goes straight on to the second, so the drain of the first set (waves running down their last long rays, ~0.4 ms per launch
// whatever its size) is filled with the second set's work.  Same code, same results as the two separate launches (tracer.hip decides which to use).
template <int LAYOUT, bool ALPHA>
__global__ __launch_bounds__(kBlock, ((LAYOUT && !ALPHA) ? CTL_INTERSECT_MIN_WAVES : 6)) CTL_INTERSECT_WAVES_ATTR void k_intersect_pair(dev_scene S, const float4* __restrict__ ro, const float4* __restrict__ rd, const uint32_t* __restrict__ n_ptr,
                                                            uint32_t* __restrict__ work, float4* __restrict__ hit, int* __restrict__ hit_node,
                                                            const float4* __restrict__ sro, const float4* __restrict__ srd, const uint32_t* __restrict__ sn_ptr,
                                                            uint32_t* __restrict__ swork, uint32_t* __restrict__ occ) {
    __shared__ __attribute__((aligned(8))) int lds_stack[lds_stack_ints(LAYOUT)];
    const uint32_t n = *n_ptr, sn = *sn_ptr;
    trav_counts tc{ 0, 0, 0, 0, 0 };
    if (LAYOUT) {
        intersect_flat_layout<false, false, LAYOUT, ALPHA>(S, ro, rd, n, work, hit, hit_node, nullptr, lds_stack, tc);
        intersect_flat_layout<true, false, LAYOUT, ALPHA>(S, sro, srd, sn, swork, nullptr, nullptr, occ, lds_stack, tc);
    } else {
        intersect_persistent<false, false, ALPHA>(S, ro, rd, n, work, hit, hit_node, nullptr, lds_stack, tc);
        intersect_persistent<true, false, ALPHA>(S, sro, srd, sn, swork, nullptr, nullptr, occ, lds_stack, tc);
    }
}

int flat_top_cache_nodes() { return kTopCache; }

// terminated paths whose last NEE shadow ray has now been traced
__global__ __launch_bounds__(kBlock) void k_finalize(wave_queues Q, pass_params P, int depth, ctl_pixel_data* __restrict__ image) {
    const uint32_t n = Q.counts[depth * 4 + 2];
    const uint32_t* occ = Q.sh_occ[depth & 1];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 rad = Q.fin.rad[i], dir = Q.fin.dir[i], px = Q.fin.px[i];
        f3 cl(rad.x, rad.y, rad.z);
        if (!occ[__float_as_uint(rad.w)]) cl = cl + f3(dir.x, dir.y, dir.z);
        add_sample_ordered(P, image, __float_as_uint(px.z), __float_as_uint(px.w), px.x, px.y, cl);
    }
}

// frame += the staged samples of a batch (pass_params::stage), pass by pass in pass order; the stage is left cleared for the next batch.  One lane = one slot of the rank's
// own tiles (compaction.h add_sample_ordered: local tile * 4096 + row-major position in the tile); the n_passes reads of a lane are n_passes coalesced streams.
__global__ __launch_bounds__(256) void k_resolve_stage(float4* __restrict__ stage, size_t stride, uint32_t n_passes, ctl_pixel_data* __restrict__ image, uint32_t W, uint32_t H, uint32_t tile_rank, uint32_t tile_world) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= stride) return;
    const uint32_t tiles_x = (W + 63u) >> 6, tile = (uint32_t)(i >> 12) * tile_world + tile_rank, p = (uint32_t)i & 4095u;
    const uint32_t x = (tile % tiles_x) * 64u + (p & 63u), y = (tile / tiles_x) * 64u + (p >> 6);
    if (x >= W || y >= H) return;     // the clipped part of a border tile: nothing was staged there
    ctl_pixel_data* r = image + ((size_t)y * W + x);
    float a = r->rgb[0], b = r->rgb[1], c = r->rgb[2], w = r->weight_sum; bool any = false;
    for (uint32_t q = 0; q < n_passes; q++) {
        const float4 v = stage[(size_t)q * stride + i];
        if (v.w != 0.0f) { a += v.x; b += v.y; c += v.z; w += v.w; any = true; stage[(size_t)q * stride + i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    }
    if (any) { r->rgb[0] = a; r->rgb[1] = b; r->rgb[2] = c; r->weight_sum = w; }
}
void launch_resolve_stage(const launch_ctx& lc, float4* stage, size_t stride, uint32_t n_passes, ctl_pixel_data* image, uint32_t W, uint32_t H, uint32_t tile_rank, uint32_t tile_world) {
    hipLaunchKernelGGL(k_resolve_stage, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, lc.stream, stage, stride, n_passes, image, W, H, tile_rank, tile_world);
}

// rays of a pass = sum over bounces of (path rays + shadow rays)  (Kernel/TraceHelper.cu:176,745)
__global__ void k_accumulate_stats(wave_queues Q, int max_depth) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long r = 0, s = 0;
        for (int d = 0; d <= max_depth; d++) { r += (unsigned long long)Q.counts[d * 4 + 0]; s += (unsigned long long)Q.counts[d * 4 + 1]; }
        Q.stats[0] += r; Q.stats[1] += s;
        Q.stats[12] += (unsigned long long)Q.counts[max_depth * 4 + 1];   // the shadow rays of the last bounce and the path rays of the first: always their own launches (tracer.hip)
        Q.stats[13] += (unsigned long long)Q.counts[0];
    }
}

// ---- sampling-sequence tables of a batch of passes, generated in HBM (sequence_generator.h: the same single XORWOW stream the host generator draws from,
// Kernel/Sampler.h:57-85).  One lane = one chunk of 16 sequences = 1440 consecutive draws; its start state is two GF(2) jumps from the pass's start state.
struct seq_pass_start { uint32_t v[5]; uint32_t d; };
__device__ __forceinline__ void xorwow_jump(const uint32_t* __restrict__ M, uint32_t* v) {   // v <- M v over GF(2); M: 160 rows x 5 words, row = image of a state bit
    uint32_t o[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) if (v[i] & (1u << j)) { const uint32_t* r = M + (i * 32 + j) * 5; for (int k = 0; k < 5; k++) o[k] ^= r[k]; }
    for (int k = 0; k < 5; k++) v[k] = o[k];
}
__global__ __launch_bounds__(kBlock) void k_sequence_fill(const uint32_t* __restrict__ jumps, const seq_pass_start* __restrict__ starts, uint32_t n_passes,
                                                           float* __restrict__ t1, float* __restrict__ t2) {
    constexpr uint32_t N = CTL_SAMPLER_NUM_SEQUENCES, L = CTL_SAMPLER_SEQUENCE_LENGTH, kSeq = 16, kChunks = N / kSeq, kDraws = kSeq * L * 3;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_passes * kChunks) return;
    const uint32_t pass = g / kChunks, c = g % kChunks;
    uint32_t v[5]; for (int k = 0; k < 5; k++) v[k] = starts[pass].v[k];
    uint32_t d = starts[pass].d + 362437u * (kDraws * c);   // the Weyl counter advances linearly
    xorwow_jump(jumps + (c & 15u) * 800u, v);
    xorwow_jump(jumps + (16u + (c >> 4)) * 800u, v);
    auto uniform = [&]() {   // xorwow::next + uniform (sequence_generator.h; Base/CudaRandom.h:112-127)
        const uint32_t t = (v[0] ^ (v[0] >> 2));
        v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
        v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
        d += 362437u;
        const float inv = 2.3283064e-10f;
        const float f = (float)(v[4] + d) * inv + (inv / 2.0f);
        return f * (1 - 1e-5f);
    };
    float* p1 = t1 + (size_t)pass * N * L; float* p2 = t2 + (size_t)pass * N * L * 2;
    for (uint32_t s = c * kSeq; s < (c + 1) * kSeq; s++) {
        for (uint32_t i = 0; i < L; i++) p1[i * N + s] = uniform();
        for (uint32_t i = 0; i < L; i++) { const float y = uniform(), x = uniform(); p2[2 * (i * N + s)] = x; p2[2 * (i * N + s) + 1] = y; }   // .y is drawn first (Sampler.h:83)
    }
}
void launch_sequence_fill(hipStream_t stream, const uint32_t* jumps, const void* starts, uint32_t n_passes, float* t1, float* t2) {
    const uint32_t lanes = n_passes * (CTL_SAMPLER_NUM_SEQUENCES / 16);
    hipLaunchKernelGGL(k_sequence_fill, dim3((lanes + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, jumps, (const seq_pass_start*)starts, n_passes, t1, t2);
}

// copySamplesToOutput (Kernel/ImagePipeline/ImagePipeline.cu:14-30) up to the linear-RGB value (PixelData::toSpectrum, Engine/Image.h:21-28)
__global__ __launch_bounds__(kBlock) void k_resolve_rgb(const ctl_pixel_data* __restrict__ image, uint32_t n, float splat_scale, float* __restrict__ out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const ctl_pixel_data p = image[i];
        const float w = p.weight_sum != 0 ? p.weight_sum : 1;
        out[i * 3 + 0] = p.rgb[0] / w + p.rgb_splat[0] * splat_scale;
        out[i * 3 + 1] = p.rgb[1] / w + p.rgb_splat[1] * splat_scale;
        out[i * 3 + 2] = p.rgb[2] / w + p.rgb_splat[2] * splat_scale;
    }
}

// copySamplesToOutput (Kernel/ImagePipeline/ImagePipeline.cu:8-23): PixelData::toSpectrum(splatScale) -> sRGB transfer curve
// (Spectrum.cu:229-235) -> RGBCOL (Spectrum.h:521-526), the pipeline without filter / post-process
__global__ __launch_bounds__(kBlock) void k_apply_pipeline(const ctl_pixel_data* __restrict__ image, uint32_t n, float splat_scale, uint32_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const ctl_pixel_data p = image[i];
        const float w = p.weight_sum != 0 ? p.weight_sum : 1;
        uint32_t packed = 255u << 24;
        for (int c = 0; c < 3; c++) {
            float v = p.rgb[c] / w + p.rgb_splat[c] * splat_scale;
            v = v <= 0.0031308f ? 12.92f * v : 1.055f * powf(v, (float)(1.0 / 2.4)) - 0.055f;
            packed |= (uint32_t)(unsigned char)(clampf(v, 0.0f, 1.0f) * 255.0f) << (8 * c);
        }
        out[i] = packed;
    }
}

// Image::AddSample (Engine/Image.cu:22-44) for a list of samples: compaction.h add_sample, the function the shade kernels deposit finished paths with when no stage is set
__global__ __launch_bounds__(kBlock) void k_add_samples(ctl_pixel_data* __restrict__ image, uint32_t W, uint32_t H, uint32_t n, const float* __restrict__ s) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) add_sample(image, W, H, s[5 * i], s[5 * i + 1], f3(s[5 * i + 2], s[5 * i + 3], s[5 * i + 4]));
}
void launch_add_samples(const launch_ctx& lc, ctl_pixel_data* image, uint32_t W, uint32_t H, uint32_t n, const float* samples5) {
    hipLaunchKernelGGL(k_add_samples, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, image, W, H, n, samples5);
}

// the counting kernels' stack-depth histogram (traverse_flat.h g_stack_hist lives in this translation unit)
void read_stack_histogram(unsigned long long* h, bool reset) {
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stack_hist), sizeof(unsigned long long) * kStackSize) != hipSuccess) throw std::runtime_error("reading the stack histogram failed");
    if (reset) { unsigned long long z[kStackSize] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stack_hist), z, sizeof(z)) != hipSuccess) throw std::runtime_error("clearing the stack histogram failed"); }
}
static unsigned g_lds_pad = 0;   // extra dynamic LDS per traversal workgroup: holds the kernels to fewer resident workgroups per CU (occupancy experiment)
// Grid of a traversal launch: exactly the workgroups that are resident together — per CU the waves per SIMD of the kernel's __launch_bounds__ (a 256-lane workgroup is one wave
// on each of the four SIMDs), or what the CU's 160 KB of LDS hold if that is fewer.  The waves' first ray claims are static (traverse.h ray_claims): a workgroup that had to
// wait for a place would keep its share of the rays waiting with it.  (lc.grid_blocks = 8 workgroups per CU.)
static int traversal_blocks(const launch_ctx& lc, int layout, bool alpha) {
    const int cus = lc.grid_blocks / 8, by_regs = (layout && !alpha) ? CTL_INTERSECT_MIN_WAVES : 6;
    const int by_lds = (int)((160u * 1024u) / ((unsigned)sizeof(int) * (unsigned)lds_stack_ints(layout) + g_lds_pad));
    return cus * (by_regs < by_lds ? by_regs : by_lds);
}
void apply_tuning_from_env() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char* e = knob_env("CTL_LDS_PAD")) { int v = atoi(e); if (v >= 0 && v <= 140000) g_lds_pad = (unsigned)v; }
    if (const char* e = knob_env("CTL_REFILL_IDLE")) { int v = atoi(e); if (v >= 1 && v <= 64) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_refill_idle), &v, sizeof(v)); }
    if (const char* e = knob_env("CTL_CHUNK_GUIDED")) { int v = atoi(e) != 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chunk_guided), &v, sizeof(v)); }
#if CTL_LEAF_QUEUE
    if (const char* e = knob_env("CTL_WQ_FLUSH")) { int v = atoi(e); if (v >= 1 && v <= 64) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wq_flush), &v, sizeof(v)); }
    if (const char* e = knob_env("CTL_WQ_MIN_INNER")) { int v = atoi(e); if (v >= 0 && v <= 64) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wq_min_inner), &v, sizeof(v)); }
#endif
    if (const char* e = knob_env("CTL_LEAF_BATCH")) { int v = atoi(e); if (v >= 1 && v <= 64) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_batch), &v, sizeof(v)); }
#if !CTL_LEAF_QUEUE && !defined(CTL_FLAT_EXPERIMENTS)
    if (const char* e = knob_env("CTL_LEAF_BATCH_ANY")) { int v = atoi(e); if (v >= 1 && v <= 64) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_batch_any), &v, sizeof(v)); }
#endif
}

// ------------------------------------------------------------------------------------------------ launch wrappers
void launch_raygen(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P) {
    hipLaunchKernelGGL(k_raygen, dim3(lc.grid_blocks / 4), dim3(kWideBlock), 0, lc.stream, S, Q, P);
}
#define CTL_LAUNCH_INTERSECT_L(ANY, CNT, L, ...)                                                                                     \
    do {                                                                                                                             \
        if (lc.alpha_test) hipLaunchKernelGGL((k_intersect<ANY, CNT, L, true>), dim3(traversal_blocks(lc, L, true)), dim3(kBlock), g_lds_pad, lc.stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((k_intersect<ANY, CNT, L, false>), dim3(traversal_blocks(lc, L, false)), dim3(kBlock), g_lds_pad, lc.stream, __VA_ARGS__);     \
    } while (0)
// The product traverses Q4 nodes.  The F4 / F2 node formats are measured experiments (0.69x and 0.66x of Q4's rays/s); their kernels are
// compiled only with -DCTL_FLAT_EXPERIMENTS, and a scene asking for them is refused otherwise (tracer.hip).
#ifdef CTL_FLAT_EXPERIMENTS
#define CTL_LAUNCH_INTERSECT(ANY, CNT, ...)                                                                                          \
    do {                                                                                                                             \
        if (!S.flat_nodes) CTL_LAUNCH_INTERSECT_L(ANY, CNT, 0, __VA_ARGS__);                                                         \
        else if (S.flat_format == kFmtF4) CTL_LAUNCH_INTERSECT_L(ANY, CNT, 2, __VA_ARGS__);                                          \
        else if (S.flat_format == kFmtQ4) CTL_LAUNCH_INTERSECT_L(ANY, CNT, 1, __VA_ARGS__);                                          \
        else CTL_LAUNCH_INTERSECT_L(ANY, CNT, 3, __VA_ARGS__);                                                                       \
    } while (0)
#else
#define CTL_LAUNCH_INTERSECT(ANY, CNT, ...)                                                                                          \
    do {                                                                                                                             \
        if (!S.flat_nodes) CTL_LAUNCH_INTERSECT_L(ANY, CNT, 0, __VA_ARGS__);                                                         \
        else if (S.flat_format == kFmtQ8) CTL_LAUNCH_INTERSECT_L(ANY, CNT, 4, __VA_ARGS__);                                          \
        else CTL_LAUNCH_INTERSECT_L(ANY, CNT, 1, __VA_ARGS__);                                                                       \
    } while (0)
#endif
void launch_intersect_closest(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node) {
    CTL_LAUNCH_INTERSECT(false, false, S, ro, rd, n_ptr, work, hit, hit_node, (uint32_t*)nullptr, (unsigned long long*)nullptr);
}
void launch_intersect_any(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, uint32_t* occ, float4* hit, int* hit_node) {
    CTL_LAUNCH_INTERSECT(true, false, S, ro, rd, n_ptr, work, hit, hit_node, occ, (unsigned long long*)nullptr);
}
void launch_intersect_pair(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node,
                           const float4* sro, const float4* srd, const uint32_t* sn_ptr, uint32_t* swork, uint32_t* occ) {
    if (!S.flat_nodes) {
        if (lc.alpha_test) hipLaunchKernelGGL((k_intersect_pair<0, true>), dim3(traversal_blocks(lc, 0, true)), dim3(kBlock), g_lds_pad, lc.stream, S, ro, rd, n_ptr, work, hit, hit_node, sro, srd, sn_ptr, swork, occ);
        else hipLaunchKernelGGL((k_intersect_pair<0, false>), dim3(traversal_blocks(lc, 0, false)), dim3(kBlock), g_lds_pad, lc.stream, S, ro, rd, n_ptr, work, hit, hit_node, sro, srd, sn_ptr, swork, occ);
#ifndef CTL_FLAT_EXPERIMENTS
    } else if (S.flat_format == kFmtQ8) {
        if (lc.alpha_test) hipLaunchKernelGGL((k_intersect_pair<4, true>), dim3(traversal_blocks(lc, 4, true)), dim3(kBlock), g_lds_pad, lc.stream, S, ro, rd, n_ptr, work, hit, hit_node, sro, srd, sn_ptr, swork, occ);
        else hipLaunchKernelGGL((k_intersect_pair<4, false>), dim3(traversal_blocks(lc, 4, false)), dim3(kBlock), g_lds_pad, lc.stream, S, ro, rd, n_ptr, work, hit, hit_node, sro, srd, sn_ptr, swork, occ);
#endif
    } else {
        if (lc.alpha_test) hipLaunchKernelGGL((k_intersect_pair<1, true>), dim3(traversal_blocks(lc, 1, true)), dim3(kBlock), g_lds_pad, lc.stream, S, ro, rd, n_ptr, work, hit, hit_node, sro, srd, sn_ptr, swork, occ);
        else hipLaunchKernelGGL((k_intersect_pair<1, false>), dim3(traversal_blocks(lc, 1, false)), dim3(kBlock), g_lds_pad, lc.stream, S, ro, rd, n_ptr, work, hit, hit_node, sro, srd, sn_ptr, swork, occ);
    }
}
void launch_intersect_count(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node,
                            uint32_t* occ, int any_hit, unsigned long long* counts3) {
    if (any_hit) CTL_LAUNCH_INTERSECT(true, true, S, ro, rd, n_ptr, work, hit, hit_node, occ, counts3);
    else CTL_LAUNCH_INTERSECT(false, true, S, ro, rd, n_ptr, work, hit, hit_node, (uint32_t*)nullptr, counts3);
}
// ---- material sort: counting sort of the path slots by the BSDF model they hit (16 buckets), between intersection and shading
__global__ __launch_bounds__(kBlock) void k_mat_count(dev_scene S, wave_queues Q, int depth) {
    __shared__ uint32_t h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = Q.counts[(depth - 1) * 4 + 0];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int tri = __float_as_int(Q.hit[i].w);
        uint32_t key = 0;
        if (tri >= 0) key = S.mats[S.node_info[Q.hit_node[i]].x + tri_mat_index(S, tri)].bsdf_type & 15u;
        Q.mat_key[i] = (unsigned char)key;
        atomicAdd(&h[key], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&Q.mat_counts[depth * 32 + threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(kBlock) void k_mat_scatter(wave_queues Q, int depth) {
    __shared__ uint32_t base[16];
    if (threadIdx.x == 0) { uint32_t s = 0; for (int k = 0; k < 16; k++) { base[k] = s; s += Q.mat_counts[depth * 32 + k]; } }
    __syncthreads();
    const uint32_t n = Q.counts[(depth - 1) * 4 + 0];
    const uint32_t n_round = (n + 63u) & ~63u;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_round; i += gridDim.x * kBlock) {
        const bool active = i < n;
        const uint32_t key = active ? Q.mat_key[i] : 0xffu;
        unsigned long long todo = __ballot(active);
        while (todo) {   // one atomic per model present in the wave
            const int leader = (int)__builtin_ctzll(todo);
            const uint32_t k = (uint32_t)__shfl((int)key, leader, 64);
            const unsigned long long mine = __ballot(active && key == k);
            uint32_t first = 0;
            if (lane == leader) first = atomicAdd(&Q.mat_counts[depth * 32 + 16 + k], (uint32_t)__popcll(mine));
            first = (uint32_t)__shfl((int)first, leader, 64);
            if (active && key == k) Q.order[base[k] + first + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull))] = i;
            todo &= ~mine;
        }
    }
}

// Model-class shading (shade_class_*.hip): split the vertices of a depth into one slot list per model class.  A workgroup takes windows of 16384 consecutive slots; inside a window the
// slots of a class are ordered by model (Q.mat_key, the byte per ray the closest-hit traversal left) and go to the class's list as ONE contiguous chunk (one atomic per class
// and window): a wave of a class launch is full, (mostly) of one model, and its 64 slots lie in one window of the queue (1 MB of path state per array) — the path-state reads stay as local as the in-kernel regrouping of
// k_shade_full keeps them, without its barriers and without waves that idle because their slots belong to another class.
#ifndef CTL_PART_PER_LANE
#define CTL_PART_PER_LANE 16   // slots per lane of a partition workgroup = windows of 16384 slots (synthetic-bathroom shade ms per pass: 1024-slot windows 2.94, 2048: 2.77, 4096: 2.62, 8192: 2.59, 16384: 2.55, 32768: 2.53)
#endif
constexpr int kPartBlock = 1024, kPartPerLane = CTL_PART_PER_LANE;
__device__ __forceinline__ uint32_t class_of_key(uint32_t k) { return ((CTL_CLASS_A_KEYS >> k) & 1u) ? 0u : (((CTL_CLASS_B_KEYS >> k) & 1u) ? 1u : (((CTL_CLASS_P_KEYS >> k) & 1u) ? 3u : (((CTL_CLASS_G_KEYS >> k) & 1u) ? 4u : 2u))); }
__global__ __launch_bounds__(kPartBlock) void k_class_partition(wave_queues Q, int depth) {
    __shared__ uint32_t s_hist[16], s_start[16];
    const uint32_t n = Q.counts[(depth - 1) * 4 + 0];
    constexpr uint32_t kWin = kPartBlock * kPartPerLane;
    for (uint32_t w0 = blockIdx.x * kWin; w0 < n; w0 += gridDim.x * kWin) {
        if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
        __syncthreads();
        uint32_t key[kPartPerLane], rank[kPartPerLane];
#pragma unroll
        for (int k = 0; k < kPartPerLane; k++) {
            const uint32_t j = w0 + k * kPartBlock + threadIdx.x;
            key[k] = j < n ? (uint32_t)Q.mat_key[j] & 15u : 16u;
            // one LDS atomic per model present in the wave (ballot loop), not one per lane
            rank[k] = 0;
            unsigned long long todo = __ballot(key[k] < 16u);
            while (todo) {
                const uint32_t kk = (uint32_t)__shfl((int)key[k], (int)__builtin_ctzll(todo), 64);
                const unsigned long long mine = __ballot(key[k] == kk);
                uint32_t first = 0;
                if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(mine)) first = atomicAdd(&s_hist[kk], (uint32_t)__popcll(mine));
                first = (uint32_t)__shfl((int)first, (int)__builtin_ctzll(mine), 64);
                if (key[k] == kk) rank[k] = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0));
                todo &= ~mine;
            }
        }
        __syncthreads();
        if (threadIdx.x < 5) {   // thread c: the chunk of class c in its list, the models of the class in key order inside it
            uint32_t tot = 0;
            for (uint32_t k = 0; k < 16; k++) if (class_of_key(k) == threadIdx.x) tot += s_hist[k];
            uint32_t at = tot ? atomicAdd(&Q.mat_counts[depth * 32 + 24 + threadIdx.x], tot) : 0u;
            for (uint32_t k = 0; k < 16; k++) if (class_of_key(k) == threadIdx.x) { s_start[k] = at; at += s_hist[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPartPerLane; k++)
            if (key[k] < 16u) Q.class_order[class_of_key(key[k])][s_start[key[k]] + rank[k]] = w0 + k * kPartBlock + threadIdx.x;
        __syncthreads();
    }
}
void launch_class_partition(const launch_ctx& lc, const wave_queues& Q, int depth) {
    hipLaunchKernelGGL(k_class_partition, dim3(lc.grid_blocks / 4), dim3(kPartBlock), 0, lc.stream, Q, depth);
}

// the shade kernel exists in feature-specialised builds (shade_basic.hip / shade_full.hip): a scene that uses only the basic
// material / light / texture set runs the variant whose code does not carry the registers of the rest (dev_scene::shade_features)
void launch_shade(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image) {
    if (S.shade_features == 0) { if (P.wavefront_rules) launch_shade_basic_wf(lc, S, Q, P, depth, image); else launch_shade_basic(lc, S, Q, P, depth, image); return; }
    if (P.model_classes) {   // one launch per model class the scene has (class a also takes the misses): shade_class_*.hip
        launch_class_partition(lc, Q, depth);
        (P.wavefront_rules ? launch_shade_class_a_wf : launch_shade_class_a)(lc, S, Q, P, depth, image);
        if (S.shade_models & CTL_CLASS_B_KEYS) (P.wavefront_rules ? launch_shade_class_b_wf : launch_shade_class_b)(lc, S, Q, P, depth, image);
        if (S.shade_models & CTL_CLASS_G_KEYS) (P.wavefront_rules ? launch_shade_class_g_wf : launch_shade_class_g)(lc, S, Q, P, depth, image);
        if (S.shade_models & CTL_CLASS_P_KEYS) (P.wavefront_rules ? launch_shade_class_p_wf : launch_shade_class_p)(lc, S, Q, P, depth, image);
        if (S.shade_models & CTL_CLASS_C_KEYS) (P.wavefront_rules ? launch_shade_class_c_wf : launch_shade_class_c)(lc, S, Q, P, depth, image);
        return;
    }
    if (P.sort_materials) {
        hipLaunchKernelGGL(k_mat_count, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, S, Q, depth);
        hipLaunchKernelGGL(k_mat_scatter, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, Q, depth);
    }
    if (P.wavefront_rules) launch_shade_full_wf(lc, S, Q, P, depth, image); else launch_shade_full(lc, S, Q, P, depth, image);
}
void launch_finalize(const launch_ctx& lc, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image) {
    hipLaunchKernelGGL(k_finalize, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, Q, P, depth, image);
}
void launch_accumulate_stats(const launch_ctx& lc, const wave_queues& Q, int max_depth) {
    hipLaunchKernelGGL(k_accumulate_stats, dim3(1), dim3(64), 0, lc.stream, Q, max_depth);
}
void launch_apply_pipeline(const launch_ctx& lc, const ctl_pixel_data* image, uint32_t n, float splat_scale, uint32_t* rgbcol_out) {
    hipLaunchKernelGGL(k_apply_pipeline, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, image, n, splat_scale, rgbcol_out);
}
void launch_resolve_rgb(const launch_ctx& lc, const ctl_pixel_data* image, uint32_t n, float splat_scale, float* rgb_out) {
    hipLaunchKernelGGL(k_resolve_rgb, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, image, n, splat_scale, rgb_out);
}

} // namespace ctl
