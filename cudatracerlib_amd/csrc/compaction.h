// compaction.h — wave- and workgroup-level queue appends (stream compaction) and the framebuffer accumulation shared by the
// ray-generation, shade and finalize kernels.
#pragma once
#include "kernels.h"

namespace ctl {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------ wave primitives
// Append one element per participating lane to a global queue: one atomic per wave (ballot + mbcnt prefix).
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool take) {
    const unsigned long long mask = __ballot(take);
    if (mask == 0) return 0;
    const uint32_t n = (uint32_t)__popcll(mask);
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
    uint32_t base = 0;
    const int leader = (int)__builtin_ctzll(mask);
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader, 64);
    return base + prefix;
}

// Append to up to three global queues from a whole workgroup with ONE atomic per queue per workgroup: ballot/mbcnt inside
// each wave, wave totals through LDS, thread k < 3 reserves the block's range.  A single queue cursor saturates at
// ~88 returning atomics/us on MI355X (MI355X_MICROARCH.md "dequeue"), which a per-wave append hits at once: 160 k waves per
// pass step on three cursors were the whole cost of the first shade kernel.  Must be called by every thread of the block.
constexpr int kWideBlock = 1024;
struct block_slots { uint32_t s[3]; };
__device__ __forceinline__ block_slots block_append3(uint32_t* c0, bool t0, uint32_t* c1, bool t1, uint32_t* c2, bool t2, uint32_t (*s_cnt)[kWideBlock / 64], uint32_t* s_base) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const unsigned long long m0 = __ballot(t0), m1 = __ballot(t1), m2 = __ballot(t2);
    if (lane == 0) { s_cnt[0][wave] = (uint32_t)__popcll(m0); s_cnt[1][wave] = (uint32_t)__popcll(m1); s_cnt[2][wave] = (uint32_t)__popcll(m2); }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t* ctr = threadIdx.x == 0 ? c0 : (threadIdx.x == 1 ? c1 : c2);
        uint32_t tot = 0;
        for (int w = 0; w < n_waves; w++) { const uint32_t c = s_cnt[threadIdx.x][w]; s_cnt[threadIdx.x][w] = tot; tot += c; }   // exclusive prefix over waves
        s_base[threadIdx.x] = (tot && ctr) ? atomicAdd(ctr, tot) : 0u;
    }
    __syncthreads();
    block_slots r;
    r.s[0] = s_base[0] + s_cnt[0][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0));
    r.s[1] = s_base[1] + s_cnt[1][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0));
    r.s[2] = s_base[2] + s_cnt[2][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0));
    __syncthreads();   // s_cnt / s_base are reused by the next iteration
    return r;
}

// The same append with queues 0 and 1 bucketed by a 3-bit key (the octant of the new ray's direction): inside the block's range the
// entries come out grouped by key, bucket after bucket, so that the 64 rays a traversal wave picks up next mostly share the octant — they
// walk the tree in the same order and touch the same node lines.  s_cnt has 17 rows (2 x 8 buckets + the plain queue).
__device__ __forceinline__ block_slots block_append3_keyed(uint32_t* c0, bool t0, uint32_t k0, uint32_t* c1, bool t1, uint32_t k1, uint32_t* c2, bool t2,
                                                           uint32_t (*s_cnt)[kWideBlock / 64], uint32_t* s_base) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    uint32_t rank0 = 0, rank1 = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const unsigned long long a = __ballot(t0 && k0 == k), b = __ballot(t1 && k1 == k);
        if (lane == 0) { s_cnt[k][wave] = (uint32_t)__popcll(a); s_cnt[8 + k][wave] = (uint32_t)__popcll(b); }
        if (t0 && k0 == k) rank0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a, 0));
        if (t1 && k1 == k) rank1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
    }
    const unsigned long long m2 = __ballot(t2);
    if (lane == 0) s_cnt[16][wave] = (uint32_t)__popcll(m2);
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t* ctr = threadIdx.x == 0 ? c0 : (threadIdx.x == 1 ? c1 : c2);
        const int row0 = threadIdx.x == 2 ? 16 : (int)threadIdx.x * 8, rows = threadIdx.x == 2 ? 1 : 8;
        uint32_t tot = 0;
        for (int r = row0; r < row0 + rows; r++)          // bucket-major exclusive prefix: all waves of bucket 0, then bucket 1, ...
            for (int w = 0; w < n_waves; w++) { const uint32_t c = s_cnt[r][w]; s_cnt[r][w] = tot; tot += c; }
        s_base[threadIdx.x] = (tot && ctr) ? atomicAdd(ctr, tot) : 0u;
    }
    __syncthreads();
    block_slots r;
    r.s[0] = s_base[0] + s_cnt[t0 ? k0 : 0][wave] + rank0;
    r.s[1] = s_base[1] + s_cnt[8 + (t1 ? k1 : 0)][wave] + rank1;
    r.s[2] = s_base[2] + s_cnt[16][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0));
    __syncthreads();
    return r;
}
__device__ __forceinline__ uint32_t octant_of(f3 d) { return (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u); }

// ------------------------------------------------------------------------------------------------ framebuffer
// Image::AddSample (Engine/Image.cu:22-44): clamp negatives, drop NaN/Inf, floor to the pixel, 4 float atomics.
__device__ __forceinline__ void add_sample(ctl_pixel_data* img, uint32_t W, uint32_t H, float sx, float sy, f3 L) {
    L = f3(max2(0.0f, L.x), max2(0.0f, L.y), max2(0.0f, L.z));
    const int x = (int)floorf(sx), y = (int)floorf(sy);
    const bool bad = !(isfinite(L.x) && isfinite(L.y) && isfinite(L.z));
    if (x < 0 || x >= (int)W || y < 0 || y >= (int)H || bad) return;
    ctl_pixel_data* r = img + ((size_t)y * W + x);
    atomicAdd(&r->rgb[0], L.x); atomicAdd(&r->rgb[1], L.y); atomicAdd(&r->rgb[2], L.z); atomicAdd(&r->weight_sum, 1.0f);
}

// AddSample of a finished path of pass `pass_b` of the batch whose own pixel is `pixel` (kernels.h pass_params::stage).  pX = pixel + jitter rounds into the NEXT pixel once in ~10^4
// samples (1919 + 0.99999994f is 1920.0f) — as in the reference; such a sample is added atomically, it is not the only one its landing pixel gets in this pass.
template <class PARAMS>
__device__ __forceinline__ void add_sample_ordered(const PARAMS& P, ctl_pixel_data* img, uint32_t pixel, uint32_t pass_b, float sx, float sy, f3 L) {
    if (!P.stage) { add_sample(img, P.width, P.height, sx, sy, L); return; }
    L = f3(max2(0.0f, L.x), max2(0.0f, L.y), max2(0.0f, L.z));
    const int x = (int)floorf(sx), y = (int)floorf(sy);
    const bool bad = !(isfinite(L.x) && isfinite(L.y) && isfinite(L.z));
    if (x < 0 || x >= (int)P.width || y < 0 || y >= (int)P.height || bad) return;
    const uint32_t idx = (uint32_t)y * P.width + (uint32_t)x;
    // the stage holds THIS RANK'S pixels only (stage_stride = its 64 x 64 tiles x 4096): slot = local tile * 4096 + row-major position in the tile (k_resolve_stage, kernels.hip, inverts it)
    const uint32_t tile = ((uint32_t)y >> 6) * ((P.width + 63u) >> 6) + ((uint32_t)x >> 6);
    if (idx == pixel) P.stage[(size_t)pass_b * P.stage_stride + (size_t)(tile / P.tile_world) * 4096u + (((uint32_t)y & 63u) << 6) + ((uint32_t)x & 63u)] = make_float4(L.x, L.y, L.z, 1.0f);
    else { ctl_pixel_data* r = img + idx; atomicAdd(&r->rgb[0], L.x); atomicAdd(&r->rgb[1], L.y); atomicAdd(&r->rgb[2], L.z); atomicAdd(&r->weight_sum, 1.0f); }
}

} // namespace ctl
