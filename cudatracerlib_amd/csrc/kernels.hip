// kernels.hip — gfx950 kernels of the wavefront path tracer: ray generation, BVH traversal + Woop intersection,
// shading (BSDF sample/eval + next-event estimation + Russian roulette) with wave-level stream compaction, and
// framebuffer accumulation.  The per-path semantics are those of the reference's PathTrace<DIRECT>
// (Integrators/PathTracer.cu:10-113) re-cut at its two trace points; see DESIGN.md "Kernels".
#include "kernels.h"
#include "traverse.h"
#include "shading.h"
#include <cstdlib>

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
        const bool valid = x < P.width && y < P.height;
        const uint32_t slot = block_append3(&Q.counts[0], valid, nullptr, false, nullptr, false, s_cnt, s_base).s[0];
        if (!valid) continue;
        const uint32_t pixel = y * P.width + x;
        sampler rng{ P.t1 + pass_b * n1, P.t2 + pass_b * n1, pixel, 0, 0 };
        const f2 j = rng.next2();
        const f2 pX{ (float)x + j.x, (float)y + j.y };
        (void)rng.next2();   // aperture sample, unused by the perspective sensor but drawn (PathTracer.cu:190)
        f3 o, d; sensor_sample_ray(S.cam, pX, o, d);
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

// ------------------------------------------------------------------------------------------------ intersection
// Persistent waves with lane refill and an LDS traversal stack — see traverse.h (the reference's g_warpCounter pool,
// Kernel/TraceHelper.cu:386-399, re-derived for 64-wide waves).
template <bool ANY_HIT, bool COUNT, bool FLAT>
__global__ __launch_bounds__(kBlock) void k_intersect(dev_scene S, const float4* __restrict__ ro, const float4* __restrict__ rd, const uint32_t* __restrict__ n_ptr,
                                                       uint32_t* __restrict__ work, float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ,
                                                       unsigned long long* __restrict__ counts3) {
    __shared__ int lds_stack[(FLAT ? kLdsStackFlat : kLdsStack) * kBlock];
    const uint32_t n = *n_ptr;
    trav_counts tc{ 0, 0, 0, 0, 0 };
    if (FLAT) intersect_flat<ANY_HIT, COUNT>(S, ro, rd, n, work, hit, hit_node, occ, lds_stack, tc);
    else intersect_persistent<ANY_HIT, COUNT>(S, ro, rd, n, work, hit, hit_node, occ, lds_stack, tc);
    if (COUNT) {
        atomicAdd(&counts3[0], (unsigned long long)tc.n_inner); atomicAdd(&counts3[1], (unsigned long long)tc.n_tri); atomicAdd(&counts3[2], (unsigned long long)tc.n_inst);
        atomicAdd(&counts3[3], (unsigned long long)tc.w_inner); atomicAdd(&counts3[4], (unsigned long long)tc.w_tri);
    }
}

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

// ------------------------------------------------------------------------------------------------ shading
// One lane = one queued path vertex.  `depth` is the megakernel's 1-based depth of this vertex.
__global__ __launch_bounds__(kWideBlock) void k_shade(dev_scene S, wave_queues Q, pass_params P, int depth, ctl_pixel_data* __restrict__ image) {
    __shared__ uint32_t s_cnt[3][kWideBlock / 64]; __shared__ uint32_t s_base[3];
    const int cur = (depth - 1) & 1, nxt = depth & 1;
    const path_soa& A = Q.path[cur];
    const path_soa& B = Q.path[nxt];
    const uint32_t n = Q.counts[(depth - 1) * 4 + 0];
    uint32_t* n_next = &Q.counts[depth * 4 + 0];
    uint32_t* n_shadow = &Q.counts[depth * 4 + 1];
    uint32_t* n_final = &Q.counts[depth * 4 + 2];
    const uint32_t* occ_prev = Q.sh_occ[(depth - 1) & 1];
    float4* sh_o = Q.sh_o[depth & 1]; float4* sh_d = Q.sh_d[depth & 1];

    const uint32_t n_round = (n + (kWideBlock - 1u)) & ~(kWideBlock - 1u);   // whole workgroups iterate together (barriers in block_append3)
    for (uint32_t i = blockIdx.x * kWideBlock + threadIdx.x; i < n_round; i += gridDim.x * kWideBlock) {
        const bool active = i < n;
        bool alive = false, want_shadow = false, terminated = false;
        f3 cl(0.0f), cf(0.0f), directF(0.0f), new_o(0.0f), new_d(0.0f), last_nor(0.0f), sh_org(0.0f), sh_dir(0.0f);
        float bsdf_pdf_out = 0.0f, sh_tmax = 0.0f; uint32_t pixel = 0, d1 = 0, d2 = 0, pass_b = 0; bool specular = false; float2 px = make_float2(0.f, 0.f);
        if (active) {
            const float4 ro = A.ray_o[i], rd = A.ray_d[i], thr = A.thr[i], rad = A.rad[i], nor = A.nor[i], pend = A.pend[i];
            const float4 hit = Q.hit[i]; const int hnode = Q.hit_node[i];
            px = A.px[i];
            pixel = __float_as_uint(rad.w);
            const uint32_t packed = __float_as_uint(nor.w);
            pass_b = (packed >> 16) & 0xffu;
            const uint32_t n1 = CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH;
            sampler rng{ P.t1 + pass_b * n1, P.t2 + pass_b * n1, pixel, packed & 0xffu, (packed >> 8) & 0xffu };
            bool specularBounce = ((packed >> 24) & kFlagSpecular) != 0;
            cl = f3(rad.x, rad.y, rad.z); cf = f3(thr.x, thr.y, thr.z);
            float brdf_scattering_pdf = thr.w;
            last_nor = f3(nor.x, nor.y, nor.z);
            // next-event estimation of the previous vertex: add it now that its shadow ray is resolved
            // (deferred `cl += cf * UniformSampleOneLight(...)`, PathTracer.cu:81-82)
            const uint32_t sidx = __float_as_uint(pend.w);
            if (sidx != kNoShadow && !occ_prev[sidx]) cl = cl + f3(pend.x, pend.y, pend.z);
            const f3 r_o(ro.x, ro.y, ro.z), r_d(rd.x, rd.y, rd.z);
            const int tri = __float_as_int(hit.w);
            if (tri >= 0) {
                // TraceResult::getBsdfSample (Kernel/TraceResult.cu:11-43)
                bsdf_rec b;
                b.eta = 1.0f; b.sampled_type = 0; b.type_mask = kEAll;
                b.dg.P = r_o + hit.x * r_d;
                fill_dg(S, hit.y, hit.z, tri, hnode, b.dg);
                b.wi = b.dg.sys.to_local(-r_d);
                const uint4 ninfo = S.node_info[hnode];
                const ctl_material& mat = S.mats[ninfo.x + tri_mat_index(S, tri)];
                if (mat.two_sided && b.wi.z < 0) { b.dg.n = -b.dg.n; b.dg.sys.n = -b.dg.sys.n; b.wi.z *= -1.0f; }
                // emission, MIS-weighted against NEE of the previous vertex (PathTracer.cu:64-77)
                const uint32_t nli = mat.node_light_index;
                if (nli != 0xffffffffu) {
                    const uint32_t li = nli == 0 ? ninfo.y : ninfo.z;
                    const ctl_light& light = S.lights[li];
                    float misWeight = 1.0f;
                    if (!(!P.direct || depth == 1 || specularBounce)) {
                        const float direct_pdf = light_pdf_direct(light, r_d, last_nor, b.dg.n, hit.x) * pdf_emitter(S, li);
                        misWeight = power_heuristic(brdf_scattering_pdf, direct_pdf);
                    }
                    cl = cl + misWeight * cf * light_eval(light, b.dg.sys.n, -r_d);
                }
                const f3 f = bsdf_sample(mat, b, brdf_scattering_pdf, rng.next2());
                last_nor = b.dg.sys.n;
                if (P.direct && (mat.combined_type & kESmooth)) {
                    // UniformSampleOneLight + EstimateDirect (Kernel/TraceAlgorithms.cu:44-101), occlusion deferred
                    if (S.num_lights) {
                        const f2 sl = rng.next2();
                        float lpdf; const int li = sample_emitter(S, lpdf, sl.x);
                        if (li >= 0) {
                            direct_rec dr; dr.ref = b.dg.P; dr.refN = b.dg.sys.n;
                            const f3 value = light_sample_direct(S, S.lights[li], dr, rng.next2());
                            if (!is_zero(value)) {
                                bsdf_rec b2 = b;
                                b2.wo = b.dg.sys.to_local(dr.d); b2.type_mask = kEAll & ~kEDelta;
                                const f3 bsdfVal = bsdf_f(mat, b2);
                                if (!is_zero(bsdfVal)) {
                                    float weight = 1.0f;
                                    if (dr.measure != kMeasureDiscrete) {
                                        const float bp = bsdf_pdf(mat, b2);
                                        const float directPdf = (dr.measure == kMeasureArea ? dr.pdf * dr.dist / fabsf(dot(dr.n, dr.d)) : dr.pdf) * lpdf;
                                        weight = power_heuristic(directPdf, bp);
                                    }
                                    directF = cf * ((value * bsdfVal * weight) / lpdf);
                                    want_shadow = true;
                                    sh_org = dr.ref; sh_dir = dr.d; sh_tmax = dr.dist - S.eps;   // Occluded(r, 0, dist) (KernelDynamicScene.cu:70-80)
                                }
                            }
                        }
                    }
                }
                specularBounce = (b.sampled_type & kEDelta) != 0;
                cf = cf * f;
                new_o = b.dg.P; new_d = b.dg.sys.to_world(b.wo);
                alive = true;
                // a path whose throughput became exactly zero cannot contribute any more (the reference keeps tracing it
                // until Russian roulette removes it; radiance is identical)
                if (is_zero(cf)) alive = false;
                if (alive && depth >= P.max_path_length) alive = false;   // while (depth++ < maxPathLength)
                if (alive && depth > P.rr_start_depth && !specularBounce) {   // PathTracer.cu:91-96
                    const float q = max3c(cf);
                    if (rng.next1() >= q) alive = false; else cf = cf / q;
                }
            }
            else if (S.env_map_index != 0xffffffffu) {
                // miss with an environment emitter (PathTracer.cu:99-111), MIS-weighted against NEE of the previous vertex
                const ctl_light& light = S.lights[S.env_map_index];
                float misWeight = 1.0f;
                if (!(!P.direct || depth == 1 || specularBounce)) {
                    const float direct_pdf = env_pdf_direct(S, light, r_d) * pdf_emitter(S, S.env_map_index);
                    misWeight = power_heuristic(brdf_scattering_pdf, direct_pdf);
                }
                cl = cl + misWeight * cf * env_eval(S, light, r_d);
            }
            terminated = !alive;
            specular = specularBounce; bsdf_pdf_out = brdf_scattering_pdf; d1 = rng.d1; d2 = rng.d2;
        }
        // ---- stream compaction: survivors, shadow rays and waiting terminations are appended densely, one atomic per wave
        const block_slots bs = block_append3(n_shadow, want_shadow, n_next, alive, n_final, terminated && want_shadow, s_cnt, s_base);
        const uint32_t sslot = bs.s[0], nslot = bs.s[1], fslot = bs.s[2];
        if (!active) continue;
        if (want_shadow) { sh_o[sslot] = make_float4(sh_org.x, sh_org.y, sh_org.z, S.eps); sh_d[sslot] = make_float4(sh_dir.x, sh_dir.y, sh_dir.z, sh_tmax); }
        if (alive) {
            B.ray_o[nslot] = make_float4(new_o.x, new_o.y, new_o.z, S.eps);
            B.ray_d[nslot] = make_float4(new_d.x, new_d.y, new_d.z, 3.402823466e+38f);
            B.thr[nslot] = make_float4(cf.x, cf.y, cf.z, bsdf_pdf_out);
            B.rad[nslot] = make_float4(cl.x, cl.y, cl.z, __uint_as_float(pixel));
            B.nor[nslot] = make_float4(last_nor.x, last_nor.y, last_nor.z, __uint_as_float((d1 % CTL_SAMPLER_SEQUENCE_LENGTH) | ((d2 % CTL_SAMPLER_SEQUENCE_LENGTH) << 8) | (pass_b << 16)   /* only d mod 30 matters (Sampler_device.h:98,104) */ | ((specular ? kFlagSpecular : 0u) << 24)));
            B.pend[nslot] = make_float4(directF.x, directF.y, directF.z, __uint_as_float(want_shadow ? sslot : kNoShadow));
            B.px[nslot] = px;
        } else if (want_shadow) {
            Q.fin.rad[fslot] = make_float4(cl.x, cl.y, cl.z, __uint_as_float(sslot));
            Q.fin.dir[fslot] = make_float4(directF.x, directF.y, directF.z, 0.0f);
            Q.fin.px[fslot] = px;
        } else {
            add_sample(image, P.width, P.height, px.x, px.y, cl);   // img.AddSample(pX.x, pX.y, col) (PathTracer.cu:192)
        }
    }
}

// terminated paths whose last NEE shadow ray has now been traced
__global__ __launch_bounds__(kBlock) void k_finalize(wave_queues Q, pass_params P, int depth, ctl_pixel_data* __restrict__ image) {
    const uint32_t n = Q.counts[depth * 4 + 2];
    const uint32_t* occ = Q.sh_occ[depth & 1];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 rad = Q.fin.rad[i], dir = Q.fin.dir[i]; const float2 px = Q.fin.px[i];
        f3 cl(rad.x, rad.y, rad.z);
        if (!occ[__float_as_uint(rad.w)]) cl = cl + f3(dir.x, dir.y, dir.z);
        add_sample(image, P.width, P.height, px.x, px.y, cl);
    }
}

// rays of a pass = sum over bounces of (path rays + shadow rays)  (Kernel/TraceHelper.cu:176,745)
__global__ void k_accumulate_stats(wave_queues Q, int max_depth) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long r = 0, s = 0;
        for (int d = 0; d <= max_depth; d++) { r += (unsigned long long)Q.counts[d * 4 + 0]; s += (unsigned long long)Q.counts[d * 4 + 1]; }
        Q.stats[0] += r; Q.stats[1] += s;
    }
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

void apply_tuning_from_env() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char* e = getenv("CTL_REFILL_IDLE")) { int v = atoi(e); if (v >= 1 && v <= 64) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_refill_idle), &v, sizeof(v)); }
    if (const char* e = getenv("CTL_ANY_SORTED")) { int v = atoi(e) != 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_any_sorted), &v, sizeof(v)); }
    if (const char* e = getenv("CTL_TRI_BATCH")) { int v = atoi(e); if (v >= 1 && v <= 64) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tri_batch), &v, sizeof(v)); }
}

// ------------------------------------------------------------------------------------------------ launch wrappers
void launch_raygen(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P) {
    hipLaunchKernelGGL(k_raygen, dim3(lc.grid_blocks / 4), dim3(kWideBlock), 0, lc.stream, S, Q, P);
}
#define CTL_LAUNCH_INTERSECT(ANY, CNT, ...)                                                                                          \
    do {                                                                                                                             \
        if (S.flat_nodes) hipLaunchKernelGGL((k_intersect<ANY, CNT, true>), dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((k_intersect<ANY, CNT, false>), dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, __VA_ARGS__);            \
    } while (0)
void launch_intersect_closest(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node) {
    CTL_LAUNCH_INTERSECT(false, false, S, ro, rd, n_ptr, work, hit, hit_node, (uint32_t*)nullptr, (unsigned long long*)nullptr);
}
void launch_intersect_any(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, uint32_t* occ, float4* hit, int* hit_node) {
    CTL_LAUNCH_INTERSECT(true, false, S, ro, rd, n_ptr, work, hit, hit_node, occ, (unsigned long long*)nullptr);
}
void launch_intersect_count(const launch_ctx& lc, const dev_scene& S, const float4* ro, const float4* rd, const uint32_t* n_ptr, uint32_t* work, float4* hit, int* hit_node,
                            uint32_t* occ, int any_hit, unsigned long long* counts3) {
    if (any_hit) CTL_LAUNCH_INTERSECT(true, true, S, ro, rd, n_ptr, work, hit, hit_node, occ, counts3);
    else CTL_LAUNCH_INTERSECT(false, true, S, ro, rd, n_ptr, work, hit, hit_node, (uint32_t*)nullptr, counts3);
}
void launch_shade(const launch_ctx& lc, const dev_scene& S, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image) {
    hipLaunchKernelGGL(k_shade, dim3(lc.grid_blocks / 8), dim3(kWideBlock), 0, lc.stream, S, Q, P, depth, image);   // one 16-wave workgroup per CU (122 VGPRs)
}
void launch_finalize(const launch_ctx& lc, const wave_queues& Q, const pass_params& P, int depth, ctl_pixel_data* image) {
    hipLaunchKernelGGL(k_finalize, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, Q, P, depth, image);
}
void launch_accumulate_stats(const launch_ctx& lc, const wave_queues& Q, int max_depth) {
    hipLaunchKernelGGL(k_accumulate_stats, dim3(1), dim3(64), 0, lc.stream, Q, max_depth);
}
void launch_resolve_rgb(const launch_ctx& lc, const ctl_pixel_data* image, uint32_t n, float splat_scale, float* rgb_out) {
    hipLaunchKernelGGL(k_resolve_rgb, dim3(lc.grid_blocks), dim3(kBlock), 0, lc.stream, image, n, splat_scale, rgb_out);
}

} // namespace ctl
