// megakernel.hip — the "PathTracer" plugin: the reference's megakernel integrator (Integrators/PathTracer.cu: pathKernel2
// :182-194 looping PathTrace<DIRECT> :10-113) as ONE kernel, one lane = one (pixel, pass) path from the camera to termination.
// Same device functions as the wavefront tracer (shading.h), same sampler draws in the same order, shadow rays resolved inline.
// It exists for A/B comparison behind the same plugin API; the wavefront formulation is the fast one.
#include "kernels.h"
#define CTL_TEX_PARTIALS 1   // this integrator computes ray differentials at the first hit (PathTracer.cu:60-61) and filters image textures there
#include "shading.h"
#include "compaction.h"
#include "tracer.h"
#ifdef CTL_FLAT_EXPERIMENTS
#include "experiments/traverse_flat_variants.h"
#else
#include "traverse_flat.h"
#include "traverse_flat8.h"
#endif
#include "mitsuba_loader.h"   // unsupported_error
#include <climits>

namespace ctl {

// single-ray traversal of the flattened BVH (traverse_flat8.h for the 8-wide format, traverse_flat.h for the 4-wide one): closest hit, or any hit in (tmin, tmax).
// ONE LDS array serves both instantiations: a lane is never in a closest-hit and an any-hit search at the same time (20 KiB per 256-lane workgroup, not 40).
__device__ __forceinline__ lds_int* single_stack_column() {
    __shared__ int s_stack[kSingleLdsRows * 256];   // one column per lane of the 256-lane workgroup
    return (lds_int*)s_stack + threadIdx.x;
}
template <bool ANY_HIT>
__device__ __noinline__ bool trace_single(const dev_scene& S, f3 o, f3 d, float tmin, float tmax, float& ht, float& hu, float& hv, int& htri, int& hnode) {
#ifndef CTL_FLAT_EXPERIMENTS
    if (S.flat_format == kFmtQ8) return trace_single_flat8<ANY_HIT, true>(S, single_stack_column(), o, d, tmin, tmax, ht, hu, hv, htri, hnode);
#endif
    return trace_single_flat<ANY_HIT, true>(S, single_stack_column(), o, d, tmin, tmax, ht, hu, hv, htri, hnode);
}

// pathKernel2<DIRECT> + PathTrace<DIRECT> (Integrators/PathTracer.cu:182-194, 10-113), no participating media
#ifndef CTL_MEGA_WAVES
#define CTL_MEGA_WAVES 2
#endif
__global__ __launch_bounds__(256, CTL_MEGA_WAVES) void k_path_trace(dev_scene S, pass_params P, ctl_pixel_data* __restrict__ image, unsigned long long* __restrict__ ray_count) {
    const uint32_t tiles_x = (P.width + 63) / 64;
    const uint32_t n_total = P.n_local_pixels * P.batch;
    const uint32_t n1 = CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH;
    unsigned long long rays = 0;
    for (uint32_t gi = blockIdx.x * 256u + threadIdx.x; gi < n_total; gi += gridDim.x * 256u) {
        const uint32_t pass_b = gi / P.n_local_pixels, li = gi - pass_b * P.n_local_pixels;
        const uint32_t tile = P.tile_rank + (li >> 12) * P.tile_world, p = li & 4095u, micro = p >> 6, ln = p & 63u;
        const uint32_t x_ = (tile % tiles_x) * 64 + (micro & 7u) * 8 + (ln & 7u), y_ = (tile / tiles_x) * 64 + (micro >> 3) * 8 + (ln >> 3);
        if (P.debug_out && gi != 0) break;
        const uint32_t x = P.debug_out ? P.debug_x : x_, y = P.debug_out ? P.debug_y : y_;
        if (x >= P.width || y >= P.height) continue;
        sampler rng{ P.t1 + pass_b * n1, P.t2 + pass_b * n1, y * P.width + x, 0, 0 };
        // PathTracer::DebugInternal (PathTracer.cu:172-180) starts at the pixel's own position and draws the aperture sample only
        const f2 j = P.debug_out ? f2{ 0.0f, 0.0f } : rng.next2();
        const f2 pX{ (float)x + j.x, (float)y + j.y };
        const f2 ap = rng.next2();   // aperture sample
        f3 r_o, r_d, r_ox, r_dx, r_oy, r_dy; sensor_sample_ray_differential(S.cam, pX, ap, r_o, r_d, r_ox, r_dx, r_oy, r_dy);   // pathKernel2: sampleSensorRay(r, rX, rY, ...) (PathTracer.cu:186-190)
        f3 cl(0.0f), cf(1.0f), last_nor(0.0f);
        int depth = 0; bool specularBounce = false, had_hit = false;
        float brdf_scattering_pdf = 0;
        while (depth++ < P.max_path_length) {
            float t, u, v; int tri, node;
            had_hit = trace_single<false>(S, r_o, r_d, S.eps, 3.402823466e+38f, t, u, v, tri, node);
            rays++;
            if (!had_hit) break;
            bsdf_rec b; b.eta = 1.0f; b.sampled_type = 0; b.type_mask = kEAll;
            b.dg.P = r_o + t * r_d;
            fill_dg(S, u, v, tri, node, b.dg);
            b.wi = b.dg.sys.to_local(-r_d);
            const uint4 ninfo = S.node_info[node];
            const ctl_material& mat = S.mats[ninfo.x + tri_mat_index(S, tri)];
            if (mat.map_kind != CTL_MAP_NONE) sample_normal_map(mat, b.dg);
            if (mat.two_sided && b.wi.z < 0) { b.dg.n = -b.dg.n; b.dg.sys.n = -b.dg.sys.n; b.wi.z *= -1.0f; }
            if (depth == 1) compute_partials(b.dg, r_ox, r_dx, r_oy, r_dy);   // PathTracer.cu:60-61
            const uint32_t nli = mat.node_light_index;
            if (nli != 0xffffffffu) {
                const uint32_t li2 = nli == 0 ? ninfo.y : ninfo.z;
                const ctl_light& light = scene_lights(S)[li2];
                float misWeight = 1.0f;
                if (!(!P.direct || depth == 1 || specularBounce)) misWeight = power_heuristic(brdf_scattering_pdf, light_pdf_direct(light, r_d, last_nor, b.dg.n, t) * pdf_emitter(S, li2));
                cl = cl + misWeight * cf * light_eval(S, light, b.dg.P, b.dg.sys.n, -r_d);
            }
            const f3 f = bsdf_sample_top(mat, b, brdf_scattering_pdf, rng.next2());
            last_nor = b.dg.sys.n;
            if (P.direct && (mat.combined_type & kESmooth)) cl = cl + cf * f3(0.0f);   // `cl += cf * UniformSampleOneLight(...)` with a zero estimate (PathTracer.cu:81-82): + 0, or NaN for a NaN / infinite throughput (shade_kernel.inc has the why)
            if (P.direct && (mat.combined_type & kESmooth) && S.num_lights) {   // UniformSampleOneLight + EstimateDirect (TraceAlgorithms.cu:44-101)
                const f2 sl = rng.next2();
                float lpdf; const int li2 = sample_emitter(S, lpdf, sl.x);
                if (li2 >= 0) {
                    direct_rec dr; dr.ref = b.dg.P; dr.refN = b.dg.sys.n;
                    const f3 value = light_sample_direct(S, scene_lights(S)[li2], dr, rng.next2());
                    if (!is_zero(value)) {
                        bsdf_rec b2 = b; b2.wo = b.dg.sys.to_local(dr.d); b2.type_mask = kEAll & ~kEDelta;
                        const f3 bsdfVal = bsdf_f_top(mat, b2);
                        if (!is_zero(bsdfVal)) {
                            float st, su, sv; int stri, snode;
                            rays++;
                            if (!trace_single<true>(S, dr.ref, dr.d, S.eps, dr.dist - S.eps, st, su, sv, stri, snode)) {   // Occluded(r, 0, dist)
                                float weight = 1.0f;
                                if (dr.measure != kMeasureDiscrete) weight = power_heuristic((dr.measure == kMeasureArea ? dr.pdf * dr.dist / fabsf(dot(dr.n, dr.d)) : dr.pdf) * lpdf, bsdf_pdf_top(mat, b2));
                                cl = cl + cf * sdiv(value * bsdfVal * weight, lpdf);
                            }
                        }
                    }
                }
            }
            specularBounce = (b.sampled_type & kEDelta) != 0;
            cf = cf * f;
            r_o = b.dg.P; r_d = b.dg.sys.to_world(b.wo);
            if (is_zero(cf)) break;   // cannot contribute any more (the wavefront tracer makes the same cut)
            if (depth > P.rr_start_depth && !specularBounce) {
                const float q = max3c(cf);
                if (rng.next1() >= q) break;
                cf = sdiv(cf, q);
            }
        }
        if (!had_hit && S.env_map_index != 0xffffffffu) {   // PathTracer.cu:99-111
            const ctl_light& light = scene_lights(S)[S.env_map_index];
            float misWeight = 1.0f;
            if (!(!P.direct || depth == 1 || specularBounce)) misWeight = power_heuristic(brdf_scattering_pdf, env_pdf_direct(S, light, r_d) * pdf_emitter(S, S.env_map_index));
            cl = cl + misWeight * cf * env_eval(S, light, r_d);
        } else if (!had_hit) cl = cl + cf * f3(0.0f);   // EvalEnvironment == Spectrum(0) without a map, added all the same: a NaN / infinite throughput poisons the sample (oracle/ocore.h pathTrace)
        if (P.debug_out) { P.debug_out[0] = cl.x; P.debug_out[1] = cl.y; P.debug_out[2] = cl.z; }
        else add_sample(image, P.width, P.height, pX.x, pX.y, cl);
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(ray_count, rays);
}

// ---- PathTraceRegularization<DIRECT> (Integrators/PathTracer.cu:115-173): the plugin with Regularization = true
// Light::samplePosition of the emitters the mollified connection uses (SceneTypes/Light.cu:33-40, :304-311, :246-258); area and environment emitters are skipped
__device__ f3 light_sample_position(const ctl_light& L, f2 sample, f3& p) {
    if (L.type == CTL_LIGHT_POINT || L.type == CTL_LIGHT_SPOT) { p = f3(L.position[0], L.position[1], L.position[2]); return f3(L.radiance[0], L.radiance[1], L.radiance[2]) * (4 * kPi); }
    if (L.type == CTL_LIGHT_DISTANT) {
        const f2 q = square_to_disk_concentric(sample);
        const frame F = light_frame(L);
        const f3 perpOffset = F.to_world(f3(q.x, q.y, 0) * L.bsphere_radius), d = F.to_world(f3(0.0f, 0.0f, 1.0f));
        p = d * L.bsphere_radius + perpOffset;
        const float surfaceArea = kPi * L.bsphere_radius * L.bsphere_radius, invSurfaceArea = 1.0f / surfaceArea;
        return sdiv(f3(L.radiance[0], L.radiance[1], L.radiance[2]), invSurfaceArea);
    }
    p = f3(0.0f); return f3(0.0f);
}
// InfiniteLight::evalEnvironment(ray, rX, rY) (SceneTypes/Light.cu:496-518)
__device__ f3 env_eval_differential(const dev_scene& S, const ctl_light& L, f3 dir, f3 dirX, f3 dirY) {
    const f3 v = xform_dir_transpose(L.to_world, dir);
    const f2 uv{ m_atan2(v.x, -v.z) * kInvTwoPi, m_acos(fminf(1.0f, fmaxf(-1.0f, v.y))) * kInvPi };
    const f3 dvdx = xform_dir_transpose(L.to_world, dirX) - v, dvdy = xform_dir_transpose(L.to_world, dirY) - v;
    const float t1 = kInvTwoPi / (v.x * v.x + v.z * v.z), t2 = -kInvPi / fmaxf(sqrtf(fmaxf(0.0f, 1.0f - v.y * v.y)), 1e-4f);
    const f2 dudx{ t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y }, dudy{ t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y };
    return mip_eval(S.images[L.env_image], S.mip_levels[L.env_image], S.mip_weight_lut, uv, dudx, dudy) * f3(L.env_scale[0], L.env_scale[1], L.env_scale[2]);
}
// EstimateDirect with light_pdf = 1 (Kernel/TraceAlgorithms.cu:44-73) — one term of UniformSampleAllLights (:75-90)
__device__ f3 estimate_direct_all(const dev_scene& S, const ctl_material& mat, const bsdf_rec& b, sampler& rng, unsigned long long& rays) {
    f3 L(0.0f);
    for (uint32_t i = 0; i < S.num_lights; i++) {
        direct_rec dr; dr.ref = b.dg.P; dr.refN = b.dg.sys.n;
        const f3 value = light_sample_direct(S, scene_lights(S)[S.light_indices[i]], dr, rng.next2());
        if (is_zero(value)) continue;
        bsdf_rec b2 = b; b2.wo = b.dg.sys.to_local(dr.d); b2.type_mask = kEAll & ~kEDelta;
        const f3 bsdfVal = bsdf_f_top(mat, b2);
        if (is_zero(bsdfVal)) continue;
        float st, su, sv; int stri, snode;
        rays++;
        if (trace_single<true>(S, dr.ref, dr.d, S.eps, dr.dist - S.eps, st, su, sv, stri, snode)) continue;
        float weight = 1.0f;
        if (dr.measure != kMeasureDiscrete) weight = power_heuristic((dr.measure == kMeasureArea ? dr.pdf * dr.dist / fabsf(dot(dr.n, dr.d)) : dr.pdf) * 1.0f, bsdf_pdf_top(mat, b2));
        L = L + (value * bsdfVal * weight) / 1.0f;
    }
    return L;
}
// mollifier[pass]: radius2 of PathTracer::RenderBlock (PathTracer.cu:196-203) for the pass, computed on the host
__global__ __launch_bounds__(256) void k_path_trace_regularization(dev_scene S, pass_params P, const float* __restrict__ mollifier, ctl_pixel_data* __restrict__ image,
                                                                   unsigned long long* __restrict__ ray_count) {
    const uint32_t tiles_x = (P.width + 63) / 64;
    const uint32_t n_total = P.n_local_pixels * P.batch;
    const uint32_t n1 = CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH;
    unsigned long long rays = 0;
    for (uint32_t gi = blockIdx.x * 256u + threadIdx.x; gi < n_total; gi += gridDim.x * 256u) {
        const uint32_t pass_b = gi / P.n_local_pixels, li = gi - pass_b * P.n_local_pixels;
        const uint32_t tile = P.tile_rank + (li >> 12) * P.tile_world, p = li & 4095u, micro = p >> 6, ln = p & 63u;
        const uint32_t x = (tile % tiles_x) * 64 + (micro & 7u) * 8 + (ln & 7u), y = (tile / tiles_x) * 64 + (micro >> 3) * 8 + (ln >> 3);
        if (x >= P.width || y >= P.height) continue;
        sampler rng{ P.t1 + pass_b * n1, P.t2 + pass_b * n1, y * P.width + x, 0, 0 };
        const f2 j = rng.next2();
        const f2 pX{ (float)x + j.x, (float)y + j.y };
        const f2 ap = rng.next2();   // aperture sample
        f3 r_o, r_d, r_ox, r_dx, r_oy, r_dy; sensor_sample_ray_differential(S.cam, pX, ap, r_o, r_d, r_ox, r_dx, r_oy, r_dy);
        const float g_fRMollifier = mollifier[pass_b];
        f3 cl(0.0f), cf(1.0f);
        int depth = 0; bool specularBounce = false, had_hit = false;
        for (;;) {   // while (traceRay(r, &r2) && depth++ < maxPathLength)
            float t, u, v; int tri, node;
            had_hit = trace_single<false>(S, r_o, r_d, S.eps, 3.402823466e+38f, t, u, v, tri, node);
            rays++;
            if (!(had_hit && depth++ < P.max_path_length)) break;
            bsdf_rec b; b.eta = 1.0f; b.sampled_type = 0; b.type_mask = kEAll;
            b.dg.P = r_o + t * r_d;
            fill_dg(S, u, v, tri, node, b.dg);
            b.wi = b.dg.sys.to_local(-r_d);
            const uint4 ninfo = S.node_info[node];
            const ctl_material& mat = S.mats[ninfo.x + tri_mat_index(S, tri)];
            if (mat.map_kind != CTL_MAP_NONE) sample_normal_map(mat, b.dg);
            if (mat.two_sided && b.wi.z < 0) { b.dg.n = -b.dg.n; b.dg.sys.n = -b.dg.sys.n; b.wi.z *= -1.0f; }
            if (depth == 1) compute_partials(b.dg, r_ox, r_dx, r_oy, r_dy);
            const uint32_t nli = mat.node_light_index;
            if (nli != 0xffffffffu && (!P.direct || depth == 1 || specularBounce)) cl = cl + cf * light_eval(S, scene_lights(S)[nli == 0 ? ninfo.y : ninfo.z], b.dg.P, b.dg.sys.n, -r_d);
            float pdf_unused;
            const f3 f = bsdf_sample_top(mat, b, pdf_unused, rng.next2());
            if (P.direct) {
                if (mat.combined_type & kEDelta) {   // mollified connection of the sampled direction to a point / spot / distant emitter (PathTracer.cu:131-147)
                    f2 sample = rng.next2();
                    if (S.num_lights) {
                        float emPdf; const int li2 = sample_emitter(S, emPdf, sample.x);
                        uint32_t slot = 0; while (slot < S.num_lights && S.light_indices[slot] != (uint32_t)li2) slot++;   // sampleEmitter rescales the sample it consumed (KernelDynamicScene.cu:25-40)
                        const float fU = S.light_cdf[slot], fL = slot > 0 ? S.light_cdf[slot - 1] : 0.0f;
                        sample.x = (sample.x - fL) / (fU - fL);
                        const ctl_light& l = scene_lights(S)[li2];
                        f3 lp; const f3 l_s = sdiv(light_sample_position(l, sample, lp), emPdf);
                        const float lDist = length(lp - b.dg.P);
                        const f3 lDir = (lp - b.dg.P) / lDist;
                        if (!(l.type == CTL_LIGHT_DIFFUSE || l.type == CTL_LIGHT_INFINITE)) {
                            float st, su, sv; int stri, snode;
                            rays++;
                            if (!trace_single<true>(S, b.dg.P, lDir, S.eps, lDist - S.eps, st, su, sv, stri, snode)) {   // Occluded(r, 0, lDist)
                                const float eps = m_atan(g_fRMollifier / lDist);
                                const float normalization = 1.0f / (2 * kPi * (1 - m_cos(eps)));
                                const float l_dot_o = dot(lDir, b.dg.sys.to_world(b.wo));
                                const float indicator = m_acos(l_dot_o) <= eps ? 1.0f : 0.0f;
                                cl = cl + cf * f * l_s * (normalization * indicator);
                            }
                        }
                    }
                } else cl = cl + cf * estimate_direct_all(S, mat, b, rng, rays);
            }
            specularBounce = (b.sampled_type & kEDelta) != 0;
            cf = cf * f;
            if (depth > P.rr_start_depth) {
                const float q = max3c(cf);
                if (rng.next1() < q) cf = sdiv(cf, q);
                else break;
            }
            r_o = b.dg.P; r_d = b.dg.sys.to_world(b.wo);
            had_hit = false;
        }
        if (S.env_map_index != 0xffffffffu) {   // PathTracer.cu:168-171, as written: the last ray's environment radiance is added whether it escaped or not
            const ctl_light& env = scene_lights(S)[S.env_map_index];
            if (!had_hit && depth == 0) cl = cf * env_eval_differential(S, env, r_d, r_dx, r_dy);
            else cl = cl + cf * env_eval(S, env, r_d);
        } else if (!had_hit && depth == 0) cl = cf * f3(0.0f);
        else cl = cl + cf * f3(0.0f);
        add_sample(image, P.width, P.height, pX.x, pX.y, cl);
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(ray_count, rays);
}

PathTracer::PathTracer() {
    m_sParameters.addBool("Direct", true);                        // Integrators/PathTracer.h:10-19
    m_sParameters.addBool("Regularization", false);               // PathTraceRegularization (k_path_trace_regularization)
    m_sParameters.addInterval("MaxPathLength", 50, 1, INT_MAX);
    m_sParameters.addInterval("RRStartDepth", 5, 1, INT_MAX);
    int dev = 0; hipDeviceProp_t prop; CTL_HIP(hipGetDevice(&dev)); CTL_HIP(hipGetDeviceProperties(&prop, dev));
    grid_blocks = prop.multiProcessorCount * 8;
}
void PathTracer::InitializeScene(Scene* s) {
    if (!s->S.flat_nodes) throw unsupported_error("PathTracer (megakernel): the scene must be created with CTL_SCENE_FLATTEN");
    Tracer<true>::InitializeScene(s);
}
void PathTracer::Resize(unsigned int _w, unsigned int _h) {
    Tracer<true>::Resize(_w, _h);
    n_local_pixels = shard_pixel_count(_w, _h, shard_rank, shard_world);
    if (!count_.p) count_.alloc(1);
}

void PathTracer::DoRender(Image* I, const float* d_t1, const float* d_t2, unsigned int n_batch) {
    // the reference's megakernel renders a block once per mention by the sampler, i.e. the SAME sample twice; only the wavefront plugin honours block samplers here
    if (pass_block_counts_) throw unsupported_error("PathTracer (megakernel): block samplers other than Uniform are served by the WavefrontPathTracer plugin only");
    pass_params P{};
    P.t1 = d_t1; P.t2 = (const float2*)d_t2; P.batch = n_batch; P.width = w; P.height = h; P.tile_rank = shard_rank; P.tile_world = shard_world; P.n_local_pixels = n_local_pixels;
    P.direct = m_sParameters.getValue("Direct"); P.max_path_length = m_sParameters.getValue("MaxPathLength"); P.rr_start_depth = m_sParameters.getValue("RRStartDepth");
    CTL_HIP(hipMemsetAsync(count_.p, 0, sizeof(unsigned long long), stream));
    timer.begin(stream, 2);
    if (m_sParameters.getValue("Regularization")) {
        // PathTracer::RenderBlock (PathTracer.cu:196-203): the mollifier radius of a pass from the scene box and the passes done, that pass included
        const float* lo = m_pScene->box_min; const float* hi = m_pScene->box_max;
        const float initialRadius = ((hi[0] - lo[0]) + (hi[1] - lo[1]) + (hi[2] - lo[2])) / 100;
        const float ALPHA = 0.75f;
        std::vector<float> m(n_batch);
        for (unsigned int k = 0; k < n_batch; k++) m[k] = m_pow(m_pow(initialRadius, float(2)) / m_pow(float(m_uPassesDone - n_batch + k + 1), 0.5f * (1 - ALPHA)), 1.0f / 2.0f);
        if (mollifier_.n < n_batch) mollifier_.alloc(n_batch);
        CTL_HIP(hipMemcpyAsync(mollifier_.p, m.data(), n_batch * sizeof(float), hipMemcpyHostToDevice, stream));
        CTL_HIP(hipStreamSynchronize(stream));   // `m` is pageable and leaves scope
        hipLaunchKernelGGL(k_path_trace_regularization, dim3(grid_blocks), dim3(256), 0, stream, m_pScene->S, P, (const float*)mollifier_.p, I->device(), count_.p);
    } else
    hipLaunchKernelGGL(k_path_trace, dim3(grid_blocks), dim3(256), 0, stream, m_pScene->S, P, I->device(), count_.p);
    timer.end(stream);
    CTL_HIP(hipMemcpyAsync(&host_count_, count_.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    CTL_HIP(hipStreamSynchronize(stream));
    total_rays_ += host_count_;
}
void PathTracer::takeRayCounts(uint64_t& path_rays, uint64_t& shadow_rays_) { path_rays = total_rays_; shadow_rays_ = 0; total_rays_ = 0; }
// PathTracer::DebugInternal (Integrators/PathTracer.cu:172-180): PathTrace<true> for one pixel, from the pixel's own position, with the tables Debug() just generated
void PathTracer::DebugInternal(Image* I, unsigned int x, unsigned int y, const float* d_t1, const float* d_t2, float rgb[3]) {
    if (debug_.n < 3) debug_.alloc(3);
    pass_params P{};
    P.t1 = d_t1; P.t2 = (const float2*)d_t2; P.batch = 1; P.width = w; P.height = h; P.tile_rank = 0; P.tile_world = 1; P.n_local_pixels = 1;
    P.direct = 1; P.max_path_length = m_sParameters.getValue("MaxPathLength"); P.rr_start_depth = m_sParameters.getValue("RRStartDepth");
    P.debug_out = debug_.p; P.debug_x = x; P.debug_y = y;
    CTL_HIP(hipMemsetAsync(count_.p, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(k_path_trace, dim3(1), dim3(256), 0, stream, m_pScene->S, P, I->device(), count_.p);
    CTL_HIP(hipMemcpyAsync(rgb, debug_.p, 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
    CTL_HIP(hipStreamSynchronize(stream));
}

} // namespace ctl
