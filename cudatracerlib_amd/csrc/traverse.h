// traverse.h — two-level BVH traversal + Woop triangle test, persistent wave64 formulation for gfx950.
//
// Behaviour = intersectKernel<ANY_HIT> (Kernel/TraceHelper.cu:326-734): scene BVH -> instance transform -> mesh BVH ->
// Woop triangles, closest (or first) hit with t in (tmin, tmax).  The arithmetic that decides the reported hit (ray
// transform, t, u, v) is evaluated in the reference's expression order without FMA contraction, so (t, u, v, triangle)
// are bit-identical to the reference's host path; the slab tests only cull.
//
// Execution model (re-derived for 64-wide waves, not the reference's 32-lane ballot code, TraceHelper.cu:386-399):
//  * a wave owns 64 ray slots and keeps running; a lane whose ray is finished writes its hit and, once enough lanes are
//    idle, the idle lanes are refilled together with fresh rays (ballot + mbcnt prefix into a per-wave chunk that is
//    claimed from the global cursor with one atomic per 512 rays) — incoherent rays finish at very different times,
//    and without refill most lanes of a wave idle behind its slowest ray;
//  * the traversal stack lives in LDS ([entry][thread] layout: a lane always hits its own bank), only entries beyond
//    kLdsStack spill to scratch.
#pragma once
#include "device_scene.h"
#include "mipmap.h"

namespace ctl {

struct trav_counts { uint32_t n_inner, n_tri, n_inst, w_inner, w_tri; };

constexpr int kLdsStack = 24;        // stack entries per lane kept in LDS, two-level kernel (24 x 256 x 4 B = 24 KiB per workgroup)
__device__ int g_refill_idle = 12;   // refill as soon as this many lanes of the wave are idle (CTL_REFILL_IDLE overrides; round 2: 4: 2386, 8: 2433,
                                     // 12: 2466, 20: 2483, 32: 2473 Mrays/s on synthetic-SM; with round 3's kernel, ms per fused launch: 8: 15.39, 12: 15.23, 16: 15.33, 20: 15.53, 28: 16.37)
constexpr uint32_t kChunk = 512;     // most rays a wave claims from the global cursor per atomic
__device__ int g_chunk_guided = 1;   // 1: claims shrink with the rays that are left (guided_chunk); 0: always kChunk (CTL_CHUNK_GUIDED=0)

// Rays a wave claims next.  A fixed claim of 512 costs a launch ~0.5 ms whatever its size: at the start every one of the 8192 resident waves
// claims 512 rays (a launch of 2 M rays keeps half of the waves idle while the others work through eight rounds of rays), and at the end
// the wave with the last claim runs on alone.  Guided self-scheduling instead: a claim is the share of the REMAINING rays (as this wave last
// saw the cursor) that would keep every wave supplied for two more claims, between 64 and kChunk, in whole waves of 64.
__device__ __forceinline__ uint32_t guided_chunk(uint32_t n, uint32_t last_base) {
    if (!g_chunk_guided) return kChunk;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6), left = n > last_base ? n - last_base : 0u;
    const uint32_t c = (left / (2u * waves)) & ~63u;
    return c < 64u ? 64u : (c > kChunk ? kChunk : c);
}

// A wave's claims from the launch's ray cursor.  The FIRST claim is static — wave w takes rays [w c0, (w + 1) c0), c0 = the guided claim for the whole queue — and only what
// lies beyond the waves' static shares goes through the cursor (which counts from 0 there).  Round 6: one address serves ~88 atomics per microsecond, a launch has 8192 (7168)
// waves, and every wave used to open with one claim and end with one that found the cursor past the end — two bursts of ~90 us per queue whatever its size, four per fused
// launch: the 0.33 ms a near-empty traversal launch took.  A queue of fewer rays than the static shares cover now needs no atomic at all.  (The grid holds exactly the
// workgroups that are resident together, kernels.hip traversal_blocks: a workgroup that started late would sit on its static share.)
#ifndef CTL_STATIC_FIRST_CLAIM
#define CTL_STATIC_FIRST_CLAIM 1
#endif
struct ray_claims {
    uint32_t next, end, static_total; bool exhausted;
    __device__ __forceinline__ void init(uint32_t n) {
#if CTL_STATIC_FIRST_CLAIM
        const uint32_t waves = gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), c0 = guided_chunk(n, 0u);
        const uint64_t all = (uint64_t)waves * c0, mine = (uint64_t)wave * c0;
        static_total = all < n ? (uint32_t)all : n;
        next = mine < n ? (uint32_t)mine : n; end = mine + c0 < n ? (uint32_t)(mine + c0) : n;
        exhausted = next >= end && static_total >= n;   // no static share (the queue is shorter than that) and then nothing behind the shares either
#else
        static_total = 0u; next = end = 0u; exhausted = n == 0u;
#endif
    }
    // the next range once [next, end) is used up; false (and exhausted) when there is none.  Wave-uniform; lane 0 does the atomic.
    __device__ __forceinline__ bool refill(uint32_t n, uint32_t* __restrict__ work, int lane) {
        if (static_total >= n) { exhausted = true; next = end = n; return false; }
        const uint32_t claim = guided_chunk(n, end);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(work, claim);
        base = static_total + (uint32_t)__shfl((int)base, 0, 64);
        if (base >= n || base < static_total) { exhausted = true; next = end = n; return false; }   // (< static_total: the cursor ran past 2^32 - static_total, i.e. far past the end)
        next = base; end = base + claim < n && base + claim > base ? base + claim : n;
        return true;
    }
};

__device__ __forceinline__ float rcp_guarded(float d) {   // TraceHelper.cu:417-420: 1/(|d| > 2^-80 ? d : copysign(2^-80, d))
    const float ooeps = 8.271806125530277e-25f;   // exp2(-80)
    return 1.0f / (fabsf(d) > ooeps ? d : copysign_bits(ooeps, d));
}

// slab test of both children of one node; the reference's spanBegin/spanEnd (Math/MathFunc.h:443-444) reduce to
// float min/max because tmin >= 0.  fma here is culling-only.
__device__ __forceinline__ void slab2(const float4 n0, const float4 n1, const float4 nz, float idx, float idy, float idz, float oox, float ooy, float ooz,
                                      float tmin, float tmax, float& c0min, float& c0max, float& c1min, float& c1max) {
    const float c0lox = __builtin_fmaf(n0.x, idx, -oox), c0hix = __builtin_fmaf(n0.y, idx, -oox);
    const float c0loy = __builtin_fmaf(n0.z, idy, -ooy), c0hiy = __builtin_fmaf(n0.w, idy, -ooy);
    const float c0loz = __builtin_fmaf(nz.x, idz, -ooz), c0hiz = __builtin_fmaf(nz.y, idz, -ooz);
    const float c1lox = __builtin_fmaf(n1.x, idx, -oox), c1hix = __builtin_fmaf(n1.y, idx, -oox);
    const float c1loy = __builtin_fmaf(n1.z, idy, -ooy), c1hiy = __builtin_fmaf(n1.w, idy, -ooy);
    const float c1loz = __builtin_fmaf(nz.z, idz, -ooz), c1hiz = __builtin_fmaf(nz.w, idz, -ooz);
    c0min = fmaxf(fmaxf(fminf(c0lox, c0hix), fminf(c0loy, c0hiy)), fmaxf(fminf(c0loz, c0hiz), tmin));
    c0max = fminf(fminf(fmaxf(c0lox, c0hix), fmaxf(c0loy, c0hiy)), fminf(fmaxf(c0loz, c0hiz), tmax));
    c1min = fmaxf(fmaxf(fminf(c1lox, c1hix), fminf(c1loy, c1hiy)), fmaxf(fminf(c1loz, c1hiz), tmin));
    c1max = fminf(fminf(fmaxf(c1lox, c1hix), fmaxf(c1loy, c1hiy)), fminf(fmaxf(c1loz, c1hiz), tmax));
}

// per-lane traversal stack: entries [0, kLdsStack) in LDS, the rest in scratch
template <int N> struct lane_stack_t {
    int* lds;                                   // this lane's column, stride = blockDim.x
    int ovf[kStackSize - N];
    __device__ __forceinline__ int get(int i) const { return i < N ? lds[i * 256] : ovf[i - N]; }
    __device__ __forceinline__ void set(int i, int v) { if (i < N) lds[i * 256] = v; else ovf[i - N] = v; }
};
typedef lane_stack_t<kLdsStack> lane_stack;

// The whole intersect kernel body: `n` rays (ro, rd) -> hit / hit_node (closest) and/or occ (any-hit flag).
// ALPHA: candidate hits pass Material::AlphaTest first (the reference does this in its single-ray traceRay only, TraceHelper.cu:135-153)
template <bool ANY_HIT, bool COUNT, bool ALPHA = false>
__device__ __forceinline__ void intersect_persistent(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                                     float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_stack, trav_counts& cnt) {
    const int lane = threadIdx.x & 63;
    const int refill_idle = g_refill_idle;
    lane_stack st; st.lds = lds_stack + threadIdx.x;
    // ---- per-lane ray state
    bool has_ray = false;
    uint32_t ray_id = 0;
    float orgx = 0, orgy = 0, orgz = 0, dirx = 0, diry = 0, dirz = 0, tmin = 0;     // world-space ray
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;                            // current-space ray
    float idx = 0, idy = 0, idz = 0, oox = 0, ooy = 0, ooz = 0;
    float widx = 0, widy = 0, widz = 0, woox = 0, wooy = 0, wooz = 0;                // world-space slab terms, restored at instance exit
    float ht = 0, hu = 0, hv = 0; int htri = -1, hnode = -1;
    const float4* __restrict__ nodes = S.top_nodes;
    int sp = 0, node = kSentinel, sp_enter = 0, cur_inst = -1; uint32_t leaf_base = 0, tri_base = 0; bool bottom = false;
    // ---- per-wave ray chunk (uniform)
    ray_claims rc; rc.init(n); if (S.n_nodes == 0) rc.exhausted = true;
    uint32_t& chunk_next = rc.next; uint32_t& chunk_end = rc.end; bool& exhausted = rc.exhausted;
    if (S.n_nodes == 0) {   // empty scene: every ray misses (TraceHelper.cu:442-443)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            if (ANY_HIT && occ) occ[i] = 0u;
            if (hit) { hit[i] = make_float4(rd[i].w, 0.f, 0.f, __int_as_float(-1)); hit_node[i] = -1; }
        }
        return;
    }

    for (;;) {
        // ---- refill idle lanes
        const unsigned long long idle = __ballot(!has_ray);
        if (idle != 0ull && !exhausted && (__popcll(idle) >= refill_idle || idle == ~0ull)) {
            if (chunk_next >= chunk_end) rc.refill(n, work, lane);
            if (!exhausted) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0));
                const uint32_t avail = chunk_end - chunk_next, want = (uint32_t)__popcll(idle);
                const uint32_t my = chunk_next + prefix;
                if (!has_ray && prefix < avail) {
                    const float4 o = ro[my], d = rd[my];
                    ray_id = my; has_ray = true;
                    orgx = o.x; orgy = o.y; orgz = o.z; tmin = o.w; dirx = d.x; diry = d.y; dirz = d.z;
                    ox = orgx; oy = orgy; oz = orgz; dx = dirx; dy = diry; dz = dirz;
                    idx = rcp_guarded(dx); idy = rcp_guarded(dy); idz = rcp_guarded(dz);
                    oox = ox * idx; ooy = oy * idy; ooz = oz * idz;
                    widx = idx; widy = idy; widz = idz; woox = oox; wooy = ooy; wooz = ooz;
                    ht = d.w; hu = hv = 0.0f; htri = -1; hnode = -1;
                    nodes = S.top_nodes; bottom = false; sp = 0; st.set(0, kSentinel); node = S.start_node;
                }
                chunk_next += want < avail ? want : avail;
            }
        }
        if (__ballot(has_ray) == 0ull) { if (exhausted) break; continue; }

        if (has_ray) {
            // ---- inner nodes: the wave stays here while any lane still has one
            while ((unsigned)node < (unsigned)kSentinel) {
                const float4 n0 = nodes[node], n1 = nodes[node + 1], nz = nodes[node + 2], cn = nodes[node + 3];
                if (COUNT) { cnt.n_inner++; if (lane == (int)__builtin_ctzll(__ballot(1))) cnt.w_inner++; }
                float c0min, c0max, c1min, c1max;
                slab2(n0, n1, nz, idx, idy, idz, oox, ooy, ooz, tmin, ht, c0min, c0max, c1min, c1max);
                int c0 = __float_as_int(cn.x), c1 = __float_as_int(cn.y);
                const bool t0 = (c0max >= c0min), t1 = (c1max >= c1min);
                if (!t0 && !t1) { node = st.get(sp); sp--; }
                else {
                    node = t0 ? c0 : c1;
                    if (t0 && t1) { if (c1min < c0min) { const int t = node; node = c1; c1 = t; } sp++; st.set(sp, c1); }
                }
            }
            bool finished = false;
            if (node < 0) {
                if (!bottom) {
                    // ---- enter instance ~node (TraceHelper.cu:526-560)
                    cur_inst = ~node;
                    if (COUNT) cnt.n_inst++;
                    const float4 r0 = S.inst[cur_inst * 4], r1 = S.inst[cur_inst * 4 + 1], r2 = S.inst[cur_inst * 4 + 2], r3 = S.inst[cur_inst * 4 + 3];
                    m34 m; m.r[0][0] = r0.x; m.r[0][1] = r0.y; m.r[0][2] = r0.z; m.r[0][3] = r0.w; m.r[1][0] = r1.x; m.r[1][1] = r1.y; m.r[1][2] = r1.z; m.r[1][3] = r1.w;
                    m.r[2][0] = r2.x; m.r[2][1] = r2.y; m.r[2][2] = r2.z; m.r[2][3] = r2.w;
                    const f3 d = xform_dir(m, f3(dirx, diry, dirz)), o = xform_point_w(m, f3(orgx, orgy, orgz), r3.x);
                    ox = o.x; oy = o.y; oz = o.z; dx = d.x; dy = d.y; dz = d.z;
                    idx = rcp_guarded(dx); idy = rcp_guarded(dy); idz = rcp_guarded(dz);
                    oox = ox * idx; ooy = oy * idy; ooz = oz * idz;
                    nodes = S.bot_nodes + __float_as_uint(r3.y);
                    leaf_base = __float_as_uint(r3.z); tri_base = __float_as_uint(r3.w);
                    sp++; st.set(sp, kExitMarker); sp_enter = sp;
                    bottom = true; node = 0;
                } else {
                    // ---- leaf: Woop triangles (TraceHelper.cu:636-694)
                    for (uint32_t addr = leaf_base + (uint32_t)(~node);; addr++) {
                        const float4 v00 = S.leaf_tris[addr * 4], v11 = S.leaf_tris[addr * 4 + 1], v22 = S.leaf_tris[addr * 4 + 2];
                        const uint32_t index = __float_as_uint(S.leaf_tris[addr * 4 + 3].x);
                        if (COUNT) { cnt.n_tri++; if (lane == (int)__builtin_ctzll(__ballot(1))) cnt.w_tri++; }
                        const float Oz = v00.w - ox * v00.x - oy * v00.y - oz * v00.z;
                        const float invDz = 1.0f / (dx * v00.x + dy * v00.y + dz * v00.z);
                        const float t = Oz * invDz;
                        if (t > tmin && t < ht) {
                            const float Ox = v11.w + ox * v11.x + oy * v11.y + oz * v11.z;
                            const float Dx = dx * v11.x + dy * v11.y + dz * v11.z;
                            const float u = Ox + t * Dx;
                            if (u >= 0.0f) {
                                const float Oy = v22.w + ox * v22.x + oy * v22.y + oz * v22.z;
                                const float Dy = dx * v22.x + dy * v22.y + dz * v22.z;
                                const float v = Oy + t * Dy;
                                if (v >= 0.0f && u + v <= 1.0f && (!ALPHA || alpha_survives(S.tri_data, S.node_info, S.mats, S.images, (int)((index >> 1) + tri_base), cur_inst, u, v))) {
                                    ht = t; hu = u; hv = v; htri = (int)((index >> 1) + tri_base); hnode = cur_inst;
                                    if (ANY_HIT) { finished = true; break; }
                                }
                            }
                        }
                        if (index & 1) break;
                    }
                    node = st.get(sp); sp--;
                }
            }
            if (!finished) {
                // a mesh BVH that ends in the sentinel (one-leaf meshes carry 0x76543210 as second child) only ends that mesh
                if (bottom && node == kSentinel) { sp = sp_enter - 1; node = kExitMarker; }
                if (node == kExitMarker) {
                    ox = orgx; oy = orgy; oz = orgz; dx = dirx; dy = diry; dz = dirz;
                    idx = widx; idy = widy; idz = widz; oox = woox; ooy = wooy; ooz = wooz;
                    nodes = S.top_nodes; bottom = false;
                    node = st.get(sp); sp--;
                }
                finished = (node == kSentinel);
            }
            if (finished) {
                if (ANY_HIT && occ) occ[ray_id] = htri >= 0 ? 1u : 0u;
                if (hit) { hit[ray_id] = make_float4(ht, hu, hv, __int_as_float(htri)); hit_node[ray_id] = hnode; }
                has_ray = false; node = kSentinel;
            }
        }
    }
}

} // namespace ctl
