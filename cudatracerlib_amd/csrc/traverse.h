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
constexpr int kLdsStackFlat = 19;    // flat kernel: 19 rows + 1 spare row (absorbs unused push slots) = 20 KiB per workgroup -> 8 workgroups = 32 waves per CU
__device__ int g_tri_batch = 1;      // flat kernel: leaf entries are tested once this many lanes wait at a leaf.  Measured on MI355X
                                     // (gpurun_out/tune_tri.log): batching leaves LOSES (973 -> 835 Mrays/s from 1 to 40) — the kernel is
                                     // memory-latency bound and every waiting lane is a load not in flight; kept as a knob (CTL_TRI_BATCH)
__device__ int g_any_sorted = 0;     // flat kernel, any-hit: visit hit children nearest-first instead of in slot order (CTL_ANY_SORTED)
__device__ int g_refill_idle = 20;   // refill as soon as this many lanes of the wave are idle (CTL_REFILL_IDLE overrides; measured 4: 2386, 8: 2433,
                                     // 12: 2466, 20: 2483, 32: 2473 Mrays/s on synthetic-SM)
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
    uint32_t chunk_next = 0, chunk_end = 0; bool exhausted = (n == 0) || S.n_nodes == 0;
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
            if (chunk_next >= chunk_end) {
                const uint32_t claim = guided_chunk(n, chunk_end);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work, claim);
                base = __shfl(base, 0, 64);
                chunk_next = base; chunk_end = base + claim < n ? base + claim : n;
                if (base >= n) { exhausted = true; chunk_next = chunk_end = n; }
            }
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

// ------------------------------------------------------------------------------------------------------------------
// Single-level variant over the flattened world-space BVH (flatten.cpp).  One loop iteration = one 64-B fetch group per
// lane — an inner node OR one leaf entry — so every lane that holds a ray does useful work in every iteration; only the
// math after the (shared) fetch diverges between the two kinds.
template <bool ANY_HIT, bool COUNT, bool ALPHA = false>
__device__ __forceinline__ void intersect_flat(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                               float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_stack, trav_counts& cnt) {
    const int lane = threadIdx.x & 63;
    const int refill_idle = g_refill_idle, tri_batch = g_tri_batch;
    lane_stack_t<kLdsStackFlat> st; st.lds = lds_stack + threadIdx.x;
    bool has_ray = false;
    uint32_t ray_id = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, tmin = 0, idx = 0, idy = 0, idz = 0, oox = 0, ooy = 0, ooz = 0;
    float ht = 0, hu = 0, hv = 0; int htri = -1, hnode = -1;
    int sp = 0, node = kSentinel;
    const float4* __restrict__ nodes = S.flat_nodes;
    const float4* __restrict__ leaves = S.flat_leaves;
    uint32_t chunk_next = 0, chunk_end = 0; bool exhausted = (n == 0);

    for (;;) {
        const unsigned long long idle = __ballot(!has_ray);
        if (idle != 0ull && !exhausted && (__popcll(idle) >= refill_idle || idle == ~0ull)) {
            if (chunk_next >= chunk_end) {
                const uint32_t claim = guided_chunk(n, chunk_end);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work, claim);
                base = __shfl(base, 0, 64);
                chunk_next = base; chunk_end = base + claim < n ? base + claim : n;
                if (base >= n) { exhausted = true; chunk_next = chunk_end = n; }
            }
            if (!exhausted) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0));
                const uint32_t avail = chunk_end - chunk_next, want = (uint32_t)__popcll(idle);
                const uint32_t my = chunk_next + prefix;
                if (!has_ray && prefix < avail) {
                    const float4 o = ro[my], d = rd[my];
                    ray_id = my; has_ray = true;
                    ox = o.x; oy = o.y; oz = o.z; tmin = o.w; dx = d.x; dy = d.y; dz = d.z;
                    idx = rcp_guarded(dx); idy = rcp_guarded(dy); idz = rcp_guarded(dz);
                    oox = ox * idx; ooy = oy * idy; ooz = oz * idz;
                    ht = d.w; hu = hv = 0.0f; htri = -1; hnode = -1;
                    sp = 0; st.set(0, kSentinel); node = S.flat_root;
                }
                chunk_next += want < avail ? want : avail;
            }
        }
        if (__ballot(has_ray) == 0ull) { if (exhausted) break; continue; }

        // leaf entries are tested in batches: lanes that reached a leaf sit out until enough of them wait (or nobody has an
        // inner node left), so the Woop code runs with many lanes instead of a handful in every iteration
        const bool is_leaf = has_ray && node < 0;
        const unsigned long long at_leaf = __ballot(is_leaf);
        const bool do_leaf = __popcll(at_leaf) >= tri_batch || at_leaf == __ballot(has_ray);
        if (has_ray && (!is_leaf || do_leaf)) {
            const float4* __restrict__ p = is_leaf ? leaves + (size_t)(~node) * 4 : nodes + node;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            bool finished = false;
            if (!is_leaf) {
                if (COUNT) { cnt.n_inner++; if (lane == (int)__builtin_ctzll(__ballot(1))) cnt.w_inner++; }
                // 4-wide node (flatten.h): q0 = origin.xyz + {ex,ey,ez,mask}; q1 = qlo_x,qhi_x,qlo_y,qhi_y; q2 = qlo_z,qhi_z,child0,child1; q3 = child2,child3
                const uint32_t meta = __float_as_uint(q0.w);
                const float ax = __uint_as_float((meta & 0xffu) << 23) * idx, ay = __uint_as_float(((meta >> 8) & 0xffu) << 23) * idy, az = __uint_as_float(((meta >> 16) & 0xffu) << 23) * idz;
                const float bx = __builtin_fmaf(q0.x, idx, -oox), by = __builtin_fmaf(q0.y, idy, -ooy), bz = __builtin_fmaf(q0.z, idz, -ooz);
                const uint32_t lx = __float_as_uint(q1.x), hx = __float_as_uint(q1.y), ly = __float_as_uint(q1.z), hy = __float_as_uint(q1.w), lz = __float_as_uint(q2.x), hz = __float_as_uint(q2.y);
                // near / far plane words picked by the sign of the ray direction (ax.. carry the sign of 1/d): the entry distance is
                // the max of three near planes and the exit distance the min of three far planes, no per-plane min/max
                const bool px = idx >= 0.0f, py = idy >= 0.0f, pz = idz >= 0.0f;
                const uint32_t nx = px ? lx : hx, fx = px ? hx : lx, ny = py ? ly : hy, fy = py ? hy : ly, nz = pz ? lz : hz, fz = pz ? hz : lz;
                uint32_t key[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float tnx = __builtin_fmaf((float)((nx >> (8 * c)) & 0xffu), ax, bx), tfx = __builtin_fmaf((float)((fx >> (8 * c)) & 0xffu), ax, bx);
                    const float tny = __builtin_fmaf((float)((ny >> (8 * c)) & 0xffu), ay, by), tfy = __builtin_fmaf((float)((fy >> (8 * c)) & 0xffu), ay, by);
                    const float tnz = __builtin_fmaf((float)((nz >> (8 * c)) & 0xffu), az, bz), tfz = __builtin_fmaf((float)((fz >> (8 * c)) & 0xffu), az, bz);
                    const float cmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                    const float cmax = fminf(fminf(tfx, tfy), fminf(tfz, ht));
                    const bool h = (cmax >= cmin) && ((meta >> (24 + c)) & 1u);
                    // sort key: entry distance (>= 0, so its bit pattern orders like the float) with the child slot in the two low bits
                    key[c] = h ? ((__float_as_uint(cmin) & ~3u) | (uint32_t)c) : 0xffffffffu;
                }
                // front-to-back: 5-comparator network on the keys; misses (0xffffffff) end up last, so the hits are a prefix
#define CTL_CSWAP(a, b) { const uint32_t lo_ = key[a] < key[b] ? key[a] : key[b], hi_ = key[a] < key[b] ? key[b] : key[a]; key[a] = lo_; key[b] = hi_; }
                CTL_CSWAP(0, 1) CTL_CSWAP(2, 3) CTL_CSWAP(0, 2) CTL_CSWAP(1, 3) CTL_CSWAP(1, 2)
#undef CTL_CSWAP
                const int ch0 = __float_as_int(q2.z), ch1 = __float_as_int(q2.w), ch2 = __float_as_int(q3.x), ch3 = __float_as_int(q3.y);
                auto child_of = [&](uint32_t k) { const uint32_t s = k & 3u; return s == 0 ? ch0 : (s == 1 ? ch1 : (s == 2 ? ch2 : ch3)); };
                const int n_hit = (key[0] != 0xffffffffu) + (key[1] != 0xffffffffu) + (key[2] != 0xffffffffu) + (key[3] != 0xffffffffu);
                if (n_hit == 0) { node = st.get(sp); sp--; }
                else {
                    node = child_of(key[0]);   // nearest continues, the others go onto the stack farthest first
                    if (n_hit > 1) {
                        const int top = sp + n_hit - 1;
                        if (top < kLdsStackFlat) {   // common case: three unconditional LDS stores, unused ones into the spare row
                            st.lds[top * 256] = child_of(key[1]);
                            st.lds[(n_hit >= 3 ? top - 1 : kLdsStackFlat) * 256] = child_of(key[2]);
                            st.lds[(n_hit >= 4 ? top - 2 : kLdsStackFlat) * 256] = child_of(key[3]);
                            sp = top;
                        } else {
                            if (n_hit >= 4) { sp++; st.set(sp, child_of(key[3])); }
                            if (n_hit >= 3) { sp++; st.set(sp, child_of(key[2])); }
                            sp++; st.set(sp, child_of(key[1]));
                        }
                    }
                }
            } else {
                if (COUNT) { cnt.n_tri++; if (lane == (int)__builtin_ctzll(__ballot(1))) cnt.w_tri++; }
                const uint32_t index = __float_as_uint(q3.x);
                const float Oz = q0.w - ox * q0.x - oy * q0.y - oz * q0.z;
                const float invDz = __builtin_amdgcn_rcpf(dx * q0.x + dy * q0.y + dz * q0.z);   // 1 ulp; the flattened layout promises fp32 round-off, not bit equality
                const float t = Oz * invDz;
                if (t > tmin && t < ht) {
                    const float Ox = q1.w + ox * q1.x + oy * q1.y + oz * q1.z;
                    const float Dx = dx * q1.x + dy * q1.y + dz * q1.z;
                    const float u = Ox + t * Dx;
                    if (u >= 0.0f) {
                        const float Oy = q2.w + ox * q2.x + oy * q2.y + oz * q2.z;
                        const float Dy = dx * q2.x + dy * q2.y + dz * q2.z;
                        const float v = Oy + t * Dy;
                        if (v >= 0.0f && u + v <= 1.0f && (!ALPHA || alpha_survives(S.tri_data, S.node_info, S.mats, S.images, (int)(index >> 1), (int)__float_as_uint(q3.y), u, v))) {
                            ht = t; hu = u; hv = v; htri = (int)(index >> 1); hnode = (int)__float_as_uint(q3.y);
                            if (ANY_HIT) finished = true;
                        }
                    }
                }
                if (index & 1) { node = st.get(sp); sp--; } else node = node - 1;   // ~(entry + 1) == node - 1
            }
            if (!finished) finished = (node == kSentinel);
            if (finished) {
                if (ANY_HIT && occ) occ[ray_id] = htri >= 0 ? 1u : 0u;
                if (hit) { hit[ray_id] = make_float4(ht, hu, hv, __int_as_float(htri)); hit_node[ray_id] = hnode; }
                has_ray = false; node = kSentinel;
            }
        }
    }
}

} // namespace ctl
