// traverse_flat.h — traversal of the flattened world-space BVH (flatten.h, 4-wide "Q4" nodes) by persistent wave64 waves, one ray per lane.
//
// What is reported is the reference's: every leaf entry is evaluated with the two-level arithmetic of intersectKernel
// (Kernel/TraceHelper.cu:526-560 ray into the instance's object space, :646-682 Woop test with an exact division), in the
// reference's expression order, so (t, u, v, triangle, node) are bit-identical to the two-level traversal (traverse.h) whenever
// both look at the winning triangle; the world-space tree only culls.  Rays that hit two triangles at exactly the same t may
// report either (the visiting order differs), as between any two BVHs.
//
// Execution model (measured on MI355X, profiles/r02* .. r04*).  The kernel is bound by VALU issue (~90 % of the SIMDs' issue cycles by the counters of round 4) with
// its memory waits (41 % of wave time, seven waves deep) just hidden behind it: 8 % fewer VALU instructions bought 1.7 %, 11 % more cost 2.6 %, 22 % more cost 16 %, and taking
// whole node visits away pays one for one.  So the design spends arithmetic where it removes fetches or visits, and nowhere else:
//  * lane refill as in traverse.h (a wave claims rays from a device cursor, idle lanes are refilled together);
//  * node steps and leaf steps are separate wave-wide phases.  A lane that reaches a leaf POSTPONES it (one pending leaf per
//    lane) and keeps descending the tree speculatively; the wave runs the leaf code only once enough lanes hold a pending leaf
//    (or nobody has an inner node left); a leaf step tests one entry per lane.  In a unified loop the Woop code ran with ~8 of 64 lanes;
//  * the node step is branch-free: children sorted by entry distance with a compare + select network, pushes are unconditional LDS
//    stores (unused ones land in a spare row), the pop is an LDS read issued before the slab arithmetic.
// Everything that was measured against this and lost — 128-B fp32 and 2-wide node formats, the quad-cooperative node fetch, the top of the tree in LDS, prefetch touches,
// stack entries with their entry distance, 8 waves per SIMD, and round 4's two structural variants (the 8-wide node format of traverse_flat8.h: 10 % slower; entry tests
// from a wave-wide LDS queue, experiments/traverse_flat_wq.h: 38-47 % slower) — lives in csrc/experiments/ and builds only with -DCTL_FLAT_EXPERIMENTS / -DCTL_LEAF_QUEUE.
#pragma once
#include "traverse.h"
#include "flat_slab.h"

namespace ctl {

enum { kFmtQ4 = 0, kFmtF4 = 1, kFmtF2 = 2, kFmtQ8 = 3 };   // = flat_format (flatten.h); Q8 has its own kernel body (traverse_flat8.h), F4 / F2 are experiment builds

#ifndef CTL_FLAT_LDS_ROWS
#define CTL_FLAT_LDS_ROWS 23   // round 6: 23 rows + 1 spare = 24 KiB per 256-lane workgroup, SIX workgroups per CU (rounds 3-5: 19 rows, 20 KiB, seven).  Held to six by LDS padding the round-5 kernel
                               // ran 0.75 % faster, to five 0.8 % slower, to four 10 % (profiles/r06_lanes.log item 1); with the rows put to use: synthetic-SM 3123 / 3130 -> 3181 / 3170 Mrays/s,
                               // one rank of eight 19.86 -> 19.06 ms per 20 passes; 80 VGPRs (6 waves by registers too) the same (profiles/r06_traversal.log)
#endif
constexpr int kFlatLdsRows = CTL_FLAT_LDS_ROWS;          // stack entries per lane in LDS (+ 1 spare row); deeper entries live in scratch (19 rows: 0.015 % of the bench rays, profiles/r03a_stack_histogram.log)
constexpr int kFlatStackInts = 1;
constexpr int kTopCache = 0, kTopCacheFloats = 12;   // (experiment builds keep the top of the tree in LDS)
__device__ unsigned long long g_stack_hist[kStackSize];   // counting kernels only: rays by the deepest traversal-stack entry they used (ctl_traversal_stack_histogram)
__device__ int g_leaf_batch = 20;         // run the leaf phase once this many lanes hold a pending leaf entry (knob CTL_LEAF_BATCH; round 5, on the re-optimised tree: synthetic-SM 12: 3118, 16: 3137, 20: 3143, 24: 3123, 32: 3072 Mrays/s; synthetic-sm-hard 4232 / 4275 / 4382 / 4384 / 4311 — one value serves both, no per-scene choice needed; earlier trees: 8: 2253, 12: 2299, 16: 2315, 20: 2311, 24: 2288, 32: 2216 Mrays/s (profiles/r03_threshold_ab.log)

__device__ int g_leaf_batch_any = 20;     // the same threshold for the any-hit traversal (knob CTL_LEAF_BATCH_ANY; round 6 sweep: profiles/r06_traversal.log)
typedef __attribute__((address_space(3))) int flat_stack_lds_word;   // explicitly LDS: the pushes must compile to ds_write, not to generic flat stores
struct flat_stack {
    flat_stack_lds_word* lds;             // this lane's column, stride 256
    int ovf[kStackSize - kFlatLdsRows];
    __device__ __forceinline__ int get(int i) const {
        int w = lds[(i < kFlatLdsRows ? i : kFlatLdsRows) * 256];   // a ds_read whatever the depth (the spare row when the entry lives in scratch) ...
        if (i >= kFlatLdsRows) w = ovf[i - kFlatLdsRows];            // ... and the rare deep entry from scratch
        return w;
    }
    __device__ __forceinline__ void put_row(int row, int link) { lds[row * 256] = link; }   // the spare row kFlatLdsRows absorbs unused push slots
    __device__ __forceinline__ void set(int i, int link) { if (i < kFlatLdsRows) put_row(i, link); else ovf[i - kFlatLdsRows] = link; }
};

__device__ __forceinline__ float rcp_cull(float d) {   // slab tests only cull: the hardware reciprocal (1 ulp) of the guarded direction
    const float ooeps = 8.271806125530277e-25f;   // exp2(-80), TraceHelper.cu:417-420
    return __builtin_amdgcn_rcpf(fabsf(d) > ooeps ? d : copysign_bits(ooeps, d));
}

// One Woop triangle against the object-space ray (TraceHelper.cu:646-682), the reference's expression order, exact division.
// Returns true when the entry is the new closest hit (handed to the sink).
// Where an accepted hit goes.  hit_in_regs: the five values the caller keeps (single-ray traversal of the megakernel).  hit_in_memory: the wavefront kernels write
// the record to the ray's slot of the hit arrays at once — a ray accepts one to three hits on its way — and keep only the distance and a "found" bit: four registers
// fewer per lane for the whole traversal.
struct hit_in_regs {
    float& ht; float& hu; float& hv; int& htri; int& hnode;
    __device__ __forceinline__ float dist() const { return ht; }
    __device__ __forceinline__ void accept(float t, float u, float v, int tri, int nd, uint32_t) { ht = t; hu = u; hv = v; htri = tri; hnode = nd; }
};
struct hit_in_memory {
    float& ht; uint32_t& ray_word; float4* __restrict__ hit; int* __restrict__ hit_node; unsigned char* __restrict__ key_out;   // ray_word: ray index | found << 31
    __device__ __forceinline__ float dist() const { return ht; }
    __device__ __forceinline__ void accept(float t, float u, float v, int tri, int nd, uint32_t key) {
        ht = t;
        if (hit) { const uint32_t id = ray_word & 0x7fffffffu; hit[id] = make_float4(t, u, v, __int_as_float(tri)); hit_node[id] = nd; if (key_out) key_out[id] = (unsigned char)key; }
        ray_word |= 0x80000000u;
    }
};
// A geometric hit whose alpha test is still to be done (wavefront kernel with ALPHA: the evaluation — texture address, texel decode, a few hundred instructions — is run for all
// lanes that hold one at the same time, intersect_flat's alpha phase, instead of for one or two lanes at a time inside the entry phase).
struct alpha_cand { float t, u, v; int tri, node; uint32_t key; };
template <bool ALPHA, class SINK>
__device__ __forceinline__ bool flat_woop_test(const dev_scene& S, const float4 v00, const float4 v11, const float4 v22, uint32_t index, int nd, const f3 o, const f3 d, float tmin, SINK& sink,
                                               alpha_cand* cand = nullptr, bool* deferred = nullptr) {
    const float Oz = v00.w - o.x * v00.x - o.y * v00.y - o.z * v00.z;
    const float invDz = 1.0f / (d.x * v00.x + d.y * v00.y + d.z * v00.z);
    const float t = Oz * invDz;
    if (t > tmin && t < sink.dist()) {
        const float Ox = v11.w + o.x * v11.x + o.y * v11.y + o.z * v11.z;
        const float Dx = d.x * v11.x + d.y * v11.y + d.z * v11.z;
        const float u = Ox + t * Dx;
        if (u >= 0.0f) {
            const float Oy = v22.w + o.x * v22.x + o.y * v22.y + o.z * v22.z;
            const float Dy = d.x * v22.x + d.y * v22.y + d.z * v22.z;
            const float v = Oy + t * Dy;
            const int tri = (int)((index & 0x0fffffffu) >> 1);   // bits 28..31: the BSDF model of the entry's material on the device copy (dev_scene::flat_leaf_keys), else 0
            // nd: bit 31 on the device copy = the entry's material has an alpha map (tracer.hip; scenes with alpha maps only): only those entries are alpha-tested
            const int node = nd & 0x7fffffff;
            if (v >= 0.0f && u + v <= 1.0f) {
                if (ALPHA && nd < 0) {
                    if (cand) { cand->t = t; cand->u = u; cand->v = v; cand->tri = tri; cand->node = node; cand->key = index >> 28; *deferred = true; return false; }
                    if (!alpha_survives(S.tri_data, S.node_info, S.mats, S.images, tri, node, u, v)) return false;
                }
                sink.accept(t, u, v, tri, node, index >> 28);
                return true;
            }
        }
    }
    return false;
}
struct leaf_words { float4 v00, v11, v22, r0, r1, r2; uint2 iw; float w33; };
__device__ __forceinline__ void flat_leaf_load(const dev_scene& S, uint32_t e, leaf_words& L) {
    const float4* __restrict__ p = S.flat_leaves + (size_t)e * 8;
    L.v00 = p[0]; L.v11 = p[1]; L.v22 = p[2];
    L.iw = *(const uint2*)(p + 3);   // {globalTri << 1 | last, node}
    L.r0 = p[4]; L.r1 = p[5]; L.r2 = p[6];
    L.w33 = 1.0f;
    if (!S.inst_w_one) L.w33 = p[7].x;   // float4x4.h:402-406 divides by w; x / 1.0f == x, so scenes whose w are all 1 skip it
}
template <bool ANY_HIT, bool ALPHA, class SINK>
__device__ __forceinline__ int flat_leaf_eval(const dev_scene& S, uint32_t e, const leaf_words& L, float orgx, float orgy, float orgz, float dirx, float diry, float dirz, float tmin,
                                              SINK& sink, bool& got, alpha_cand* cand = nullptr, bool* deferred = nullptr) {
    m34 m; m.r[0][0] = L.r0.x; m.r[0][1] = L.r0.y; m.r[0][2] = L.r0.z; m.r[0][3] = L.r0.w; m.r[1][0] = L.r1.x; m.r[1][1] = L.r1.y; m.r[1][2] = L.r1.z; m.r[1][3] = L.r1.w;
    m.r[2][0] = L.r2.x; m.r[2][1] = L.r2.y; m.r[2][2] = L.r2.z; m.r[2][3] = L.r2.w;
    const f3 d = xform_dir(m, f3(dirx, diry, dirz));
    f3 o = xform_point(m, f3(orgx, orgy, orgz));
    if (!S.inst_w_one) o = f3(o.x / L.w33, o.y / L.w33, o.z / L.w33);
    if (flat_woop_test<ALPHA>(S, L.v00, L.v11, L.v22, L.iw.x, (int)L.iw.y, o, d, tmin, sink, cand, deferred)) { got = true; if (ANY_HIT) return -1; }
    return (L.iw.x & 1u) ? -1 : (int)(e + 1);
}
// One leaf entry (flat_leaf, 128 B) against the world-space ray: the ray through the node's inverse transform (TraceHelper.cu:526-560; the rows
// travel with the entry), then the Woop test.  Returns the next entry of the leaf, or -1 when this was its last one.
template <bool ANY_HIT, bool ALPHA, class SINK>
__device__ __forceinline__ int flat_leaf_test(const dev_scene& S, uint32_t e, float orgx, float orgy, float orgz, float dirx, float diry, float dirz, float tmin, SINK& sink, bool& got,
                                              alpha_cand* cand = nullptr, bool* deferred = nullptr) {
    leaf_words L; flat_leaf_load(S, e, L);
    return flat_leaf_eval<ANY_HIT, ALPHA>(S, e, L, orgx, orgy, orgz, dirx, diry, dirz, tmin, sink, got, cand, deferred);
}

// culling-only min / max: the hardware instructions as they are (a NaN operand loses, as with fmaxf / fminf)
__device__ __forceinline__ float max3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float min3_raw(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// ---- node steps.  Each returns the children the ray enters, nearest first, in c[0..n_hit): links as stored in the node.
struct ray_cull {   // the signs of the direction are read off idx / idy / idz where they are needed: a compare either way, and three registers fewer than keeping them
    float idx, idy, idz, oox, ooy, ooz;
    __device__ __forceinline__ int sx() const { return idx < 0.0f ? 1 : 0; }
    __device__ __forceinline__ int sy() const { return idy < 0.0f ? 1 : 0; }
    __device__ __forceinline__ int sz() const { return idz < 0.0f ? 1 : 0; }
};

#define CTL_CSWAP_PAIR(i, j) { const bool s_ = dd[j] < dd[i]; const float td_ = s_ ? dd[j] : dd[i]; dd[j] = s_ ? dd[i] : dd[j]; dd[i] = td_; \
                               const int tc_ = s_ ? c[j] : c[i]; c[j] = s_ ? c[i] : c[j]; c[i] = tc_; }

// Q4: 64-B node with 8-bit child boxes (flat4_node).  compact: the child links are implied by the layout and only the first 48 B are
// loaded — three per-lane L1 accesses instead of four — unless the link that led here says the node carries an oriented slab (bit 0; flat_slab.h):
// then the last 16 B are loaded too — all four loads issued together: the slab quarter fetched after the link arithmetic, i.e. after a wait for the first three, cost 9 % of
// the whole job — and every child's entry / exit distance is clipped by its interval along the node's slab direction.
struct node_words { float4 q0, q1, q2, q3; };
__device__ __forceinline__ void node_fetch_own(const float4* __restrict__ nodes, int node, bool compact, node_words& W) {
    const float4* __restrict__ p = nodes + (node & ~3);
    W.q0 = p[0]; W.q1 = p[1]; W.q2 = p[2];
    if (!compact || (node & 1)) W.q3 = p[3];
}

template <bool ORDERED = true>
__device__ __forceinline__ int node_step_q4(const node_words& W, int node, const ray_cull& R, float ox, float oy, float oz, float dx, float dy, float dz,
                                            float tmin, float ht, int c[4], float dd[4], bool compact) {
    const float4 q0 = W.q0, q1 = W.q1, q2 = W.q2;
    const uint32_t meta = __float_as_uint(q0.w);
    const float inf = __builtin_huge_valf();
    float s_alpha = 0.0f, s_bn = -inf, s_bf = inf; uint32_t s_nw = 0u, s_fw = 0u;   // no slab: [-inf, inf] for every child
    if (compact) {
        // implied links (flatten.h): link = base + nibble, no prefix sums and no branches — 20 VALU (the first spelling of the layout, counts per slot, took 45 and a dozen exec-mask branches)
        const uint32_t w0 = __float_as_uint(q2.z), w1 = __float_as_uint(q2.w);
        const uint32_t ib4 = w0 & 0x03fffffcu, nlb15 = __builtin_amdgcn_alignbit(0xffffffffu, w1, 6);   // first inner child * 4; ~(first entry + 15)
        const uint32_t dl = nlb15 - ib4;
        const uint32_t t1 = __builtin_amdgcn_ubfe(w0, 26, 4), t2 = __builtin_amdgcn_ubfe(w1, 2, 4), t3 = __builtin_amdgcn_alignbit(w1, w0, 30) & 15u;
        const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)meta, 28, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)meta, 29, 1), m2 = (uint32_t)__builtin_amdgcn_sbfe((int)meta, 30, 1), m3 = (uint32_t)((int)meta >> 31);
        c[0] = (int)((m0 & (nlb15 + 15u)) | (~m0 & (w0 & 0x03ffffffu)));
        c[1] = (int)(ib4 + t1 + (m1 & dl));
        c[2] = (int)(ib4 + t2 + (m2 & dl));
        c[3] = (int)(ib4 + t3 + (m3 & dl));
        if (node & 1) {
            const float4 q3 = W.q3;
            slab_ray SR;
            slab_setup(__float_as_uint(q3.x), q3.y, __float_as_uint(q3.z), __float_as_uint(q3.w), q0.x, q0.y, q0.z, ox, oy, oz, dx, dy, dz, SR);
            s_alpha = SR.alpha; s_bn = SR.beta_n; s_bf = SR.beta_f; s_nw = SR.near_w; s_fw = SR.far_w;
        }
    } else {
        const float4 q3 = W.q3;
        c[0] = __float_as_int(q3.x); c[1] = __float_as_int(q3.y); c[2] = __float_as_int(q3.z); c[3] = __float_as_int(q3.w);
    }
    const float ax = __uint_as_float((meta & 0xffu) << 23) * R.idx, ay = __uint_as_float(((meta >> 8) & 0xffu) << 23) * R.idy, az = __uint_as_float(((meta >> 16) & 0xffu) << 23) * R.idz;
    const float bx = __builtin_fmaf(q0.x, R.idx, -R.oox), by = __builtin_fmaf(q0.y, R.idy, -R.ooy), bz = __builtin_fmaf(q0.z, R.idz, -R.ooz);
    const uint32_t lx = __float_as_uint(q1.x), hx = __float_as_uint(q1.y), ly = __float_as_uint(q1.z), hy = __float_as_uint(q1.w), lz = __float_as_uint(q2.x), hz = __float_as_uint(q2.y);
    const bool negx = R.idx < 0.0f, negy = R.idy < 0.0f, negz = R.idz < 0.0f;
    const uint32_t nx = negx ? hx : lx, fx = negx ? lx : hx, ny = negy ? hy : ly, fy = negy ? ly : hy, nz = negz ? hz : lz, fz = negz ? lz : hz;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float tnx = __builtin_fmaf((float)((nx >> (8 * k)) & 0xffu), ax, bx), tfx = __builtin_fmaf((float)((fx >> (8 * k)) & 0xffu), ax, bx);
        const float tny = __builtin_fmaf((float)((ny >> (8 * k)) & 0xffu), ay, by), tfy = __builtin_fmaf((float)((fy >> (8 * k)) & 0xffu), ay, by);
        const float tnz = __builtin_fmaf((float)((nz >> (8 * k)) & 0xffu), az, bz), tfz = __builtin_fmaf((float)((fz >> (8 * k)) & 0xffu), az, bz);
        const float tns = __builtin_fmaf((float)((s_nw >> (8 * k)) & 0xffu), s_alpha, s_bn), tfs = __builtin_fmaf((float)((s_fw >> (8 * k)) & 0xffu), s_alpha, s_bf);
        // v_max3 / v_min3 written out: fmaxf on a value loaded from memory (tmin, ht) makes the compiler canonicalise it first, once per node step and operand
        const float cmin = max3_raw(max3_raw(tnx, tny, tnz), tns, tmin);
        const float cmax = min3_raw(min3_raw(tfx, tfy, tfz), tfs, ht);
        dd[k] = (cmax >= cmin) ? cmin : inf;   // no "child exists" test: a missing child's box is inverted (flatten.cpp) and is never entered
    }
    if (!ORDERED) {
        // any-hit traversal (round 5, profiles/r05_traversal_probes.log): the entered children in SLOT order, no ordering network — 6 selects instead of 5 compare-and-swaps
        const bool e0 = dd[0] < inf, e1 = dd[1] < inf, e2 = dd[2] < inf, e3 = dd[3] < inf;
        const int f23 = e2 ? c[2] : c[3], f123 = e1 ? c[1] : f23;
        const int o0 = e0 ? c[0] : f123, o1 = e0 ? f123 : (e1 ? f23 : c[3]), o2 = (e0 && e1) ? f23 : c[3];
        c[0] = o0; c[1] = o1; c[2] = o2;
        return (int)e0 + (int)e1 + (int)e2 + (int)e3;
    }
    CTL_CSWAP_PAIR(0, 1) CTL_CSWAP_PAIR(2, 3) CTL_CSWAP_PAIR(0, 2) CTL_CSWAP_PAIR(1, 3) CTL_CSWAP_PAIR(1, 2)
    return dd[3] < inf ? 4 : (dd[2] < inf ? 3 : (dd[1] < inf ? 2 : (dd[0] < inf ? 1 : 0)));
}

#undef CTL_CSWAP_PAIR

// The whole intersect kernel body over the flattened structure: `n` rays (ro, rd) -> hit / hit_node (closest) and/or occ (any-hit flag).
template <bool ANY_HIT, bool COUNT, bool ALPHA>
__device__ __forceinline__ void intersect_flat(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                               float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_stack_ints, trav_counts& cnt) {
    const int lane = threadIdx.x & 63;
    __shared__ unsigned int s_hist[COUNT ? kStackSize : 1];   // counting kernels: stack-depth histogram of this workgroup's rays, added to g_stack_hist at the end
    if (COUNT) { for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) s_hist[i] = 0u; __syncthreads(); }
    const int refill_idle = g_refill_idle, leaf_batch = ANY_HIT ? g_leaf_batch_any : g_leaf_batch;
    const bool compact = S.flat_compact != 0;
    flat_stack st; st.lds = (flat_stack_lds_word*)lds_stack_ints + threadIdx.x;
    bool has_ray = false;
    uint32_t ray_id = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, tmin = 0;
    ray_cull R{ 0, 0, 0, 0, 0, 0 };
    float ht = 0;                                 // distance of the closest hit so far; its record is in hit[] / hit_node[] already (hit_in_memory), bit 31 of ray_id says there is one
    hit_in_memory sink{ ht, ray_id, hit, hit_node, ANY_HIT ? nullptr : S.hit_key_out };
    int sp = 0, node = kSentinel, pend = -1;      // pend: postponed leaf (its first entry in flat_leaves), -1 = none
    alpha_cand cand{ 0, 0, 0, 0, 0, 0 }; bool has_cand = false;   // ALPHA builds: a geometric hit that waits for its alpha test (alpha phase below)
    constexpr int kAlphaBatch = 12;   // lanes with a candidate that start an alpha phase (4 / 8 / 12 / 20 / 32: 12.07 / 10.93 / 10.59 / 10.66 / 11.94 ms of traversal per pass on synthetic-sm-hard with AlphaTest)
    int sp_max = 0;                               // COUNT: deepest stack entry of the lane's current ray
    const float4* __restrict__ nodes = S.flat_nodes;
    ray_claims rc; rc.init(n);   // traverse.h: the first claim is static, the rest goes through the cursor
    uint32_t& chunk_next = rc.next; uint32_t& chunk_end = rc.end; bool& exhausted = rc.exhausted;

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
                    ox = o.x; oy = o.y; oz = o.z; tmin = o.w; dx = d.x; dy = d.y; dz = d.z;
                    R.idx = rcp_cull(dx); R.idy = rcp_cull(dy); R.idz = rcp_cull(dz);
                    R.oox = ox * R.idx; R.ooy = oy * R.idy; R.ooz = oz * R.idz;
                    ht = d.w;
                    sp = 0; st.put_row(0, kSentinel); node = S.flat_root; pend = -1;
                }
                chunk_next += want < avail ? want : avail;
            }
        }
        if (__ballot(has_ray) == 0ull) { if (exhausted) break; continue; }

        // ---- a lane standing on a leaf with a free slot postpones it and goes on with the next stack entry
        if (has_ray && !has_cand && node < 0 && pend < 0) { pend = ~node; node = st.get(sp); sp--; }
        const bool at_inner = has_ray && !has_cand && (unsigned)node < (unsigned)kSentinel;
        const bool at_leaf = has_ray && !has_cand && pend >= 0;
        const unsigned long long m_inner = __ballot(at_inner), m_leaf = __ballot(at_leaf);
        const unsigned long long m_cand = ALPHA ? __ballot(has_cand) : 0ull;
        bool finished = false;
        if (ALPHA && m_cand != 0ull && (__popcll(m_cand) >= kAlphaBatch || (m_inner == 0ull && m_leaf == 0ull))) {
            // ---- alpha phase (ALPHA builds): every lane that holds a geometric hit of an alpha-mapped material evaluates Material::AlphaTest for it now, together; a lane with a
            // candidate has stood still since it found it (the entries behind it are tested against the hit distance this decides)
            if (has_cand) {
                has_cand = false;
                if (alpha_survives(S.tri_data, S.node_info, S.mats, S.images, cand.tri, cand.node, cand.u, cand.v)) {
                    sink.accept(cand.t, cand.u, cand.v, cand.tri, cand.node, cand.key);
                    if (ANY_HIT) finished = true;
                }
            }
        } else if (m_leaf != 0ull && (__popcll(m_leaf) >= leaf_batch || m_inner == 0ull)) {
            // ---- leaf phase: every lane that holds a leaf tests its next entry
            if (at_leaf) {
                if (COUNT) { cnt.n_tri++; if (lane == (int)__builtin_ctzll(m_leaf)) cnt.w_tri++; }
                bool got = false;
                if (ALPHA) pend = flat_leaf_test<ANY_HIT, ALPHA>(S, (uint32_t)pend, ox, oy, oz, dx, dy, dz, tmin, sink, got, &cand, &has_cand);
                else pend = flat_leaf_test<ANY_HIT, ALPHA>(S, (uint32_t)pend, ox, oy, oz, dx, dy, dz, tmin, sink, got);
                if (ANY_HIT && got) finished = true;
            }
        } else {
            // ---- node phase
            if (at_inner) {
                if (COUNT) { cnt.n_inner++; if (lane == (int)__builtin_ctzll(m_inner)) cnt.w_inner++; }
                node_words W; node_fetch_own(nodes, node, compact, W);
                const int popped = st.get(sp);   // issued early: used when no child is entered
                int c[4]; float dd[4];
#ifdef CTL_ANYHIT_ORDERED   // measurement builds: the any-hit traversal with the closest-hit one's ordering network (the A side of profiles/r05_traversal_probes.log)
                const int n_hit = node_step_q4<true>(W, node, R, ox, oy, oz, dx, dy, dz, tmin, ht, c, dd, compact);
#else
                // shadow rays take the entered children in SLOT order: any hit ends the ray, the nearest-first order buys nothing (it visited 8 % MORE nodes) and costs 5 compare-and-swaps
                const int n_hit = node_step_q4<!ANY_HIT>(W, node, R, ox, oy, oz, dx, dy, dz, tmin, ht, c, dd, compact);
#endif
                node = n_hit ? c[0] : popped;
                const int top = sp + n_hit - 1;    // n_hit == 0: one entry popped
                if (top < kFlatLdsRows) {          // common case: unconditional LDS stores, unused ones into the spare row
                    st.put_row(n_hit >= 2 ? top : kFlatLdsRows, c[1]);
                    st.put_row(n_hit >= 3 ? top - 1 : kFlatLdsRows, c[2]);
                    st.put_row(n_hit >= 4 ? top - 2 : kFlatLdsRows, c[3]);
                } else {
                    if (n_hit >= 4) st.set(top - 2, c[3]);
                    if (n_hit >= 3) st.set(top - 1, c[2]);
                    if (n_hit >= 2) st.set(top, c[1]);
                }
                sp = top;
                if (COUNT && sp > sp_max) sp_max = sp;
            }
        }
        if (has_ray && !finished) finished = (node == kSentinel) && pend < 0 && !has_cand;
        if (finished) {
            const uint32_t id = ray_id & 0x7fffffffu; const bool found = (ray_id >> 31) != 0u;
            if (ANY_HIT && occ) occ[id] = found ? 1u : 0u;
            if (hit && !found) { hit[id] = make_float4(ht, 0.0f, 0.0f, __int_as_float(-1)); hit_node[id] = -1; if (!ANY_HIT && S.hit_key_out) S.hit_key_out[id] = 0; }   // a found hit wrote its record when it was accepted
            if (COUNT) { atomicAdd(&s_hist[sp_max < kStackSize ? sp_max : kStackSize - 1], 1u); sp_max = 0; }   // the workgroup's own histogram in LDS: one global atomic per ray on two dozen addresses made the counting kernels 15 x slower than the timed ones
            has_ray = false; node = kSentinel; pend = -1;
        }
    }
    if (COUNT) { __syncthreads(); for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) if (s_hist[i]) atomicAdd(&g_stack_hist[i], (unsigned long long)s_hist[i]); }
}

// Single-ray form for the megakernel plugin (one lane walks a whole path): same node steps, same entry test.  The stack's first kSingleLdsRows entries live in LDS
// (`lds_col` = this lane's column of a [row][256] array, as in the wavefront kernel), deeper ones in a private array: a stack in scratch alone made every push and
// pop a trip through the vector memory path.
constexpr int kSingleLdsRows = 20;
typedef __attribute__((address_space(3))) int lds_int;
struct single_stack {
    lds_int* lds; int ovf[kStackSize - kSingleLdsRows];
    __device__ __forceinline__ int get(int i) const { int v = lds[(i < kSingleLdsRows ? i : 0) * 256]; if (i >= kSingleLdsRows) v = ovf[i - kSingleLdsRows]; return v; }
    __device__ __forceinline__ void set(int i, int v) { if (i < kSingleLdsRows) lds[i * 256] = v; else ovf[i - kSingleLdsRows] = v; }
};
template <bool ANY_HIT, bool ALPHA_DYNAMIC>
__device__ bool trace_single_flat(const dev_scene& S, lds_int* lds_col, f3 o, f3 d, float tmin, float tmax, float& ht, float& hu, float& hv, int& htri, int& hnode) {
    const float4* __restrict__ nodes = S.flat_nodes;
    ray_cull R;
    R.idx = rcp_cull(d.x); R.idy = rcp_cull(d.y); R.idz = rcp_cull(d.z);
    R.oox = o.x * R.idx; R.ooy = o.y * R.idy; R.ooz = o.z * R.idz;
    single_stack stack; stack.lds = lds_col; int sp = 0; stack.set(0, kSentinel);
    int node = S.flat_root;
    const bool compact = S.flat_compact != 0;
    ht = tmax; hu = hv = 0.0f; htri = -1; hnode = -1;
    while (node != kSentinel) {
        if (node >= 0) {
            int c[4]; float dd[4];
            node_words W; node_fetch_own(nodes, node, compact, W);
            const int n_hit = node_step_q4(W, node, R, o.x, o.y, o.z, d.x, d.y, d.z, tmin, ht, c, dd, compact);
            for (int i = n_hit - 1; i >= 1; i--) stack.set(++sp, c[i]);
            if (n_hit) node = c[0]; else { node = stack.get(sp); sp--; }
        } else {
            bool got = false; int next = ~node;
            // USE_ALPHA of __traceRay_internal__ (TraceHelper.cu:135-153): scenes with alpha maps test every candidate hit
            hit_in_regs sink{ ht, hu, hv, htri, hnode };
            while (next >= 0 && !(ANY_HIT && got))
                next = (ALPHA_DYNAMIC && S.alpha_maps) ? flat_leaf_test<ANY_HIT, true>(S, (uint32_t)next, o.x, o.y, o.z, d.x, d.y, d.z, tmin, sink, got)
                                                       : flat_leaf_test<ANY_HIT, false>(S, (uint32_t)next, o.x, o.y, o.z, d.x, d.y, d.z, tmin, sink, got);
            if (ANY_HIT && got) return true;
            node = stack.get(sp); sp--;
        }
    }
    return htri >= 0;
}

} // namespace ctl
