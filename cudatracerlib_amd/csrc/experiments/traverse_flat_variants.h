// experiments/traverse_flat_variants.h — the traversal header as it stood at the end of round 3, with every measured-and-lost variant arm (DESIGN.md §3 is the record): built INSTEAD of
// ../traverse_flat.h by -DCTL_FLAT_EXPERIMENTS (no 8-wide format in such a build).  Original head comment:
// traverse_flat.h — traversal of the flattened world-space BVH (flatten.h) by persistent wave64 waves, one ray per lane.
//
// What is reported is the reference's: every leaf entry is evaluated with the two-level arithmetic of intersectKernel
// (Kernel/TraceHelper.cu:526-560 ray into the instance's object space, :646-682 Woop test with an exact division), in the
// reference's expression order, so (t, u, v, triangle, node) are bit-identical to the two-level traversal (traverse.h) whenever
// both look at the winning triangle; the world-space tree only culls.  Rays that hit two triangles at exactly the same t may
// report either (the visiting order differs), as between any two BVHs.
//
// Execution model (measured on MI355X, profiles/r02*, r03*).  A node step ends in ONE wait for the slowest of the wave's ~54 active lanes, and with a sixth of the
// node fetches going past the L2 practically every step waits for a memory-side line: an iteration is a memory round trip under load plus ~900 issue cycles, seven waves deep.  At
// the end of round 3 the kernel sits at the knee between the two: 8 % fewer VALU instructions bought 1.7 %, 11 % more cost 2.6 %, 22 % more cost 16 % (CTL_EXTRA_VALU) — and taking
// whole node visits away pays one for one.  Per-lane L1 accesses (one 16-B load of one lane = one L1 cycle, tools/gather_probe.hip) and HBM-side lines (0.46 of peak) are both below
// their ceilings.  So the design spends arithmetic where it removes fetches or visits, and nowhere else:
//  * lane refill as in traverse.h (a wave claims rays from a device cursor, idle lanes are refilled together);
//  * node steps and leaf steps are separate wave-wide phases.  A lane that reaches a leaf POSTPONES it (one pending leaf per
//    lane) and keeps descending the tree speculatively; the wave runs the leaf code only once enough lanes hold a pending leaf
//    (or nobody has an inner node left); a leaf step tests one entry per lane.  In a unified loop the Woop code ran with ~8 of 64 lanes;
//  * the node step is branch-free: children sorted by entry distance with a compare + select network, pushes are unconditional LDS
//    stores (unused ones land in a spare row), the pop is an LDS read issued before the slab arithmetic;
//  * F4 nodes are plane-major: the sign of the ray direction picks the near / far plane of all four children by address.
#pragma once
#include "../traverse.h"
#include "../flat_slab.h"

namespace ctl {

enum { kFmtQ4 = 0, kFmtF4 = 1, kFmtF2 = 2, kFmtQ8 = 3 };   // = flat_format (flatten.h); Q8 has its own kernel body (traverse_flat8.h)

// Measured and NOT shipped (round 3, tools/ab_libs.sh r03e, same box): the first kTopCache nodes of the node array — the top of the tree, which flatten.cpp stores breadth-first:
// 85 = four full levels — copied into LDS by every traversal workgroup (48 B each; a node with a slab goes the global way).  Counted by the oracle on the bench scene
// (tools/bvh_quality_probe.py): 32 % of all node visits are visits of these 85 nodes (41 % for 341, 50 % for 1365).  Result: 2341 Mrays/s against 2457 without (64 / 128 nodes:
// 2335 / 2340) — a wave's lanes are spread over all levels, so every step runs the LDS arm AND the global arm one after the other, and what the kernel is short of is
// time per step, not L1 look-ups.  CTL_TOP_CACHE=85 CTL_FLAT_LDS_ROWS=17 rebuilds that variant.
#ifndef CTL_TOP_CACHE
#define CTL_TOP_CACHE 0
#endif
#ifndef CTL_FLAT_LDS_ROWS
#define CTL_FLAT_LDS_ROWS 19
#endif
constexpr int kTopCache = CTL_TOP_CACHE;
constexpr int kTopCacheFloats = (kTopCache ? kTopCache : 1) * 12;
typedef float __attribute__((ext_vector_type(4))) lds_f4v;
typedef __attribute__((address_space(3))) const lds_f4v lds_top_f4;
// called by the whole workgroup before the traversal: LDS <- the first 48 B of the first S.flat_top_cached nodes
__device__ __forceinline__ void fill_top_cache(const dev_scene& S, float* lds_top) {
    if (!kTopCache) return;
    const int n = S.flat_top_cached * 3;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ((float4*)lds_top)[i] = S.flat_nodes[(i / 3) * 4 + (i % 3)];
    __syncthreads();
}
constexpr int kFlatLdsRows = CTL_FLAT_LDS_ROWS;   // stack entries per lane in LDS (+ 1 spare row); deeper entries live in scratch
__device__ unsigned long long g_stack_hist[kStackSize];   // counting kernels only: rays by the deepest traversal-stack entry they used (ctl_traversal_stack_histogram)
__device__ int g_leaf_batch = 16;         // run the leaf phase once this many lanes hold a pending leaf entry (knob CTL_LEAF_BATCH).  With the oriented slabs fewer leaves are parked: 8: 2253, 12: 2299, 16: 2315, 20: 2311, 24: 2288, 32: 2216 Mrays/s (gpurun_out r03g / r03h)

// Stack entry = {link, entry distance of the pushed child}: a pop whose entry distance is not below the current hit distance is dropped on the spot
// (the closest hit moved in front of that child while it waited) — counted by the oracle on the bench scene: 10 % of the node visits, 5 % of the entry tests.
// 8 B per entry: 40 KiB of LDS per 256-lane workgroup, four workgroups per CU (the kernel runs as fast at 4 waves per SIMD as at 6).
// Round-3 experiments on the node fetch, all measured on one box against the shipped order of loads (2462 Mrays/s; tools/ab_libs.sh r03c / r03d), none shipped:
//   CTL_NODE_FETCH_QUAD  the four lanes of a quad read each other's nodes as contiguous 64 B and transpose them in registers (node_fetch_quad below): 2125
//   CTL_PARK_IN_STEP     a node step whose nearest child is a leaf parks it at once and walks on with the second nearest (no trip through the stack): 2461 / 2441
//   CTL_PREFETCH_NODE    the step touches one dword of the node it goes to next (consumed after the next step's own loads): 2044
//   CTL_PREFETCH_LEAF    parking a leaf touches the line of its first entry: 2413
//   merged iteration     (removed again) entry tests and node steps in ONE iteration, a lane doing either, their loads sharing registers and one wait: node-step lane
//                        utilisation 0.58 -> 0.70, a third fewer iterations, but every iteration pays for both code paths: 2127 at 6 waves, 2173 at 5 (spills at 7: 1430)
//   packed slab FMAs     (removed again) entry and exit distance of an axis in one v_pk_fma_f32, 16 instead of 32 FMA instructions per node step: 2461 against 2527 —
//                        the packed instruction is no faster than the two it replaces, and its operands want pairing moves
//   8 waves per SIMD     with CTL_LEAN_RAY (no o * idir per lane: three VALU more per step, three registers fewer) the kernels fit 64 VGPRs without spills: 2543 against
//                        2543 at 7 waves — occupancy no longer buys anything, the kernel is bound by VALU issue now
//   overlapped refill    (removed again) the ray loads of idle lanes issued in the same iteration as the other lanes' node / entry loads, their traversal state set up at
//                        the iteration's end: 16.3-16.5 ms per fused launch against 15.5 at every refill threshold from 2 to 12 — and node-step lane utilisation moves only
//                        from 0.61 to 0.64 even when every idle lane is refilled at once: lanes are not waiting for rays, they wait for their last entry test
//   comparators in VCC   (removed again) the ordering network's five compare results in VCC instead of SGPR pairs (v_cndmask_b32_e32 instead of _e64, inline asm): 15.33 against
//                        15.32 ms — no single unit is the limit any more: VALU issue, L1 lane-loads and HBM lines all sit at 55-75 % of what the probes give them alone
// What DID pay: issuing the load of the slab quarter together with the other three (node_fetch_own) instead of after the link arithmetic that waits for them — the second
// round trip through the L1 per slab node was 9 % of the whole job (2279 -> 2486).
#ifndef CTL_LEAN_RAY
#define CTL_LEAN_RAY 0
#endif
#ifndef CTL_NODE_FETCH_QUAD
#define CTL_NODE_FETCH_QUAD 0
#endif
#ifndef CTL_PARK_IN_STEP
#define CTL_PARK_IN_STEP 0
#endif
#ifndef CTL_PREFETCH_NODE
#define CTL_PREFETCH_NODE 0
#endif
#ifndef CTL_PREFETCH_LEAF
#define CTL_PREFETCH_LEAF 0
#endif
#ifndef CTL_STACK_DIST
#define CTL_STACK_DIST 0
#endif
#ifndef CTL_EXTRA_VALU
#define CTL_EXTRA_VALU 0   // pairs of extra VALU instructions per node step (a measurement: what does the kernel's time do when its instruction count moves?)
#endif
// CTL_STACK_DIST: 0 = links only (shipped); 1 = 8-byte LDS entries {link, entry distance}; 2 = the links as in 0 plus a second LDS array with the TOP 16 BITS of the entry distance
// (truncation rounds a positive float down, so the stored distance never exceeds the true one and a cull stays conservative): 6 B per entry, kFlatLdsRows = 13 keeps seven workgroups per CU.
struct stack_entry { int link; float dist; };
constexpr int kStaleLink = 0x76543211;   // mode 3: "pop again" (neither an inner link — those lie below kSentinel — nor a leaf link nor kSentinel itself)
#if CTL_STACK_DIST == 1
typedef unsigned long long flat_stack_word;   // link in the low word, distance bits in the high word (a scalar type: it can live behind an address-space-qualified pointer)
__device__ __forceinline__ flat_stack_word stack_word(int link, float dist) { return (unsigned long long)(uint32_t)link | ((unsigned long long)__float_as_uint(dist) << 32); }
__device__ __forceinline__ stack_entry stack_unpack(const flat_stack_word& w) { return stack_entry{ (int)(uint32_t)w, __uint_as_float((uint32_t)(w >> 32)) }; }
#else
typedef int flat_stack_word;
__device__ __forceinline__ flat_stack_word stack_word(int link, float) { return link; }
__device__ __forceinline__ stack_entry stack_unpack(const flat_stack_word& w) { return stack_entry{ w, -__builtin_huge_valf() }; }
#endif
__device__ __forceinline__ bool stack_entry_culled(const stack_entry& e, float ht) { return CTL_STACK_DIST ? e.dist >= ht : false; }
constexpr int kFlatStackInts = (int)(sizeof(flat_stack_word) / sizeof(int));   // ints of LDS per stack entry (link array)
typedef __attribute__((address_space(3))) flat_stack_word flat_stack_lds_word;   // explicitly LDS: the pushes must compile to ds_write, not to generic flat stores
typedef __attribute__((address_space(3))) uint16_t flat_stack_lds_dist;
struct flat_stack {
    flat_stack_lds_word* lds;             // this lane's column, stride 256
    flat_stack_word ovf[kStackSize - kFlatLdsRows];
#if CTL_STACK_DIST >= 2
    flat_stack_lds_dist* ldsd;            // this lane's column of the distance array, stride 256
    uint16_t ovfd[kStackSize - kFlatLdsRows];
#endif
    __device__ __forceinline__ stack_entry get(int i, const bool wd = true) const {
        const int row = i < kFlatLdsRows ? i : kFlatLdsRows;
        flat_stack_word w = lds[row * 256];                                     // a ds_read whatever the depth (the spare row when the entry lives in scratch) ...
#if CTL_STACK_DIST >= 2
        uint32_t dh = 0xff80u; if (wd) dh = ldsd[row * 256];
#endif
        if (i >= kFlatLdsRows) {                                                // ... and the rare deep entry from scratch
            w = ovf[i - kFlatLdsRows];
#if CTL_STACK_DIST >= 2
            if (wd) dh = ovfd[i - kFlatLdsRows];
#endif
        }
        stack_entry e = stack_unpack(w);
#if CTL_STACK_DIST >= 2
        e.dist = __uint_as_float(dh << 16);
#endif
        return e;
    }
    // store into LDS row `row` (the spare row kFlatLdsRows absorbs unused push slots)
    __device__ __forceinline__ void put_row(int row, int link, float dist, const bool wd = true) {
        lds[row * 256] = stack_word(link, dist);
#if CTL_STACK_DIST >= 2
        if (wd) ldsd[row * 256] = (uint16_t)(__float_as_uint(dist) >> 16);
#endif
    }
    __device__ __forceinline__ void set(int i, int link, float dist, const bool wd = true) {
        if (i < kFlatLdsRows) put_row(i, link, dist, wd);
        else {
            ovf[i - kFlatLdsRows] = stack_word(link, dist);
#if CTL_STACK_DIST >= 2
            if (wd) ovfd[i - kFlatLdsRows] = (uint16_t)(__float_as_uint(dist) >> 16);
#endif
        }
    }
    // pop entries until one is worth visiting (the sentinel at the bottom carries -inf and always is)
    // mode 3: ONE pop; an entry that fell behind the hit comes back as kStaleLink and the lane pops again in the next iteration (no wave-wide loop of LDS round trips)
    __device__ __forceinline__ int pop_once(int& sp, float ht, const bool wd = true) const { const stack_entry e = get(sp, wd); sp--; return (wd && stack_entry_culled(e, ht)) ? kStaleLink : e.link; }
    __device__ __forceinline__ int pop(int& sp, float ht, const bool wd = true) const { stack_entry e; do { e = get(sp, wd); sp--; } while (wd && stack_entry_culled(e, ht)); return e.link; }
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
    __device__ __forceinline__ void accept(float t, float u, float v, int tri, int nd) { ht = t; hu = u; hv = v; htri = tri; hnode = nd; }
};
struct hit_in_memory {
    float& ht; uint32_t& ray_word; float4* __restrict__ hit; int* __restrict__ hit_node;   // ray_word: ray index | found << 31
    __device__ __forceinline__ float dist() const { return ht; }
    __device__ __forceinline__ void accept(float t, float u, float v, int tri, int nd) {
        ht = t;
        if (hit) { const uint32_t id = ray_word & 0x7fffffffu; hit[id] = make_float4(t, u, v, __int_as_float(tri)); hit_node[id] = nd; }
        ray_word |= 0x80000000u;
    }
};
template <bool ALPHA, class SINK>
__device__ __forceinline__ bool flat_woop_test(const dev_scene& S, const float4 v00, const float4 v11, const float4 v22, uint32_t index, int nd, const f3 o, const f3 d, float tmin, SINK& sink) {
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
            if (v >= 0.0f && u + v <= 1.0f && (!ALPHA || alpha_survives(S.tri_data, S.node_info, S.mats, S.images, (int)(index >> 1), nd, u, v))) {
                sink.accept(t, u, v, (int)(index >> 1), nd);
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
                                              SINK& sink, bool& got) {
    m34 m; m.r[0][0] = L.r0.x; m.r[0][1] = L.r0.y; m.r[0][2] = L.r0.z; m.r[0][3] = L.r0.w; m.r[1][0] = L.r1.x; m.r[1][1] = L.r1.y; m.r[1][2] = L.r1.z; m.r[1][3] = L.r1.w;
    m.r[2][0] = L.r2.x; m.r[2][1] = L.r2.y; m.r[2][2] = L.r2.z; m.r[2][3] = L.r2.w;
    const f3 d = xform_dir(m, f3(dirx, diry, dirz));
    f3 o = xform_point(m, f3(orgx, orgy, orgz));
    if (!S.inst_w_one) o = f3(o.x / L.w33, o.y / L.w33, o.z / L.w33);
    if (flat_woop_test<ALPHA>(S, L.v00, L.v11, L.v22, L.iw.x, (int)L.iw.y, o, d, tmin, sink)) { got = true; if (ANY_HIT) return -1; }
    return (L.iw.x & 1u) ? -1 : (int)(e + 1);
}
// One leaf entry (flat_leaf, 128 B) against the world-space ray: the ray through the node's inverse transform (TraceHelper.cu:526-560; the rows
// travel with the entry), then the Woop test.  Returns the next entry of the leaf, or -1 when this was its last one.
template <bool ANY_HIT, bool ALPHA, class SINK>
__device__ __forceinline__ int flat_leaf_test(const dev_scene& S, uint32_t e, float orgx, float orgy, float orgz, float dirx, float diry, float dirz, float tmin, SINK& sink, bool& got) {
    leaf_words L; flat_leaf_load(S, e, L);
    return flat_leaf_eval<ANY_HIT, ALPHA>(S, e, L, orgx, orgy, orgz, dirx, diry, dirz, tmin, sink, got);
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

// F4: 128-B plane-major node (flat4f_node)
__device__ __forceinline__ int node_step_f4(const float4* __restrict__ nodes, int node, const ray_cull& R, float tmin, float ht, int c[4], float dd[4]) {
    const float4* __restrict__ p = nodes + node;
    const float4 nx = p[R.sx()], fx = p[1 - R.sx()], ny = p[2 + R.sy()], fy = p[3 - R.sy()], nz = p[4 + R.sz()], fz = p[5 - R.sz()];
    const float4 lk = p[6];
    const float nxa[4] = { nx.x, nx.y, nx.z, nx.w }, fxa[4] = { fx.x, fx.y, fx.z, fx.w }, nya[4] = { ny.x, ny.y, ny.z, ny.w }, fya[4] = { fy.x, fy.y, fy.z, fy.w };
    const float nza[4] = { nz.x, nz.y, nz.z, nz.w }, fza[4] = { fz.x, fz.y, fz.z, fz.w };
    c[0] = __float_as_int(lk.x); c[1] = __float_as_int(lk.y); c[2] = __float_as_int(lk.z); c[3] = __float_as_int(lk.w);
    const float inf = __builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float tnx = __builtin_fmaf(nxa[k], R.idx, -R.oox), tfx = __builtin_fmaf(fxa[k], R.idx, -R.oox);
        const float tny = __builtin_fmaf(nya[k], R.idy, -R.ooy), tfy = __builtin_fmaf(fya[k], R.idy, -R.ooy);
        const float tnz = __builtin_fmaf(nza[k], R.idz, -R.ooz), tfz = __builtin_fmaf(fza[k], R.idz, -R.ooz);
        const float cmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
        const float cmax = fminf(fminf(tfx, tfy), fminf(tfz, ht));
        dd[k] = (cmax >= cmin) ? cmin : inf;
    }
    CTL_CSWAP_PAIR(0, 1) CTL_CSWAP_PAIR(2, 3) CTL_CSWAP_PAIR(0, 2) CTL_CSWAP_PAIR(1, 3) CTL_CSWAP_PAIR(1, 2)
    return dd[3] < inf ? 4 : (dd[2] < inf ? 3 : (dd[1] < inf ? 2 : (dd[0] < inf ? 1 : 0)));
}

// Q4: 64-B node with 8-bit child boxes (flat4_node).  compact: the child links are implied by the layout and only the first 48 B are
// loaded — three per-lane L1 accesses instead of four — unless the link that led here says the node carries an oriented slab (bit 0; flat_slab.h):
// then the last 16 B are loaded too and every child's entry / exit distance is clipped by its interval along the node's slab direction.
//
// How the 48 / 64 B reach the lane (node_words).  node_fetch_own: three / four global_load_dwordx4 of the lane's own node — 64 lanes, 64 different lines, and the L1 looks every
// lane-load up on its own: 3-4 tag look-ups per node, which is what the kernel's time floor is made of (tools/coop_probe.hip: 174-220 G records/s for L2-resident records this way,
// 440-520 G/s when four lanes read one record as one contiguous 64 B).  node_fetch_quad: in round r the four lanes of a quad read the node of the quad's lane r, 16 B each — one
// look-up per node — and the 4 x 4 block of 16-B pieces is transposed inside the quad by two butterfly stages of v_cndmask_b32_dpp (32 VALU, no LDS), after which every lane holds
// its own node exactly as node_fetch_own would have loaded it.  Lanes of the wave that are not on a node take part in the loads of their quad's other lanes.
struct node_words { float4 q0, q1, q2, q3; };
__device__ __forceinline__ void node_fetch_own(const float4* __restrict__ nodes, int node, bool compact, node_words& W) {
    const float4* __restrict__ p = nodes + (node & ~3);
    W.q0 = p[0]; W.q1 = p[1]; W.q2 = p[2];
    if (!compact || (node & 1)) W.q3 = p[3];
}
// d = take_b ? b : a[lane ^ 1]   (dword by dword; take_b is a wave mask)
#define CTL_SEL4_DPP(NAME, PERM) \
__device__ __forceinline__ uint4 NAME(const uint4 a, const uint4 b, unsigned long long take_b) { \
    uint4 d; \
    asm("s_mov_b64 vcc, %12\n\ts_nop 1\n\t" \
        "v_cndmask_b32_dpp %0, %4, %8, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t" \
        "v_cndmask_b32_dpp %1, %5, %9, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t" \
        "v_cndmask_b32_dpp %2, %6, %10, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t" \
        "v_cndmask_b32_dpp %3, %7, %11, vcc " PERM " row_mask:0xf bank_mask:0xf" \
        : "=&v"(d.x), "=&v"(d.y), "=&v"(d.z), "=&v"(d.w) \
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "s"(take_b) : "vcc"); \
    return d; \
}
CTL_SEL4_DPP(sel4_quad_xor1, "quad_perm:[1,0,3,2]")
CTL_SEL4_DPP(sel4_quad_xor2, "quad_perm:[2,3,0,1]")
#undef CTL_SEL4_DPP
__device__ __forceinline__ float4 as_f4(const uint4 v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
// node: the lane's node link (index << 2 | slab flag) or anything when !active; called by the whole wave
__device__ __forceinline__ void node_fetch_quad(const float4* __restrict__ nodes, int node, bool active, bool compact, node_words& W) {
    const int sub = threadIdx.x & 3;
    const int key = active ? node : -1;
    uint4 t[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int k = r == 0 ? __builtin_amdgcn_mov_dpp(key, 0x00, 0xf, 0xf, false) : r == 1 ? __builtin_amdgcn_mov_dpp(key, 0x55, 0xf, 0xf, false)
                    : r == 2 ? __builtin_amdgcn_mov_dpp(key, 0xaa, 0xf, 0xf, false) : __builtin_amdgcn_mov_dpp(key, 0xff, 0xf, 0xf, false);
        t[r] = make_uint4(0u, 0u, 0u, 0u);
        if (k >= 0 && (sub != 3 || (k & 1) || !compact)) t[r] = *(const uint4*)(nodes + (k & ~3) + sub);   // the last quarter only where the owner's node carries a slab
    }
    // 4 x 4 transpose over (lane of the quad, round): piece (lane p, round q) = quarter p of the node of lane q  ->  (lane q, register p)
    const unsigned long long odd1 = 0xaaaaaaaaaaaaaaaaull, odd2 = 0xccccccccccccccccull;   // lanes with bit 0 / bit 1 of the lane number set
    const uint4 u0 = sel4_quad_xor1(t[1], t[0], ~odd1), u1 = sel4_quad_xor1(t[0], t[1], odd1);
    const uint4 u2 = sel4_quad_xor1(t[3], t[2], ~odd1), u3 = sel4_quad_xor1(t[2], t[3], odd1);
    W.q0 = as_f4(sel4_quad_xor2(u2, u0, ~odd2)); W.q2 = as_f4(sel4_quad_xor2(u0, u2, odd2));
    W.q1 = as_f4(sel4_quad_xor2(u3, u1, ~odd2)); W.q3 = as_f4(sel4_quad_xor2(u1, u3, odd2));
}

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
#if CTL_LEAN_RAY
    const float bx = (q0.x - ox) * R.idx, by = (q0.y - oy) * R.idy, bz = (q0.z - oz) * R.idz;   // no o * idir kept per lane
#else
    const float bx = __builtin_fmaf(q0.x, R.idx, -R.oox), by = __builtin_fmaf(q0.y, R.idy, -R.ooy), bz = __builtin_fmaf(q0.z, R.idz, -R.ooz);
#endif
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
    CTL_CSWAP_PAIR(0, 1) CTL_CSWAP_PAIR(2, 3) CTL_CSWAP_PAIR(0, 2) CTL_CSWAP_PAIR(1, 3) CTL_CSWAP_PAIR(1, 2)
#if CTL_EXTRA_VALU   // measurement: N more integer VALU instructions per node step, of the kind the link decode is made of (v_bfe / v_add / v_and on a value nothing else waits for)
    { uint32_t x = meta;
#pragma unroll
      for (int e = 0; e < CTL_EXTRA_VALU; e++) asm volatile("v_bfe_u32 %0, %0, 1, 31\n\tv_add_u32 %0, %0, %1" : "+v"(x) : "v"(meta));
      asm volatile("" :: "v"(x)); }
#endif
    return dd[3] < inf ? 4 : (dd[2] < inf ? 3 : (dd[1] < inf ? 2 : (dd[0] < inf ? 1 : 0)));
}

// F2: the reference's BVHNodeData (two fp32 child boxes, 64 B)
__device__ __forceinline__ int node_step_f2(const float4* __restrict__ nodes, int node, const ray_cull& R, float tmin, float ht, int c[4], float dd[4]) {
    const float4* __restrict__ p = nodes + node;
    const float4 n0 = p[0], n1 = p[1], nz = p[2], cn = p[3];
    float c0min, c0max, c1min, c1max;
    slab2(n0, n1, nz, R.idx, R.idy, R.idz, R.oox, R.ooy, R.ooz, tmin, ht, c0min, c0max, c1min, c1max);
    const bool t0 = c0max >= c0min, t1 = c1max >= c1min;
    const int k0 = __float_as_int(cn.x), k1 = __float_as_int(cn.y);
    const bool swap = t1 && (!t0 || c1min < c0min);   // child 1 first
    c[0] = swap ? k1 : k0; c[1] = swap ? k0 : k1;
    dd[0] = swap ? c1min : c0min; dd[1] = swap ? c0min : c1min;
    return (t0 ? 1 : 0) + (t1 ? 1 : 0);
}
#undef CTL_CSWAP_PAIR

// The whole intersect kernel body over the flattened structure: `n` rays (ro, rd) -> hit / hit_node (closest) and/or occ (any-hit flag).
template <bool ANY_HIT, bool COUNT, bool ALPHA, int FMT>
__device__ __forceinline__ void intersect_flat(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                               float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_stack_ints, const float* lds_top_floats, trav_counts& cnt, uint16_t* lds_dist = nullptr) {
    flat_stack_lds_word* lds_stack = (flat_stack_lds_word*)lds_stack_ints;
    lds_top_f4* lds_top = (lds_top_f4*)lds_top_floats;
    const uint32_t n_top = (FMT == kFmtQ4 && kTopCache) ? (uint32_t)S.flat_top_cached : 0u;
    const int lane = threadIdx.x & 63;
    __shared__ unsigned int s_hist[COUNT ? kStackSize : 1];   // counting kernels: stack-depth histogram of this workgroup's rays, added to g_stack_hist at the end
    if (COUNT) { for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) s_hist[i] = 0u; __syncthreads(); }
    const int refill_idle = g_refill_idle, leaf_batch = g_leaf_batch;
    const bool compact = S.flat_compact != 0;
    constexpr bool kWithDist = CTL_STACK_DIST != 0 && !ANY_HIT;   // an any-hit ray ends at its first hit: no entry of its stack ever falls behind one
    flat_stack st; st.lds = lds_stack + threadIdx.x;
#if CTL_STACK_DIST >= 2
    st.ldsd = (flat_stack_lds_dist*)lds_dist + threadIdx.x;
#endif
    bool has_ray = false;
    uint32_t ray_id = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, tmin = 0;
    ray_cull R{ 0, 0, 0, 0, 0, 0 };
    float ht = 0;                                 // distance of the closest hit so far; its record is in hit[] / hit_node[] already (hit_in_memory), bit 31 of ray_id says there is one
    hit_in_memory sink{ ht, ray_id, hit, hit_node };
    int sp = 0, node = kSentinel, pend = -1;      // pend: postponed leaf (its first entry in flat_leaves), -1 = none
    int sp_max = 0;                               // COUNT: deepest stack entry of the lane's current ray
    uint32_t pf_node = 0u, pf_leaf = 0u;          // touched dwords (CTL_PREFETCH_*): loads whose only purpose is to start the line's way into the L1 early
    const float4* __restrict__ nodes = S.flat_nodes;
    uint32_t chunk_next = 0, chunk_end = 0; bool exhausted = (n == 0);

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
                    ox = o.x; oy = o.y; oz = o.z; tmin = o.w; dx = d.x; dy = d.y; dz = d.z;
                    R.idx = rcp_cull(dx); R.idy = rcp_cull(dy); R.idz = rcp_cull(dz);
                    R.oox = ox * R.idx; R.ooy = oy * R.idy; R.ooz = oz * R.idz;
                    ht = d.w;
                    sp = 0; st.put_row(0, kSentinel, -__builtin_huge_valf(), kWithDist); node = S.flat_root; pend = -1;
                }
                chunk_next += want < avail ? want : avail;
            }
        }
        if (__ballot(has_ray) == 0ull) { if (exhausted) break; continue; }

        // ---- a lane standing on a leaf with a free slot postpones it and goes on with the next stack entry
        if (CTL_STACK_DIST == 3 && kWithDist) {
            if (has_ray && ((node < 0 && pend < 0) || node == kStaleLink)) { if (node < 0) pend = ~node; node = st.pop_once(sp, ht, true); }
        } else if (has_ray && node < 0 && pend < 0) {
            pend = ~node; node = st.pop(sp, ht, kWithDist);
            if (CTL_PREFETCH_LEAF) pf_leaf = *(const uint32_t*)(S.flat_leaves + (size_t)(uint32_t)pend * 8);
        }
        const bool at_inner = has_ray && (unsigned)node < (unsigned)kSentinel;
        const bool at_leaf = has_ray && pend >= 0;
        const unsigned long long m_inner = __ballot(at_inner), m_leaf = __ballot(at_leaf);
        bool finished = false;
        if (m_leaf != 0ull && (__popcll(m_leaf) >= leaf_batch || m_inner == 0ull)) {
            // ---- leaf phase: every lane that holds a leaf tests its next entry
            if (at_leaf) {
                if (COUNT) { cnt.n_tri++; if (lane == (int)__builtin_ctzll(m_leaf)) cnt.w_tri++; }
                bool got = false;
                pend = flat_leaf_test<ANY_HIT, ALPHA>(S, (uint32_t)pend, ox, oy, oz, dx, dy, dz, tmin, sink, got);
                if (CTL_PREFETCH_LEAF) asm volatile("" :: "v"(pf_leaf));
                if (ANY_HIT && got) finished = true;
            }
        } else {
            // ---- node phase
#ifdef CTL_COUNT_PROBE   // where do the lanes that take no node step stand?  n_inst (unused by the flattened layout) counts one category per build: 1 blocked on a second leaf, 2 stack exhausted and waiting for the last entry test, 3 without a ray
            if (COUNT && m_inner != 0ull) {
                if (CTL_COUNT_PROBE == 1 && has_ray && pend >= 0 && node < 0) cnt.n_inst++;
                if (CTL_COUNT_PROBE == 2 && has_ray && pend >= 0 && node == kSentinel) cnt.n_inst++;
                if (CTL_COUNT_PROBE == 3 && !has_ray) cnt.n_inst++;
            }
#endif
            node_words W;
            if (FMT == kFmtQ4 && CTL_NODE_FETCH_QUAD) node_fetch_quad(nodes, node, at_inner, compact, W);
            if (at_inner) {
                if (COUNT) { cnt.n_inner++; if (lane == (int)__builtin_ctzll(m_inner)) cnt.w_inner++; }
                const stack_entry popped = st.get(sp, kWithDist);   // issued early: used when no child is entered
                int c[4]; float dd[4]; int n_hit;
                if (FMT == kFmtF4) n_hit = node_step_f4(nodes, node, R, tmin, ht, c, dd);
                else if (FMT == kFmtQ4) {
                    if (!CTL_NODE_FETCH_QUAD) {
                        const uint32_t ni = (uint32_t)node >> 2;
                        if (kTopCache && ni < n_top && !(node & 1)) {
                            const lds_f4v a = lds_top[ni * 3], b = lds_top[ni * 3 + 1], cc = lds_top[ni * 3 + 2];
                            W.q0 = make_float4(a.x, a.y, a.z, a.w); W.q1 = make_float4(b.x, b.y, b.z, b.w); W.q2 = make_float4(cc.x, cc.y, cc.z, cc.w);
                        } else node_fetch_own(nodes, node, compact, W);
                    }
                    n_hit = node_step_q4(W, node, R, ox, oy, oz, dx, dy, dz, tmin, ht, c, dd, compact);
                }
                else n_hit = node_step_f2(nodes, node, R, tmin, ht, c, dd);
                if (CTL_PREFETCH_NODE) asm volatile("" :: "v"(pf_node));
                if (CTL_PARK_IN_STEP && FMT != kFmtF2 && n_hit && c[0] < 0 && pend < 0) {   // what the next iteration would do through the stack: same order of visits
                    pend = ~c[0]; c[0] = c[1]; c[1] = c[2]; c[2] = c[3]; dd[0] = dd[1]; dd[1] = dd[2]; dd[2] = dd[3]; n_hit--;
                    if (CTL_PREFETCH_LEAF) pf_leaf = *(const uint32_t*)(S.flat_leaves + (size_t)(uint32_t)pend * 8);
                }
                node = n_hit ? c[0] : popped.link;
                if (CTL_PREFETCH_NODE && (unsigned)node < (unsigned)kSentinel) pf_node = *(const uint32_t*)(nodes + (node & ~3));
                const int top = sp + n_hit - 1;    // n_hit == 0: one entry popped
                if (FMT == kFmtF2) {
                    if (n_hit == 2) st.set(top, c[1], dd[1], kWithDist);
                } else if (top < kFlatLdsRows) {   // common case: unconditional LDS stores, unused ones into the spare row
                    st.put_row(n_hit >= 2 ? top : kFlatLdsRows, c[1], dd[1], kWithDist);
                    st.put_row(n_hit >= 3 ? top - 1 : kFlatLdsRows, c[2], dd[2], kWithDist);
                    st.put_row(n_hit >= 4 ? top - 2 : kFlatLdsRows, c[3], dd[3], kWithDist);
                } else {
                    if (n_hit >= 4) st.set(top - 2, c[3], dd[3], kWithDist);
                    if (n_hit >= 3) st.set(top - 1, c[2], dd[2], kWithDist);
                    if (n_hit >= 2) st.set(top, c[1], dd[1], kWithDist);
                }
                sp = top;
                if (COUNT && sp > sp_max) sp_max = sp;
                if (kWithDist && n_hit == 0 && stack_entry_culled(popped, ht)) node = CTL_STACK_DIST == 3 ? kStaleLink : st.pop(sp, ht, kWithDist);   // the popped child lies behind the hit found since it was pushed: next one
            }
        }
        if (has_ray && !finished) finished = (node == kSentinel) && pend < 0;
        if (finished) {
            const uint32_t id = ray_id & 0x7fffffffu; const bool found = (ray_id >> 31) != 0u;
            if (ANY_HIT && occ) occ[id] = found ? 1u : 0u;
            if (hit && !found) { hit[id] = make_float4(ht, 0.0f, 0.0f, __int_as_float(-1)); hit_node[id] = -1; }   // a found hit wrote its record when it was accepted
            if (COUNT) { atomicAdd(&s_hist[sp_max < kStackSize ? sp_max : kStackSize - 1], 1u); sp_max = 0; }   // the workgroup's own histogram in LDS: one global atomic per ray on two dozen addresses made the counting kernels 15 x slower than the timed ones
            has_ray = false; node = kSentinel; pend = -1;
        }
    }
    if (COUNT) { __syncthreads(); for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) if (s_hist[i]) atomicAdd(&g_stack_hist[i], (unsigned long long)s_hist[i]); }
}

// Single-ray form for the megakernel plugin (one lane walks a whole path): same node steps, same entry test.  The stack's first kSingleLdsRows entries live in LDS
// (`lds_col` = this lane's column of a [row][256] array, as in the wavefront kernel), deeper ones in a private array: a stack in scratch alone made every push and
// pop a trip through the vector memory path (megakernel 64.5 -> see DESIGN.md §8 ms per pass on the bench workload).
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
    const int fmt = S.flat_format;
    ht = tmax; hu = hv = 0.0f; htri = -1; hnode = -1;
    while (node != kSentinel) {
        if (node >= 0) {
            int c[4]; float dd[4]; int n_hit;
            if (fmt == kFmtF4) n_hit = node_step_f4(nodes, node, R, tmin, ht, c, dd);
            else if (fmt == kFmtQ4) { node_words W; node_fetch_own(nodes, node, S.flat_compact != 0, W); n_hit = node_step_q4(W, node, R, o.x, o.y, o.z, d.x, d.y, d.z, tmin, ht, c, dd, S.flat_compact != 0); }
            else n_hit = node_step_f2(nodes, node, R, tmin, ht, c, dd);
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
