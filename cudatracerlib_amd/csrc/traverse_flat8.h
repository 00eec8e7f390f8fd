// traverse_flat8.h — traversal of the 8-wide flattened BVH ("Q8", flat8.h) by persistent wave64 waves, one ray per lane.
//
// What is reported is the reference's, exactly as in traverse_flat.h: every leaf entry a ray looks at is decided by the two-level arithmetic of intersectKernel
// (Kernel/TraceHelper.cu:526-560 ray into the instance's object space, :646-682 Woop test with an exact division; flat_leaf_test), the tree only culls.
//
// Execution model.  The skeleton is the 4-wide kernel's — lane refill, node steps and entry tests as separate wave-wide phases, a lane PARKS what it has to test and keeps
// descending — with the step re-cut around the node format so that nothing depends on the order of the children:
//  * a node step tests eight boxes (+ the node's oriented slab as a fourth axis, flat_slab.h) and ends with ONE byte of hit slots; its sign bits are collected with one
//    v_alignbit per child, the byte is brought into the ray's visiting order by three conditional bit swaps (slots are octant-ordered at build time), and the next child is
//    the highest set bit: no distances kept, no ordering network;
//  * the stack holds one 8-byte SIBLING GROUP per level — {first inner child | B flags, inner mask | hit byte} — instead of up to three links per step: at most one LDS store
//    per step (none unless both the entered node and the level above still have children to visit), 10 rows + 1 spare row of 8 B = 22 KiB per 256-lane workgroup, seven
//    workgroups per CU as before;
//  * the leaf slots a step hits are parked as one group {first entry of the node, leaf mask | hit byte} and tested one entry per leaf phase (a slot is one triangle:
//    entry = first + rank of the slot among the leaf slots).  A parent's B flag says which inner children can produce leaf hits (and carry a slab: their sixth 16 B are
//    loaded with the other five): a lane that still holds a parked group waits in front of such a node instead of needing a second parking place.
#pragma once
#include "traverse_flat.h"
#include "flat8.h"

namespace ctl {

constexpr int kQ8LdsRows = 10;        // sibling groups per lane in LDS (+ 1 spare row); deeper ones in scratch.  The bench tree is 12 levels deep and a level only takes a row while siblings wait there
constexpr int kQ8StackSize = kFlat8StackGroups;   // checked at upload against the tree's depth (tracer.hip)
constexpr uint32_t kQ8None = 0xffffffffu;
typedef __attribute__((address_space(3))) unsigned long long q8_lds_word;

struct q8_group { uint32_t base_b, mh; };   // mh: inner mask (bits 0..7) | slots still to visit, in visiting order (bits 8..15); bits 8..15 == 0: nothing left
__device__ __forceinline__ unsigned long long q8_pack(const q8_group g) { return (unsigned long long)g.base_b | ((unsigned long long)g.mh << 32); }
__device__ __forceinline__ q8_group q8_unpack(unsigned long long w) { return q8_group{ (uint32_t)w, (uint32_t)(w >> 32) }; }

struct q8_stack {
    q8_lds_word* lds;                                        // this lane's column, stride 256 words
    unsigned long long ovf[kQ8StackSize - kQ8LdsRows];
    __device__ __forceinline__ q8_group get(int i) const {
        unsigned long long w = lds[(i < kQ8LdsRows ? i : kQ8LdsRows) * 256];   // a ds_read whatever the depth (the spare row when the entry lives in scratch) ...
        if (i >= kQ8LdsRows) w = ovf[i - kQ8LdsRows];                            // ... and the rare deep entry from scratch
        return q8_unpack(w);
    }
    __device__ __forceinline__ void put_row(int row, const q8_group g) { lds[row * 256] = q8_pack(g); }
    __device__ __forceinline__ void set(int i, const q8_group g) { if (i < kQ8LdsRows) put_row(i, g); else ovf[i - kQ8LdsRows] = q8_pack(g); }
};

// the next child of a sibling group in the ray's visiting order: returns its link (node index << 1 | B flag) and takes it out of the group; kQ8None when the group is empty
__device__ __forceinline__ uint32_t q8_pick(q8_group& g, uint32_t octinv) {
    const uint32_t left = g.mh >> 8;
    if (left == 0u) return kQ8None;
    const uint32_t bit = 31u - (uint32_t)__builtin_clz(left);
    const uint32_t slot = bit ^ octinv;
    g.mh &= ~(0x100u << bit);
    return (flat8_child_node(g.base_b, g.mh & 0xffu, slot) << 1) | ((g.base_b >> (24u + slot)) & 1u);
}

// One node step: the eight child boxes (and the slab) of node `link` against the ray.  Returns the hit slots in VISITING order: inner children in bits 0..7, leaf slots in bits 8..15.
struct q8_words { uint4 q0, q1, q2, q3, q4, q5; };
__device__ __forceinline__ void q8_fetch(const float4* __restrict__ nodes, uint32_t link, q8_words& W) {
    const uint4* __restrict__ p = (const uint4*)(nodes + (size_t)(link >> 1) * 8);
    W.q0 = p[0]; W.q1 = p[1]; W.q2 = p[2]; W.q3 = p[3]; W.q4 = p[4];
    if (link & 1u) W.q5 = p[5];
}
__device__ __forceinline__ float q8_byte(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyte<k>
__device__ __forceinline__ uint32_t q8_step(const q8_words& W, uint32_t link, const ray_cull& R, float ox, float oy, float oz, float dx, float dy, float dz, float tmin, float ht, uint32_t octinv) {
    const uint32_t meta = W.q0.w;
    const float inf = __builtin_huge_valf();
    float s_alpha = 0.0f, s_bn = -inf, s_bf = inf; uint32_t s_n0 = 0u, s_n1 = 0u, s_f0 = 0u, s_f1 = 0u;   // no slab: [-inf, inf] for every child
    if ((link & 1u) && W.q1.z != 0u) {
        slab_ray SR;
        slab_setup(W.q1.z, __uint_as_float(W.q1.w), 0u, 0u, __uint_as_float(W.q0.x), __uint_as_float(W.q0.y), __uint_as_float(W.q0.z), ox, oy, oz, dx, dy, dz, SR);
        const bool neg = SR.neg;
        s_alpha = SR.alpha; s_bn = SR.beta_n; s_bf = SR.beta_f;
        s_n0 = neg ? W.q5.z : W.q5.x; s_n1 = neg ? W.q5.w : W.q5.y; s_f0 = neg ? W.q5.x : W.q5.z; s_f1 = neg ? W.q5.y : W.q5.w;
    }
    const float ax = __uint_as_float((meta & 0xffu) << 23) * R.idx, ay = __uint_as_float(((meta >> 8) & 0xffu) << 23) * R.idy, az = __uint_as_float(((meta >> 16) & 0xffu) << 23) * R.idz;
    const float bx = __builtin_fmaf(__uint_as_float(W.q0.x), R.idx, -R.oox), by = __builtin_fmaf(__uint_as_float(W.q0.y), R.idy, -R.ooy), bz = __builtin_fmaf(__uint_as_float(W.q0.z), R.idz, -R.ooz);
    const bool negx = R.idx < 0.0f, negy = R.idy < 0.0f, negz = R.idz < 0.0f;
    // qlo_x = q2.xy, qlo_y = q2.zw, qlo_z = q3.xy, qhi_x = q3.zw, qhi_y = q4.xy, qhi_z = q4.zw
    const uint32_t nx[2] = { negx ? W.q3.z : W.q2.x, negx ? W.q3.w : W.q2.y }, fx[2] = { negx ? W.q2.x : W.q3.z, negx ? W.q2.y : W.q3.w };
    const uint32_t ny[2] = { negy ? W.q4.x : W.q2.z, negy ? W.q4.y : W.q2.w }, fy[2] = { negy ? W.q2.z : W.q4.x, negy ? W.q2.w : W.q4.y };
    const uint32_t nz[2] = { negz ? W.q4.z : W.q3.x, negz ? W.q4.w : W.q3.y }, fz[2] = { negz ? W.q3.x : W.q4.z, negz ? W.q3.y : W.q4.w };
    const uint32_t sn[2] = { s_n0, s_n1 }, sf[2] = { s_f0, s_f1 };
    uint32_t miss = 0u;   // bit s: slot s is NOT entered.  Slot 7 first, so that slot s ends up in bit s
#pragma unroll
    for (int s = 7; s >= 0; s--) {
        const int w = s >> 2, k = s & 3;
        const float tnx = __builtin_fmaf(q8_byte(nx[w], k), ax, bx), tfx = __builtin_fmaf(q8_byte(fx[w], k), ax, bx);
        const float tny = __builtin_fmaf(q8_byte(ny[w], k), ay, by), tfy = __builtin_fmaf(q8_byte(fy[w], k), ay, by);
        const float tnz = __builtin_fmaf(q8_byte(nz[w], k), az, bz), tfz = __builtin_fmaf(q8_byte(fz[w], k), az, bz);
        const float tns = __builtin_fmaf(q8_byte(sn[w], k), s_alpha, s_bn), tfs = __builtin_fmaf(q8_byte(sf[w], k), s_alpha, s_bf);
        const float cmin = max3_raw(max3_raw(tnx, tny, tnz), tns, tmin);
        const float cmax = min3_raw(min3_raw(tfx, tfy, tfz), tfs, ht);
        // entered <=> cmax >= cmin.  The sign bit of cmax - cmin says so (x - x = +0; an empty slot's box is inverted, flatten.cpp); one v_alignbit shifts it into the byte
        miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(cmax - cmin), 31);
    }
    const uint32_t hit8 = ~miss & 0xffu;
    const uint32_t imask = meta >> 24, lmask = (W.q1.x >> 24) & ~imask;
    uint32_t both = (hit8 & imask) | ((hit8 & lmask) << 8);
    // into the ray's visiting order (flat8_to_order on both bytes at once)
    if (octinv & 1u) both = ((both & 0x5555u) << 1) | ((both >> 1) & 0x5555u);
    if (octinv & 2u) both = ((both & 0x3333u) << 2) | ((both >> 2) & 0x3333u);
    if (octinv & 4u) both = ((both & 0x0f0fu) << 4) | ((both >> 4) & 0x0f0fu);
    return both;
}

// The whole intersect kernel body over the 8-wide structure: `n` rays (ro, rd) -> hit / hit_node (closest) and/or occ (any-hit flag).
template <bool ANY_HIT, bool COUNT, bool ALPHA>
__device__ __forceinline__ void intersect_flat8(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                                float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, unsigned long long* lds_stack_words, trav_counts& cnt) {
    const int lane = threadIdx.x & 63;
    __shared__ unsigned int s_hist[COUNT ? kStackSize : 1];   // counting kernels: stack-depth histogram of this workgroup's rays, added to g_stack_hist at the end
    if (COUNT) { for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) s_hist[i] = 0u; __syncthreads(); }
    const int refill_idle = g_refill_idle, leaf_batch = g_leaf_batch;
    q8_stack st; st.lds = (q8_lds_word*)lds_stack_words + threadIdx.x;
    bool has_ray = false;
    uint32_t ray_id = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, tmin = 0;
    ray_cull R{ 0, 0, 0, 0, 0, 0 };
    float ht = 0;                                 // distance of the closest hit so far; its record is in hit[] / hit_node[] already (hit_in_memory), bit 31 of ray_id says there is one
    hit_in_memory sink{ ht, ray_id, hit, hit_node, ANY_HIT ? nullptr : S.hit_key_out };
    uint32_t node = kQ8None;                      // the node the lane visits next (index << 1 | B flag)
    q8_group grp{ 0u, 0u };                       // what is left of the sibling group `node` came from
    uint32_t p_base = 0u, p_mh = 0u;              // parked leaf group: first entry of its node, leaf mask | slots still to test (visiting order) << 8
    uint32_t octinv = 0u;
    int sp = 0, sp_max = 0;
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
                    octinv = (R.idx < 0.0f ? 0u : 1u) | (R.idy < 0.0f ? 0u : 2u) | (R.idz < 0.0f ? 0u : 4u);   // = flat8_octinv of the guarded direction (what the box tests use)
                    ht = d.w;
                    sp = 0; st.put_row(0, q8_group{ 0u, 0u }); grp = q8_group{ 0u, 0u }; node = (uint32_t)S.flat_root; p_mh = 0u;
                }
                chunk_next += want < avail ? want : avail;
            }
        }
        if (__ballot(has_ray) == 0ull) { if (exhausted) break; continue; }

        const bool parked = (p_mh >> 8) != 0u;
        const bool at_inner = has_ray && node != kQ8None && !(parked && (node & 1u));   // a lane with a parked group waits in front of a node that can hand it another one
        const bool at_leaf = has_ray && parked;
        const unsigned long long m_inner = __ballot(at_inner), m_leaf = __ballot(at_leaf);
        bool finished = false;
        if (m_leaf != 0ull && (__popcll(m_leaf) >= leaf_batch || m_inner == 0ull)) {
            // ---- leaf phase: every lane that holds a parked group tests its next entry
            if (at_leaf) {
                if (COUNT) { cnt.n_tri++; if (lane == (int)__builtin_ctzll(m_leaf)) cnt.w_tri++; }
                const uint32_t left = p_mh >> 8;
                const uint32_t bit = 31u - (uint32_t)__builtin_clz(left), slot = bit ^ octinv;
                p_mh &= ~(0x100u << bit);
                const uint32_t e = flat8_leaf_entry(p_base, p_mh & 0xffu, slot);
                bool got = false;
                (void)flat_leaf_test<ANY_HIT, ALPHA>(S, e, ox, oy, oz, dx, dy, dz, tmin, sink, got);
                if (ANY_HIT && got) finished = true;
            }
        } else {
            // ---- node phase
            if (at_inner) {
                if (COUNT) { cnt.n_inner++; if (lane == (int)__builtin_ctzll(m_inner)) cnt.w_inner++; }
                q8_words W; q8_fetch(nodes, node, W);
                const q8_group popped = st.get(sp);   // issued early: used when neither the node nor the group above it has anything left
                const uint32_t both = q8_step(W, node, R, ox, oy, oz, dx, dy, dz, tmin, ht, octinv);
                const uint32_t lmask = (W.q1.x >> 24) & ~(W.q0.w >> 24);
                if (both >> 8) { p_base = W.q1.y; p_mh = lmask | (both & 0xff00u); }   // the lane's parking place is free here: B of the link said so
                const bool enter = (both & 0xffu) != 0u, above = (grp.mh >> 8) != 0u;
                if (enter) {
                    const int row = (above && sp + 1 < kQ8LdsRows) ? sp + 1 : kQ8LdsRows;   // unconditional LDS store; unused ones (and those of deep entries) land in the spare row
                    st.put_row(row, grp);
                    if (above) { sp++; if (sp >= kQ8LdsRows) st.set(sp, grp); }
                    grp = q8_group{ W.q1.x, (W.q0.w >> 24) | ((both & 0xffu) << 8) };
                } else if (!above) { grp = popped; sp = sp > 0 ? sp - 1 : 0; }
                if (COUNT && sp > sp_max) sp_max = sp;
                node = q8_pick(grp, octinv);
            }
        }
        if (has_ray && !finished) finished = node == kQ8None && (p_mh >> 8) == 0u;
        if (finished) {
            const uint32_t id = ray_id & 0x7fffffffu; const bool found = (ray_id >> 31) != 0u;
            if (ANY_HIT && occ) occ[id] = found ? 1u : 0u;
            if (hit && !found) { hit[id] = make_float4(ht, 0.0f, 0.0f, __int_as_float(-1)); hit_node[id] = -1; if (!ANY_HIT && S.hit_key_out) S.hit_key_out[id] = 0; }   // a found hit wrote its record when it was accepted
            if (COUNT) { atomicAdd(&s_hist[sp_max < kStackSize ? sp_max : kStackSize - 1], 1u); sp_max = 0; }
            has_ray = false; node = kQ8None; p_mh = 0u;
        }
    }
    if (COUNT) { __syncthreads(); for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) if (s_hist[i]) atomicAdd(&g_stack_hist[i], (unsigned long long)s_hist[i]); }
}

// Single-ray form for the megakernel plugin (one lane walks a whole path): same node step, leaf slots tested at once.  The first groups of the stack live in LDS
// (`lds_col` = this lane's column of an int [row][256] array: two rows per group), deeper ones in a private array.
constexpr int kQ8SingleLdsGroups = kSingleLdsRows / 2;
struct q8_single_stack {
    lds_int* lds; unsigned long long ovf[kQ8StackSize - kQ8SingleLdsGroups];
    __device__ __forceinline__ q8_group get(int i) const {
        const int r = i < kQ8SingleLdsGroups ? i : 0;
        q8_group g{ (uint32_t)lds[(2 * r) * 256], (uint32_t)lds[(2 * r + 1) * 256] };
        if (i >= kQ8SingleLdsGroups) g = q8_unpack(ovf[i - kQ8SingleLdsGroups]);
        return g;
    }
    __device__ __forceinline__ void set(int i, const q8_group g) { if (i < kQ8SingleLdsGroups) { lds[(2 * i) * 256] = (int)g.base_b; lds[(2 * i + 1) * 256] = (int)g.mh; } else ovf[i - kQ8SingleLdsGroups] = q8_pack(g); }
};
template <bool ANY_HIT, bool ALPHA_DYNAMIC>
__device__ bool trace_single_flat8(const dev_scene& S, lds_int* lds_col, f3 o, f3 d, float tmin, float tmax, float& ht, float& hu, float& hv, int& htri, int& hnode) {
    const float4* __restrict__ nodes = S.flat_nodes;
    ray_cull R;
    R.idx = rcp_cull(d.x); R.idy = rcp_cull(d.y); R.idz = rcp_cull(d.z);
    R.oox = o.x * R.idx; R.ooy = o.y * R.idy; R.ooz = o.z * R.idz;
    const uint32_t octinv = (R.idx < 0.0f ? 0u : 1u) | (R.idy < 0.0f ? 0u : 2u) | (R.idz < 0.0f ? 0u : 4u);
    q8_single_stack stack; stack.lds = lds_col; int sp = 0;
    q8_group grp{ 0u, 0u };
    uint32_t node = (uint32_t)S.flat_root;
    ht = tmax; hu = hv = 0.0f; htri = -1; hnode = -1;
    hit_in_regs sink{ ht, hu, hv, htri, hnode };
    while (node != kQ8None) {
        q8_words W; q8_fetch(nodes, node, W);
        const uint32_t both = q8_step(W, node, R, o.x, o.y, o.z, d.x, d.y, d.z, tmin, ht, octinv);
        const uint32_t lmask = (W.q1.x >> 24) & ~(W.q0.w >> 24);
        for (uint32_t left = both >> 8; left != 0u;) {
            const uint32_t bit = 31u - (uint32_t)__builtin_clz(left), slot = bit ^ octinv; left &= ~(1u << bit);
            const uint32_t e = flat8_leaf_entry(W.q1.y, lmask, slot);
            bool got = false;
            // USE_ALPHA of __traceRay_internal__ (TraceHelper.cu:135-153): scenes with alpha maps test every candidate hit
            if (ALPHA_DYNAMIC && S.alpha_maps) (void)flat_leaf_test<ANY_HIT, true>(S, e, o.x, o.y, o.z, d.x, d.y, d.z, tmin, sink, got);
            else (void)flat_leaf_test<ANY_HIT, false>(S, e, o.x, o.y, o.z, d.x, d.y, d.z, tmin, sink, got);
            if (ANY_HIT && got) return true;
        }
        const bool enter = (both & 0xffu) != 0u, above = (grp.mh >> 8) != 0u;
        if (enter) { if (above) stack.set(++sp, grp); grp = q8_group{ W.q1.x, (W.q0.w >> 24) | ((both & 0xffu) << 8) }; }
        else if (!above) { if (sp == 0) break; grp = stack.get(sp); sp--; }
        node = q8_pick(grp, octinv);
    }
    return htri >= 0;
}

} // namespace ctl
