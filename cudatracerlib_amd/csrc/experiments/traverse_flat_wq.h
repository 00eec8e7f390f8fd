// experiments/traverse_flat_wq.h (-DCTL_LEAF_QUEUE=1; measured and NOT shipped, see the end of this comment) — the 4-wide flattened-BVH traversal (traverse_flat.h) with its entry tests run from a WAVE-WIDE QUEUE (round 4).
//
// Why.  In the parked-leaf kernel a lane that reaches a leaf keeps it in a register until enough lanes of the wave hold one (16 of 64), so an entry-test phase runs with
// 0.29 of its lanes on average — and an entry test is the expensive kind of wave iteration: seven 16-B loads per lane of a 128-B line that practically always comes from
// HBM, then ~150 VALU.  Measured on the 8-wide variant of this kernel (tools/r04_q8_knobs.sh): trading node-step utilisation 0.78 -> 0.60 for entry-test utilisation
// 0.30 -> 0.53 left the launch time unchanged, i.e. one entry-test iteration costs what 1.6 node iterations cost; a quarter of the 4-wide kernel's iterations are entry tests.
//
// How.  A lane that pops a leaf link no longer parks it: it appends {its lane number, the leaf's first entry} to the wave's queue in LDS (ballot + mbcnt, one ds_write) and
// walks on at once — never blocked by a second leaf.  When the queue holds kLeafQueueFlush items (or too few lanes have a node left to visit) the wave FLUSHES it: lane j takes
// item j, fetches the owner's ray with eight ds_bpermute, tests the entry (the chain of a multi-entry leaf in a loop: 1.2 % of the leaves) and, when the entry is a hit closer
// than the owner's best, competes with a 64-bit ds_min on the owner's {distance, item} word; the winner writes the hit record to the ray's slot of the hit arrays (the record
// already lived there, hit_in_memory), the owner picks the new distance up from its word.  A flush empties the queue, so "has items queued" is one flag per lane, and a ray is
// finished when its stack is empty and that flag is clear.  The reported hit is still decided entry by entry with the reference's arithmetic (flat_leaf_eval); among entries
// that hit at exactly the same t the item number decides instead of the visiting order — the tie the tests except already.
//
// Result (round 4, synthetic-SM, one box each; profiles/r04_structural_experiments.log).  Bit-exact (tests/test_gpu_intersect.py with this build), entry-test lane utilisation
// 0.29 -> 0.74-1.0, node steps 0.84 -> 0.86-0.89, 5 % FEWER VALU instructions than the parked-leaf kernel — and 14.3 -> 19.7-21.1 ms per fused launch: SQ_WAIT_ANY doubles
// (2.1e11 -> 4.4e11 quad-cycles).  First version (foreign ray + all seven entry loads in registers): 50 registers over the 72, 33.7 ms.  Flush out of line: 31.0.  Entry loads
// in two steps (this file): 21.1; flush threshold 64 / 48 / 32 / 24 / 16: 22.0 / 21.1 / 20.2 / 19.7 / 19.8 ms; touching an entry's line when it is queued: 25.0.  Reading: the
// parked-leaf kernel hides an entry test's HBM miss behind the other six waves' node steps; a flush is a chain of dependent waits (ray exchange, instance rows, Woop rows,
// ds_min, read-back) that the whole wave sits through, and lane utilisation was never what the time was made of.
#pragma once
#include "../traverse_flat.h"

namespace ctl {

#ifndef CTL_WQ_PREFETCH
#define CTL_WQ_PREFETCH 0
#endif
constexpr int kWqLdsRows = 17;                 // stack rows per lane in LDS (+ 1 spare): 18 KiB + the queues' 4 KiB = 22 KiB per 256-lane workgroup, seven workgroups per CU (0.11 % of the bench rays go deeper: scratch)
constexpr int kWqCapacity = 128;               // items per wave: a flush is due at g_leaf_batch_wq <= 64 items and one iteration appends at most 64
constexpr int kWqLdsInts = (kWqLdsRows + 1) * 256 + 4 * kWqCapacity + 2 * 256;   // stack | 4 queues | 256 best words (8 B each)
__device__ int g_wq_flush = 48;                // flush the queue at this many items (knob CTL_WQ_FLUSH)
__device__ int g_wq_min_inner = 20;            // ... or as soon as fewer lanes than this have a node to visit while items wait (knob CTL_WQ_MIN_INNER)

typedef __attribute__((address_space(3))) int wq_lds_int;
typedef __attribute__((address_space(3))) uint32_t wq_lds_u32;
typedef __attribute__((address_space(3))) unsigned long long wq_lds_u64;

struct wq_stack {
    wq_lds_int* lds; int ovf[kStackSize - kWqLdsRows];
    __device__ __forceinline__ int get(int i) const { int w = lds[(i < kWqLdsRows ? i : kWqLdsRows) * 256]; if (i >= kWqLdsRows) w = ovf[i - kWqLdsRows]; return w; }
    __device__ __forceinline__ void put_row(int row, int link) { lds[row * 256] = link; }
    __device__ __forceinline__ void set(int i, int link) { if (i < kWqLdsRows) put_row(i, link); else ovf[i - kWqLdsRows] = link; }
};
// a candidate hit of the entry a lane tests for another lane's ray
struct hit_candidate {
    float ht, hu, hv; int htri, hnode; bool any;
    __device__ __forceinline__ float dist() const { return ht; }
    __device__ __forceinline__ void accept(float t, float u, float v, int tri, int nd, uint32_t) { ht = t; hu = u; hv = v; htri = tri; hnode = nd; any = true; }
};

// One leaf entry for ANOTHER lane's ray, in two steps: the instance rows first (48 B), the ray taken into object space, only then the Woop rows (the same 128-B line, by then
// on its way or in the L1).  All seven loads at once — as the parked-leaf kernel issues them, one wait — need 28 registers next to the foreign ray, the candidate and the
// lane's own traversal state: 80 where the kernel has 72, and the allocator's spills landed in the node step (33.7 instead of 14.2 ms per launch).  The second step costs an
// L1-hit latency per flush, i.e. per ~7 iterations.
template <bool ANY_HIT, bool ALPHA, class SINK>
__device__ __forceinline__ int wq_leaf_test(const dev_scene& S, uint32_t e, float orgx, float orgy, float orgz, float dirx, float diry, float dirz, float tmin, SINK& sink, bool& got) {
    const float4* __restrict__ p = S.flat_leaves + (size_t)e * 8;
    m34 m;
    { const float4 r0 = p[4], r1 = p[5], r2 = p[6];
      m.r[0][0] = r0.x; m.r[0][1] = r0.y; m.r[0][2] = r0.z; m.r[0][3] = r0.w; m.r[1][0] = r1.x; m.r[1][1] = r1.y; m.r[1][2] = r1.z; m.r[1][3] = r1.w;
      m.r[2][0] = r2.x; m.r[2][1] = r2.y; m.r[2][2] = r2.z; m.r[2][3] = r2.w; }
    const f3 d = xform_dir(m, f3(dirx, diry, dirz));
    f3 o = xform_point(m, f3(orgx, orgy, orgz));
    if (!S.inst_w_one) { const float w33 = p[7].x; o = f3(o.x / w33, o.y / w33, o.z / w33); }
    asm volatile("" : "+v"(p) : "v"(o.x), "v"(d.x));   // the Woop rows are fetched AFTER the transform (see above)
    const float4 v00 = p[0], v11 = p[1], v22 = p[2]; const uint2 iw = *(const uint2*)(p + 3);
    if (flat_woop_test<ALPHA>(S, v00, v11, v22, iw.x, (int)iw.y, o, d, tmin, sink)) { got = true; if (ANY_HIT) return -1; }
    return (iw.x & 1u) ? -1 : (int)(e + 1);
}

template <bool ANY_HIT, bool COUNT, bool ALPHA>
__device__ __forceinline__ void intersect_flat_wq(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                                  float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_ints, trav_counts& cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned int s_hist[COUNT ? kStackSize : 1];
    if (COUNT) { for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) s_hist[i] = 0u; __syncthreads(); }
    const int refill_idle = g_refill_idle, q_flush = g_wq_flush, q_min_inner = g_wq_min_inner;
    const bool compact = S.flat_compact != 0;
    wq_stack st; st.lds = (wq_lds_int*)lds_ints + threadIdx.x;
    wq_lds_u32* queue = (wq_lds_u32*)lds_ints + (kWqLdsRows + 1) * 256 + wave * kWqCapacity;
    wq_lds_u64* best = (wq_lds_u64*)((wq_lds_u32*)lds_ints + (kWqLdsRows + 1) * 256 + 4 * kWqCapacity) + wave * 64;   // per lane: {distance of the closest hit so far (bits), item that found it}
    bool has_ray = false, queued = false;
    uint32_t ray_id = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, tmin = 0;
    ray_cull R{ 0, 0, 0, 0, 0, 0 };
    float ht = 0;
    int sp = 0, node = kSentinel, sp_max = 0;
    uint32_t qcount = 0;                          // items in the wave's queue (uniform)
    int pf_entry = -1; uint32_t pf_word = 0u;     // CTL_WQ_PREFETCH: the entry a lane queued in this iteration; one dword of its line is touched behind the node loads, so that the flush finds the line on its way
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
                    ray_id = my; has_ray = true; queued = false;
                    ox = o.x; oy = o.y; oz = o.z; tmin = o.w; dx = d.x; dy = d.y; dz = d.z;
                    R.idx = rcp_cull(dx); R.idy = rcp_cull(dy); R.idz = rcp_cull(dz);
                    R.oox = ox * R.idx; R.ooy = oy * R.idy; R.ooz = oz * R.idz;
                    ht = d.w;
                    best[lane] = ((unsigned long long)__float_as_uint(ht) << 32) | 0xffffffffull;
                    sp = 0; st.put_row(0, kSentinel); node = S.flat_root;
                }
                chunk_next += want < avail ? want : avail;
            }
        }
        if (__ballot(has_ray) == 0ull) { if (exhausted) break; continue; }

        // ---- a lane standing on a leaf hands it to the wave's queue and goes on with its next stack entry
        {
            const bool on_leaf = has_ray && node < 0;
            const unsigned long long mq = __ballot(on_leaf);
            if (mq != 0ull) {
                const uint32_t pos = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0));
                if (on_leaf) { queue[pos] = ((uint32_t)lane << 26) | (uint32_t)~node; queued = true; pf_entry = ~node; node = st.get(sp); sp--; }
                qcount += (uint32_t)__popcll(mq);
            }
        }
        const bool at_inner = has_ray && (unsigned)node < (unsigned)kSentinel;
        const unsigned long long m_inner = __ballot(at_inner);
        if (qcount != 0u && ((int)qcount >= q_flush || __popcll(m_inner) < q_min_inner)) {
            // ---- flush: lane j tests item j for the lane that queued it
            if (CTL_WQ_PREFETCH) asm volatile("" :: "v"(pf_word));
            for (uint32_t base = 0; base < qcount; base += 64u) {
                const bool act = base + (uint32_t)lane < qcount;
                const uint32_t item = act ? queue[base + lane] : ((uint32_t)lane << 26);
                const int owner = (int)(item >> 26);
                const float fox = __shfl(ox, owner, 64), foy = __shfl(oy, owner, 64), foz = __shfl(oz, owner, 64);
                const float fdx = __shfl(dx, owner, 64), fdy = __shfl(dy, owner, 64), fdz = __shfl(dz, owner, 64), ftmin = __shfl(tmin, owner, 64);
                const uint32_t frid = (uint32_t)__shfl((int)ray_id, owner, 64) & 0x7fffffffu;
                hit_candidate C{ __uint_as_float((uint32_t)(best[owner] >> 32)), 0.0f, 0.0f, -1, -1, false };
                if (act) {
                    if (COUNT && lane == 0) cnt.w_tri++;
                    int e = (int)(item & 0x03ffffffu);
                    do {   // the entries of the leaf (one, for 98.8 % of the leaves)
                        if (COUNT) cnt.n_tri++;
                        bool got = false;
                        e = wq_leaf_test<ANY_HIT, ALPHA>(S, (uint32_t)e, fox, foy, foz, fdx, fdy, fdz, ftmin, C, got);
                    } while (e >= 0);
                }
                const unsigned long long key = ((unsigned long long)__float_as_uint(C.ht) << 32) | (unsigned long long)(base + (uint32_t)lane);
                if (act && C.any) (void)__hip_atomic_fetch_min(&best[owner], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_min_u64
                if (act && C.any && best[owner] == key && hit) { hit[frid] = make_float4(C.ht, C.hu, C.hv, __int_as_float(C.htri)); hit_node[frid] = C.hnode; }   // the closest candidate of this round: the record goes to the ray's slot
            }
            // the lane's own culling terms and hit distance are rebuilt rather than kept alive across the flush
            R.idx = rcp_cull(dx); R.idy = rcp_cull(dy); R.idz = rcp_cull(dz);
            R.oox = ox * R.idx; R.ooy = oy * R.idy; R.ooz = oz * R.idz;
            if (has_ray) {
                const uint32_t nb = (uint32_t)(best[lane] >> 32);
                if (nb != __float_as_uint(ht)) ray_id |= 0x80000000u;
                ht = __uint_as_float(nb);
            }
            qcount = 0u; queued = false;
        } else {
            // ---- node phase
            if (at_inner) {
                if (COUNT) { cnt.n_inner++; if (lane == (int)__builtin_ctzll(m_inner)) cnt.w_inner++; }
                node_words W; node_fetch_own(nodes, node, compact, W);
#if CTL_WQ_PREFETCH
                if (pf_entry >= 0) { pf_word = __builtin_nontemporal_load((const uint32_t*)(S.flat_leaves + (size_t)(uint32_t)pf_entry * 8 + 4)); pf_entry = -1; }   // issued AFTER the node loads: the step's wait does not include it
#endif
                const int popped = st.get(sp);   // issued early: used when no child is entered
                int c[4]; float dd[4];
                const int n_hit = node_step_q4(W, node, R, ox, oy, oz, dx, dy, dz, tmin, ht, c, dd, compact);
                node = n_hit ? c[0] : popped;
                const int top = sp + n_hit - 1;    // n_hit == 0: one entry popped
                if (top < kWqLdsRows) {            // common case: unconditional LDS stores, unused ones into the spare row
                    st.put_row(n_hit >= 2 ? top : kWqLdsRows, c[1]);
                    st.put_row(n_hit >= 3 ? top - 1 : kWqLdsRows, c[2]);
                    st.put_row(n_hit >= 4 ? top - 2 : kWqLdsRows, c[3]);
                } else {
                    if (n_hit >= 4) st.set(top - 2, c[3]);
                    if (n_hit >= 3) st.set(top - 1, c[2]);
                    if (n_hit >= 2) st.set(top, c[1]);
                }
                sp = top;
                if (COUNT && sp > sp_max) sp_max = sp;
            }
        }
        const bool found = (ray_id >> 31) != 0u;
        const bool finished = has_ray && !queued && (node == kSentinel || (ANY_HIT && found));
        if (finished) {
            const uint32_t id = ray_id & 0x7fffffffu;
            if (ANY_HIT && occ) occ[id] = found ? 1u : 0u;
            if (hit && !found) { hit[id] = make_float4(ht, 0.0f, 0.0f, __int_as_float(-1)); hit_node[id] = -1; }   // a found hit's record was written by the lane that tested it
            if (COUNT) { atomicAdd(&s_hist[sp_max < kStackSize ? sp_max : kStackSize - 1], 1u); sp_max = 0; }
            has_ray = false; node = kSentinel;
        }
    }
    if (COUNT) { __syncthreads(); for (int i = threadIdx.x; i < kStackSize; i += blockDim.x) if (s_hist[i]) atomicAdd(&g_stack_hist[i], (unsigned long long)s_hist[i]); }
}

} // namespace ctl
