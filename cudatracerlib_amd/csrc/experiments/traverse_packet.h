// traverse_packet.h — wave-uniform ("packet") closest-hit traversal of the flattened Q4 tree for COHERENT rays: the first bounce, whose rays k_raygen writes in blocks of 64 = one
// 8 x 8 pixel block of one pass.  All 64 lanes of a wave walk ONE traversal — the union of what its rays would visit — and every lane tests every node and every leaf entry of it
// against its own ray.  What that buys on a machine that is short of issue slots (round 6, tools/packet_union_probe.py: the oracle's traversal with a visit log, synthetic-SM 1080p):
//   one ray per lane:  33.2 node steps + 4.1 entry tests per ray, at 0.78 / 0.34 of the lanes busy, 253 VALU per node step (box tests 88, oriented slab 50, per-lane links 21,
//                      ordering network 25, per-lane stack and loop ~70) -> ~12 500 VALU wave-instructions per 64 rays
//   packet:            47.2 nodes + 16.3 entries per block of 64 (1.42 x / 3.98 x one ray's), every lane busy, ~90 VALU per node step — the links, the order, the stack and the
//                      loop are wave-uniform, i.e. scalar; no slab — -> ~6 500
// The hits are the reference's: every leaf entry is evaluated by flat_leaf_eval (traverse_flat.h), the tree only decides which entries a ray looks at; only the ORDER in which
// entries are looked at differs from the one-ray-per-lane kernel (rays that hit two triangles at exactly the same distance may report either, as between any two traversals).
#pragma once
#include "traverse_flat.h"

namespace ctl {

constexpr int kPacketStack = kStackSize;   // wave-uniform stack entries (LDS, one row per wave); the per-ray bound 3 * depth + 2 holds for the union as well: at most three children are pushed per level

// entry distance of this lane's ray into each of the four children of a Q4 node, inf = not entered: node_step_q4's box arithmetic alone (no slab, no links, no order)
__device__ __forceinline__ void packet_child_tests(const float4 q0, const float4 q1, const float4 q2, const ray_cull& R, float tmin, float ht, float dd[4]) {
    const uint32_t meta = __float_as_uint(q0.w);
    const float inf = __builtin_huge_valf();
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
        const float cmin = max_raw(max3_raw(tnx, tny, tnz), tmin);
        const float cmax = min_raw(min3_raw(tfx, tfy, tfz), ht);
        dd[k] = (cmax >= cmin) ? cmin : inf;   // a missing child's box is inverted (flatten.cpp): never entered
    }
}

__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// n rays (ro, rd) in blocks of 64 coherent rays -> hit / hit_node / key (closest hits).  `wstack`: this wave's row of kPacketStack ints in LDS.
__device__ __forceinline__ void intersect_packet(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                                 float4* __restrict__ hit, int* __restrict__ hit_node, volatile flat_stack_lds_word* wstack) {
    const int lane = threadIdx.x & 63;
    const float4* __restrict__ nodes = S.flat_nodes;
    const float inf = __builtin_huge_valf();
    ray_claims rc; rc.init(n);
    for (;;) {
        if (rc.next >= rc.end) { if (rc.exhausted || !rc.refill(n, work, lane)) break; }
        const uint32_t base = rc.next; rc.next += 64u;       // claims are whole blocks of 64 (guided_chunk), the queue's last block may be short
        const uint32_t my = base + (uint32_t)lane;
        const bool valid = my < rc.end;
        float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1, tmin = inf, ht = 0;   // a lane without a ray enters nothing and accepts nothing
        if (valid) { const float4 o = ro[my], d = rd[my]; ox = o.x; oy = o.y; oz = o.z; tmin = o.w; dx = d.x; dy = d.y; dz = d.z; ht = d.w; }
        ray_cull R; R.idx = rcp_cull(dx); R.idy = rcp_cull(dy); R.idz = rcp_cull(dz); R.oox = ox * R.idx; R.ooy = oy * R.idy; R.ooz = oz * R.idz;
        uint32_t ray_word = my;
        hit_in_memory sink{ ht, ray_word, hit, hit_node, S.hit_key_out };
        const int leader = (int)__builtin_ctzll(__ballot(valid));     // the lane whose entry distances order the children (block-coherent rays: any lane's order is a good one)
        // the wave-uniform stack lives in two VGPRs: entry i in lane i & 63 of stk[i >> 6] (v_writelane / v_readlane with a scalar lane index: no memory, no wait)
        int stk0 = kSentinel, stk1 = kSentinel;
        auto push = [&](int i, int v) {   // (no clang builtin for v_writelane_b32 in this toolchain)
            // (gfx9 VALU instructions read one scalar operand over the constant bus: the lane select goes through M0)
            if (i < 64) asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(stk0) : "s"(v), "s"(i) : "m0"); else asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(stk1) : "s"(v), "s"(i - 64) : "m0");
        };
        auto top = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(stk0, i) : __builtin_amdgcn_readlane(stk1, i - 64); };
        int sp = 0, node = uniform_int(S.flat_root);
        float4 wq0 = make_float4(0, 0, 0, 0), wq1 = wq0, wq2 = wq0; bool have_w = false;   // node words fetched ahead
        while (node != kSentinel) {
            if (node >= 0) {
                // (a uniform traversal is ONE chain of dependent fetches per wave where the one-ray-per-lane kernel has 64 in flight: the node behind a leaf is fetched while the leaf is tested)
                if (!have_w) { const float4* __restrict__ p = nodes + (node & ~3); wq0 = p[0]; wq1 = p[1]; wq2 = p[2]; }
                have_w = false;
                const float4 q0 = wq0, q1 = wq1, q2 = wq2;
                float dd[4]; packet_child_tests(q0, q1, q2, R, tmin, ht, dd);
                // wave-uniform from here: which children does ANY ray enter, their links (implied by the layout, flatten.h), the leader's order
                const uint32_t meta = uniform_u32(__float_as_uint(q0.w)), w0 = uniform_u32(__float_as_uint(q2.z)), w1 = uniform_u32(__float_as_uint(q2.w));
                const uint32_t ib4 = w0 & 0x03fffffcu, nlb15 = (0xffffffffu << 26) | (w1 >> 6);   // first inner child * 4; ~(first entry + 15)  (alignbit(0xffffffff, w1, 6))
                const uint32_t dl = nlb15 - ib4;
                const uint32_t t1 = (w0 >> 26) & 15u, t2 = (w1 >> 2) & 15u, t3 = ((w1 << 2) | (w0 >> 30)) & 15u;
                int c[4];
                c[0] = (int)(((meta >> 28) & 1u) ? nlb15 + 15u : (w0 & 0x03ffffffu));
                c[1] = (int)(ib4 + t1 + (((meta >> 29) & 1u) ? dl : 0u));
                c[2] = (int)(ib4 + t2 + (((meta >> 30) & 1u) ? dl : 0u));
                c[3] = (int)(ib4 + t3 + ((meta >> 31) ? dl : 0u));
                float key[4]; bool in[4]; int n_in = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    in[k] = __ballot(dd[k] < inf) != 0ull;
                    key[k] = in[k] ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dd[k]), leader)) : inf;   // (inf when the leader itself does not enter it: behind the children it does)
                    n_in += in[k] ? 1 : 0;
                }
                // nearest first: a sorting network over four wave-uniform (entered, key, link) triples — scalar code; children nobody enters sort last
#define CTL_PK_SWAP(i, j) { const bool s_ = (in[j] && !in[i]) || (in[i] == in[j] && key[j] < key[i]); if (s_) { const float tk = key[i]; key[i] = key[j]; key[j] = tk; const int tc = c[i]; c[i] = c[j]; c[j] = tc; const bool tb = in[i]; in[i] = in[j]; in[j] = tb; } }
                CTL_PK_SWAP(0, 1) CTL_PK_SWAP(2, 3) CTL_PK_SWAP(0, 2) CTL_PK_SWAP(1, 3) CTL_PK_SWAP(1, 2)
#undef CTL_PK_SWAP
                if (n_in == 0) { node = top(sp); sp--; }
                else {
                    // the nearest entered child next, the others pushed farthest first: c[k] (k = 1 .. n_in - 1) at sp + (n_in - k)
#pragma unroll
                    for (int k = 1; k < 4; k++) if (k < n_in) {
                        push(sp + (n_in - k), c[k]);
                    }
                    sp += n_in - 1; node = c[0];
                }
            } else {
                uint32_t e = (uint32_t)(~node);
                const int nxt = top(sp);     // what comes after this leaf is known already: if it is an inner node its fetch goes out now, in front of the entry tests
                if (nxt >= 0 && nxt != kSentinel) { const float4* __restrict__ p = nodes + (nxt & ~3); wq0 = p[0]; wq1 = p[1]; wq2 = p[2]; have_w = true; }
                for (;;) {
                    leaf_words L; flat_leaf_load(S, e, L);
                    bool got = false;
                    (void)flat_leaf_eval<false, false>(S, e, L, ox, oy, oz, dx, dy, dz, tmin, sink, got);
                    if (uniform_u32(L.iw.x) & 1u) break;
                    e++;
                }
                node = nxt; sp--;
            }
        }
        if (valid && hit && !(ray_word >> 31)) { hit[my] = make_float4(ht, 0.0f, 0.0f, __int_as_float(-1)); hit_node[my] = -1; if (S.hit_key_out) S.hit_key_out[my] = 0; }   // (a found hit wrote its record when it was accepted)
    }
}

} // namespace ctl
