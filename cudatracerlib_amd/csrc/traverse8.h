// traverse8.h — traversal of the flattened world-space BVH with 8-wide, 128-byte nodes (flatten.h flat8_node).
//
// Why eight: tools/gather_probe.hip shows that an MI355X serves ~56-60 G random records/s out of HBM (95-127 G/s out of L2)
// no matter whether a record is 16, 64 or 128 bytes — a traversal step costs one RECORD, not its bytes.  A node that fills the
// 128-byte L2 line it occupies tests eight children per fetch; rays need ~20 node fetches where the 4-wide layout needs ~34.
// Execution model as in traverse.h: persistent waves, lane refill, one fetch group per lane per iteration (node or leaf entry),
// LDS stack with a spare row for unused push slots.
#pragma once
#include "traverse.h"

namespace ctl {

template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void intersect_flat8(const dev_scene& S, const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, uint32_t* __restrict__ work,
                                                float4* __restrict__ hit, int* __restrict__ hit_node, uint32_t* __restrict__ occ, int* lds_stack, trav_counts& cnt) {
    const int lane = threadIdx.x & 63;
    const int refill_idle = g_refill_idle;
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
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work, kChunk);
                base = __shfl(base, 0, 64);
                chunk_next = base; chunk_end = base + kChunk < n ? base + kChunk : n;
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

        if (has_ray) {
            const bool is_leaf = node < 0;
            const float4* __restrict__ p = is_leaf ? leaves + (size_t)(~node) * 4 : nodes + node;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            float4 q4 = make_float4(0, 0, 0, 0), q5 = q4;
            if (!is_leaf) { q4 = p[4]; q5 = p[5]; }
            bool finished = false;
            if (!is_leaf) {
                if (COUNT) { cnt.n_inner++; if (lane == (int)__builtin_ctzll(__ballot(1))) cnt.w_inner++; }
                // q0 = origin.xyz + {ex,ey,ez,mask}; q1 = qlo_x[2], qhi_x[2]; q2 = qlo_y[2], qhi_y[2]; q3 = qlo_z[2], qhi_z[2]; q4, q5 = children
                const uint32_t meta = __float_as_uint(q0.w);
                const float ax = __uint_as_float((meta & 0xffu) << 23) * idx, ay = __uint_as_float(((meta >> 8) & 0xffu) << 23) * idy, az = __uint_as_float(((meta >> 16) & 0xffu) << 23) * idz;
                const float bx = __builtin_fmaf(q0.x, idx, -oox), by = __builtin_fmaf(q0.y, idy, -ooy), bz = __builtin_fmaf(q0.z, idz, -ooz);
                // near / far plane words by the sign of the ray direction
                const bool px = idx >= 0.0f, py = idy >= 0.0f, pz = idz >= 0.0f;
                const uint32_t nx[2] = { __float_as_uint(px ? q1.x : q1.z), __float_as_uint(px ? q1.y : q1.w) }, fx[2] = { __float_as_uint(px ? q1.z : q1.x), __float_as_uint(px ? q1.w : q1.y) };
                const uint32_t ny[2] = { __float_as_uint(py ? q2.x : q2.z), __float_as_uint(py ? q2.y : q2.w) }, fy[2] = { __float_as_uint(py ? q2.z : q2.x), __float_as_uint(py ? q2.w : q2.y) };
                const uint32_t nz[2] = { __float_as_uint(pz ? q3.x : q3.z), __float_as_uint(pz ? q3.y : q3.w) }, fz[2] = { __float_as_uint(pz ? q3.z : q3.x), __float_as_uint(pz ? q3.w : q3.y) };
                uint32_t key[8];
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int w = c >> 2, sh = 8 * (c & 3);
                    const float tnx = __builtin_fmaf((float)((nx[w] >> sh) & 0xffu), ax, bx), tfx = __builtin_fmaf((float)((fx[w] >> sh) & 0xffu), ax, bx);
                    const float tny = __builtin_fmaf((float)((ny[w] >> sh) & 0xffu), ay, by), tfy = __builtin_fmaf((float)((fy[w] >> sh) & 0xffu), ay, by);
                    const float tnz = __builtin_fmaf((float)((nz[w] >> sh) & 0xffu), az, bz), tfz = __builtin_fmaf((float)((fz[w] >> sh) & 0xffu), az, bz);
                    const float cmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                    const float cmax = fminf(fminf(tfx, tfy), fminf(tfz, ht));
                    const bool h = (cmax >= cmin) && ((meta >> (24 + c)) & 1u);
                    // sort key: entry distance (>= 0: its bit pattern orders like the float) with the child slot in the three low bits
                    key[c] = h ? ((__float_as_uint(cmin) & ~7u) | (uint32_t)c) : 0xffffffffu;
                }
                // front-to-back: 19-comparator network; misses (0xffffffff) end up last, the hits are a prefix
#define CTL_CSWAP(a, b) { const uint32_t lo_ = key[a] < key[b] ? key[a] : key[b], hi_ = key[a] < key[b] ? key[b] : key[a]; key[a] = lo_; key[b] = hi_; }
                CTL_CSWAP(0, 1) CTL_CSWAP(2, 3) CTL_CSWAP(4, 5) CTL_CSWAP(6, 7) CTL_CSWAP(0, 2) CTL_CSWAP(1, 3) CTL_CSWAP(4, 6) CTL_CSWAP(5, 7)
                CTL_CSWAP(1, 2) CTL_CSWAP(5, 6) CTL_CSWAP(0, 4) CTL_CSWAP(3, 7) CTL_CSWAP(1, 5) CTL_CSWAP(2, 6) CTL_CSWAP(1, 4) CTL_CSWAP(3, 6)
                CTL_CSWAP(2, 4) CTL_CSWAP(3, 5) CTL_CSWAP(3, 4)
#undef CTL_CSWAP
                auto child_of = [&](uint32_t k) {
                    const uint32_t s = k & 7u;
                    const float lo = (s & 2u) ? ((s & 1u) ? q4.w : q4.z) : ((s & 1u) ? q4.y : q4.x), hi = (s & 2u) ? ((s & 1u) ? q5.w : q5.z) : ((s & 1u) ? q5.y : q5.x);
                    return __float_as_int((s & 4u) ? hi : lo);
                };
                int n_hit = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) n_hit += key[c] != 0xffffffffu;
                if (n_hit == 0) { node = st.get(sp); sp--; }
                else {
                    node = child_of(key[0]);   // nearest continues, the others go onto the stack farthest first
                    const int top = sp + n_hit - 1;
                    if (top < kLdsStackFlat) {
#pragma unroll
                        for (int i = 1; i < 8; i++) { if (i >= n_hit) break; st.lds[(top - (i - 1)) * 256] = child_of(key[i]); }
                    } else {
#pragma unroll
                        for (int i = 7; i >= 1; i--) if (i < n_hit) st.set(sp + n_hit - i, child_of(key[i]));
                    }
                    sp = top;
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
                        if (v >= 0.0f && u + v <= 1.0f) {
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
