// traverse.h — two-level BVH traversal + Woop triangle test for one ray per lane (gfx950, wave64).
// Behaviour = intersectKernel<ANY_HIT> (Kernel/TraceHelper.cu:326-734): scene BVH -> instance transform ->
// mesh BVH -> Woop triangles, closest (or first) hit with t in (tmin, tmax).  The arithmetic that decides the
// reported hit (ray transform, t, u, v) is evaluated in the reference's expression order without FMA contraction,
// so (t, u, v, triangle) are bit-identical to the reference's host path; the slab tests only cull.
#pragma once
#include "device_scene.h"

namespace ctl {

struct trav_counts { uint32_t n_inner, n_tri, n_inst; };

struct ray_hit { float t, u, v; int tri, node; };

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

template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ ray_hit traverse(const dev_scene& S, float3 org, float tmin, float3 dir, float tmax, int* __restrict__ stack, trav_counts& cnt) {
    ray_hit h; h.t = tmax; h.tri = -1; h.node = -1; h.u = h.v = 0.0f;
    if (S.n_nodes == 0) return h;
    // current-space ray (world at the top level, object space inside an instance)
    float ox = org.x, oy = org.y, oz = org.z, dx = dir.x, dy = dir.y, dz = dir.z;
    float idx = rcp_guarded(dx), idy = rcp_guarded(dy), idz = rcp_guarded(dz);
    float oox = ox * idx, ooy = oy * idy, ooz = oz * idz;
    // world-space copies restored when an instance is left
    const float widx = idx, widy = idy, widz = idz, woox = oox, wooy = ooy, wooz = ooz;
    const float4* __restrict__ nodes = S.top_nodes;
    int sp = 0; stack[0] = kSentinel;
    int node = S.start_node;
    bool bottom = false; int sp_enter = 0, cur_inst = -1; uint32_t leaf_base = 0, tri_base = 0;

    while (node != kSentinel) {
        // ---- inner nodes
        while ((unsigned)node < (unsigned)kSentinel) {
            const float4 n0 = nodes[node], n1 = nodes[node + 1], nz = nodes[node + 2], cn = nodes[node + 3];
            if (COUNT) cnt.n_inner++;
            float c0min, c0max, c1min, c1max;
            slab2(n0, n1, nz, idx, idy, idz, oox, ooy, ooz, tmin, h.t, c0min, c0max, c1min, c1max);
            int c0 = __float_as_int(cn.x), c1 = __float_as_int(cn.y);
            const bool t0 = (c0max >= c0min), t1 = (c1max >= c1min);
            if (!t0 && !t1) { node = stack[sp]; sp--; }
            else {
                node = t0 ? c0 : c1;
                if (t0 && t1) { if (c1min < c0min) { int t = node; node = c1; c1 = t; } sp++; stack[sp] = c1; }
            }
        }
        if (node < 0) {
            if (!bottom) {
                // ---- enter instance ~node (TraceHelper.cu:526-560)
                cur_inst = ~node;
                if (COUNT) cnt.n_inst++;
                const float4 r0 = S.inst[cur_inst * 4], r1 = S.inst[cur_inst * 4 + 1], r2 = S.inst[cur_inst * 4 + 2], r3 = S.inst[cur_inst * 4 + 3];
                m34 m; m.r[0][0] = r0.x; m.r[0][1] = r0.y; m.r[0][2] = r0.z; m.r[0][3] = r0.w; m.r[1][0] = r1.x; m.r[1][1] = r1.y; m.r[1][2] = r1.z; m.r[1][3] = r1.w;
                m.r[2][0] = r2.x; m.r[2][1] = r2.y; m.r[2][2] = r2.z; m.r[2][3] = r2.w;
                const f3 d = xform_dir(m, f3(dir.x, dir.y, dir.z)), o = xform_point_w(m, f3(org.x, org.y, org.z), r3.x);
                ox = o.x; oy = o.y; oz = o.z; dx = d.x; dy = d.y; dz = d.z;
                idx = rcp_guarded(dx); idy = rcp_guarded(dy); idz = rcp_guarded(dz);
                oox = ox * idx; ooy = oy * idy; ooz = oz * idz;
                nodes = S.bot_nodes + __float_as_uint(r3.y);
                leaf_base = __float_as_uint(r3.z); tri_base = __float_as_uint(r3.w);
                sp++; stack[sp] = kExitMarker; sp_enter = sp;
                bottom = true; node = 0;
            } else {
                // ---- leaf: Woop triangles (TraceHelper.cu:636-694)
                for (uint32_t addr = leaf_base + (uint32_t)(~node);; addr++) {
                    const float4 v00 = S.leaf_tris[addr * 4], v11 = S.leaf_tris[addr * 4 + 1], v22 = S.leaf_tris[addr * 4 + 2];
                    const uint32_t index = __float_as_uint(S.leaf_tris[addr * 4 + 3].x);
                    if (COUNT) cnt.n_tri++;
                    const float Oz = v00.w - ox * v00.x - oy * v00.y - oz * v00.z;
                    const float invDz = 1.0f / (dx * v00.x + dy * v00.y + dz * v00.z);
                    const float t = Oz * invDz;
                    if (t > tmin && t < h.t) {
                        const float Ox = v11.w + ox * v11.x + oy * v11.y + oz * v11.z;
                        const float Dx = dx * v11.x + dy * v11.y + dz * v11.z;
                        const float u = Ox + t * Dx;
                        if (u >= 0.0f) {
                            const float Oy = v22.w + ox * v22.x + oy * v22.y + oz * v22.z;
                            const float Dy = dx * v22.x + dy * v22.y + dz * v22.z;
                            const float v = Oy + t * Dy;
                            if (v >= 0.0f && u + v <= 1.0f) {
                                h.t = t; h.u = u; h.v = v; h.tri = (int)((index >> 1) + tri_base); h.node = cur_inst;
                                if (ANY_HIT) return h;
                            }
                        }
                    }
                    if (index & 1) break;
                }
                node = stack[sp]; sp--;
            }
        }
        // a mesh BVH that ends in the sentinel (one-leaf meshes carry 0x76543210 as second child) only ends that mesh
        if (bottom && node == kSentinel) { sp = sp_enter - 1; node = kExitMarker; }
        if (node == kExitMarker) {
            ox = org.x; oy = org.y; oz = org.z; dx = dir.x; dy = dir.y; dz = dir.z;
            idx = widx; idy = widy; idz = widz; oox = woox; ooy = wooy; ooz = wooz;
            nodes = S.top_nodes; bottom = false;
            node = stack[sp]; sp--;
        }
    }
    return h;
}

} // namespace ctl
