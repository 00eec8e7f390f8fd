// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// ocore.h — scene access, Woop triangles, two-level BVH traversal, TriangleData / fillDG, sampler,
// sensor, lights, BSDFs and the megakernel PathTrace<DIRECT> — the semantic oracle of the path.
#pragma once
#include <atomic>
#include <cstdlib>
#include "omath.h"
#include "../include/ctl_amd.h"   // boundary structs only (data layout contract)
#include "../cudatracerlib_amd/csrc/flat8.h"   // layout + link decode of the product's 8-wide node (a data-layout contract as well: the mirror below restates the arithmetic)
#include <vector>
#include <stdexcept>

namespace orc {

static const int EntrypointSentinel = 0x76543210;   // Kernel/TraceHelper.cu:20

// --------------------------------------------------------------------------- Woop (Engine/TriIntersectorData.cu)
// TriIntersectorData.cu:5-18
inline void woopSetData(ctl_woop_tri& w, V3 a, V3 b, V3 c) {
    M44 m;
    V3 e0 = a - c, e1 = b - c, n = cross(a - c, b - c);
    m(0, 0) = e0.x; m(1, 0) = e0.y; m(2, 0) = e0.z; m(3, 0) = 0;
    m(0, 1) = e1.x; m(1, 1) = e1.y; m(2, 1) = e1.z; m(3, 1) = 0;
    m(0, 2) = n.x;  m(1, 2) = n.y;  m(2, 2) = n.z;  m(3, 2) = 0;
    m(0, 3) = c.x;  m(1, 3) = c.y;  m(2, 3) = c.z;  m(3, 3) = 1;
    m = inverse(m);
    w.a[0] = m(2, 0); w.a[1] = m(2, 1); w.a[2] = m(2, 2); w.a[3] = -m(2, 3);
    for (int j = 0; j < 4; j++) { w.b[j] = m(0, j); w.c[j] = m(1, j); }
}
// TriIntersectorData.cu:20-32
inline void woopGetData(const ctl_woop_tri& w, V3& v0, V3& v1, V3& v2) {
    M44 m = M44::identity();
    for (int j = 0; j < 4; j++) { m(0, j) = w.b[j]; m(1, j) = w.c[j]; m(2, j) = w.a[j]; }
    m(2, 3) *= -1.0f;
    m = inverse(m);
    V3 e02(m(0, 0), m(1, 0), m(2, 0)), e12(m(0, 1), m(1, 1), m(2, 1));
    v2 = V3(m(0, 3), m(1, 3), m(2, 3));
    v0 = v2 + e02;
    v1 = v2 + e12;
}
// The per-triangle test, identical in TriIntersectorData.cu:34-61, TraceHelper.cu:118-165 (host traceRay)
// and TraceHelper.cu:646-682 (intersectKernel).  Returns true and updates (t,u,v) when accepted.
inline bool woopIntersect(const ctl_woop_tri& w, V3 o, V3 d, float tmin, float tmax, float& t_out, float& u_out, float& v_out) {
    float Oz = w.a[3] - o.x * w.a[0] - o.y * w.a[1] - o.z * w.a[2];
    float invDz = 1.0f / (d.x * w.a[0] + d.y * w.a[1] + d.z * w.a[2]);
    float t = Oz * invDz;
    if (t > tmin && t < tmax) {
        float Ox = w.b[3] + o.x * w.b[0] + o.y * w.b[1] + o.z * w.b[2];
        float Dx = d.x * w.b[0] + d.y * w.b[1] + d.z * w.b[2];
        float u = Ox + t * Dx;
        if (u >= 0.0f) {
            float Oy = w.c[3] + o.x * w.c[0] + o.y * w.c[1] + o.z * w.c[2];
            float Dy = d.x * w.c[0] + d.y * w.c[1] + d.z * w.c[2];
            float v = Oy + t * Dy;
            if (v >= 0.0f && u + v <= 1.0f) { t_out = t; u_out = u; v_out = v; return true; }
        }
    }
    return false;
}

// --------------------------------------------------------------------------- traversal
struct Hit {   // Kernel/TraceResult.h:18-35
    float dist; float u, v; uint32_t tri; uint32_t node;
    bool hasHit() const { return tri != UINT32_MAX; }
    void init() { dist = FLT_MAX; tri = UINT32_MAX; node = UINT32_MAX; u = v = 0; }
};
struct TravCounts { uint64_t n_inner = 0, n_tri = 0, n_inst = 0; std::vector<uint32_t>* node_log = nullptr; std::vector<uint32_t>* entry_log = nullptr; };   // logs (flattened Q4 traversal only): which node / leaf entry each step looked at (orc_packet_union_probe)

// Math/MathFunc.h:443-444 — integer min/max on float bit patterns; for tmin >= 0 and non-NaN inputs the decision
// `cmax >= cmin` is the same as with float min/max (negative entries lose against d >= 0 in spanBegin and make
// spanEnd negative either way), so the restatement uses float semantics.
inline float spanBegin(float a0, float a1, float b0, float b1, float c0, float c1, float d) {
    return fmax2(fmax2(fmin2(a0, a1), fmin2(b0, b1)), fmax2(fmin2(c0, c1), d));
}
inline float spanEnd(float a0, float a1, float b0, float b1, float c0, float c1, float d) {
    return fmin2(fmin2(fmax2(a0, a1), fmax2(b0, b1)), fmin2(fmax2(c0, c1), d));
}

// Engine/SpatialStructures/BVH/BVHTraversal.h:122-232 (pointer overload).  `nodes` is a float4 view; node addresses
// are float4 indices.  `node_tmin`: 0 on the single-ray path (BVHTraversal.h:169-176), the ray's tmin on the
// wavefront path (TraceHelper.cu:469-476) — only culling differs, never the accepted hit.
template <typename CLB>
inline bool tracerayTemplate(V3 ori, V3 dir, float& rayT, float node_tmin, const CLB& clb, const float* nodes4, int bvhNodesOffset,
                             int startNode, TravCounts* cnt, const bool* stop = nullptr) {
    if (startNode < 0) return clb(~startNode);
    bool found = false;
    int stack[64]; stack[0] = EntrypointSentinel;
    const float ooeps = exp2f(-80.0f);
    float idirx = 1.0f / (fabsf(dir.x) > ooeps ? dir.x : copysign_bits(ooeps, dir.x));
    float idiry = 1.0f / (fabsf(dir.y) > ooeps ? dir.y : copysign_bits(ooeps, dir.y));
    float idirz = 1.0f / (fabsf(dir.z) > ooeps ? dir.z : copysign_bits(ooeps, dir.z));
    float oodx = ori.x * idirx, oody = ori.y * idiry, oodz = ori.z * idirz;
    int sp = 0, leafAddr = 0, nodeAddr = startNode;
    while (nodeAddr != EntrypointSentinel && !(stop && *stop)) {
        while ((unsigned)nodeAddr < (unsigned)EntrypointSentinel) {
            const float* n = nodes4 + (size_t)(bvhNodesOffset + nodeAddr) * 4;
            if (cnt) cnt->n_inner++;
            const float c0lox = n[0] * idirx - oodx, c0hix = n[1] * idirx - oodx, c0loy = n[2] * idiry - oody, c0hiy = n[3] * idiry - oody;
            const float c0loz = n[8] * idirz - oodz, c0hiz = n[9] * idirz - oodz, c1loz = n[10] * idirz - oodz, c1hiz = n[11] * idirz - oodz;
            const float c0min = spanBegin(c0lox, c0hix, c0loy, c0hiy, c0loz, c0hiz, node_tmin);
            const float c0max = spanEnd(c0lox, c0hix, c0loy, c0hiy, c0loz, c0hiz, rayT);
            const float c1lox = n[4] * idirx - oodx, c1hix = n[5] * idirx - oodx, c1loy = n[6] * idiry - oody, c1hiy = n[7] * idiry - oody;
            const float c1min = spanBegin(c1lox, c1hix, c1loy, c1hiy, c1loz, c1hiz, node_tmin);
            const float c1max = spanEnd(c1lox, c1hix, c1loy, c1hiy, c1loz, c1hiz, rayT);
            int cx, cy; std::memcpy(&cx, n + 12, 4); std::memcpy(&cy, n + 13, 4);
            bool swp = (c1min < c0min), t0 = (c0max >= c0min), t1 = (c1max >= c1min);
            if (!t0 && !t1) { nodeAddr = stack[sp]; sp--; }
            else {
                nodeAddr = t0 ? cx : cy;
                if (t0 && t1) { if (swp) std::swap(nodeAddr, cy); sp++; stack[sp] = cy; }
            }
            if (nodeAddr < 0 && leafAddr >= 0) { leafAddr = nodeAddr; nodeAddr = stack[sp]; sp--; }
            if (!(leafAddr >= 0)) break;   // host: mask = leafAddr >= 0 (BVHTraversal.h:209-213)
        }
        while (leafAddr < 0 && !(stop && *stop)) {
            found |= clb(~leafAddr);
            leafAddr = nodeAddr;
            if (nodeAddr < 0) { nodeAddr = stack[sp]; sp--; }
        }
    }
    return found;
}

struct Scene {
    ctl_scene_desc d;
    bool half_host_quirk = false;   // reproduce half::ToFloat's host branch (Math/half.h:76-83)
    bool alpha_test = false;        // KernelDynamicScene::doAlphaMapping (DynamicScene.cpp:586): traceRay<USE_ALPHA = true>
    const ctl_flat_bvh_desc* flat = nullptr;   // when set, traceRay walks the product's flattened BVH (the arrays ctl_flat_bvh_build hands out)
    const struct MipPyramid* pyramids = nullptr;   // one per image (built by the caller) when first-hit ray differentials are on
};
inline bool sceneHasAlphaMaps(const ctl_scene_desc& d) {   // MaterialBuffer::hasAlphaMappings
    for (uint32_t i = 0; i < d.n_materials; i++) if (d.materials[i].alpha_state != CTL_ALPHA_DISABLED) return true;
    return false;
}
inline bool alphaSurvive(const Scene& S, uint32_t tri, uint32_t nodeIdx, float u, float v);   // TraceHelper.cu:135-153, defined with the textures

// Kernel/TraceHelper.cu:88-180 (__traceRay_internal__<false> + traceRay).  any_hit/tmax generalise it to the
// wavefront kernel's interface (TraceHelper.cu:326-734): a ctl_ray carries tmin in a.w and tmax in b.w.
// Traversal of the product's FLATTENED world-space BVH (cudatracerlib_amd/csrc/flatten.h; not a reference structure).  The tree only
// culls: every leaf entry is evaluated exactly as the two-level traversal evaluates a triangle — the ray through the node's inverse
// transform (TraceHelper.cu:526-560), then the Woop test (:646-682) — so the accepted hit is the reference's.  Plain depth-first
// order, children nearest first (the product's kernel visits speculatively and may count more nodes; the counts returned here are the
// algorithmic ones).  n_inst stays 0: there is no instance entry.
// measurement probe for tools/bvh_quality_probe.py --slab (off unless orc_slab_probe(1) was called)
inline bool g_slab_probe = false; inline std::atomic<uint64_t> g_slab_tests{ 0 }, g_slab_rejects{ 0 };
// same switch: node visits by the node's position in the array (the top of the tree is stored breadth-first): bucket b counts visits of nodes with index < g_top_probe_limits[b]
inline const uint32_t g_top_probe_limits[8] = { 85u, 256u, 341u, 512u, 1365u, 5461u, 65536u, 0xffffffffu }; inline std::atomic<uint64_t> g_top_probe[8];
// The product's 8-wide node format (CTL_FLAT_Q8, cudatracerlib_amd/csrc/flat8.h): octant-ordered slots, one sibling group per level on the stack, the leaf slots a step hits
// tested before its inner children are entered.  Same culling arithmetic as the 4-wide mirror below (8-bit boxes, the oriented slab as a fourth axis), restated.
// $ORC_Q8_ORDER = dist (what-if): children nearest first by entry distance instead of by octant, to price the octant order.
inline bool traceRayFlat8(const Scene& S, V3 ori, V3 dir, float tmin_tri, float tmax, bool any_hit, float node_tmin, Hit& res, TravCounts* cnt) {
    const ctl_scene_desc& g = S.d; const ctl_flat_bvh_desc& F = *S.flat;
    res.init(); res.dist = tmax;
    if (!F.n_nodes || !F.n_leaves) return false;
    static const bool by_dist = [] { const char* e = getenv("ORC_Q8_ORDER"); return e && !strcmp(e, "dist"); }();
    const float ooeps = exp2f(-80.0f), inf = INFINITY;
    const float idx = 1.0f / (fabsf(dir.x) > ooeps ? dir.x : copysign_bits(ooeps, dir.x)), idy = 1.0f / (fabsf(dir.y) > ooeps ? dir.y : copysign_bits(ooeps, dir.y)),
                idz = 1.0f / (fabsf(dir.z) > ooeps ? dir.z : copysign_bits(ooeps, dir.z));
    const float oox = ori.x * idx, ooy = ori.y * idy, ooz = ori.z * idz;
    const int sx = idx < 0.0f, sy = idy < 0.0f, sz = idz < 0.0f;
    const uint32_t octinv = (sx ? 0u : 1u) | (sy ? 0u : 2u) | (sz ? 0u : 4u);
    const uint32_t* nodes = (const uint32_t*)F.nodes; const uint32_t* leaves = (const uint32_t*)F.leaves;
    struct group { uint32_t base_b, imask, hits; };          // hits in visiting order (flat8_to_order): the highest set bit first
    std::vector<group> stack((size_t)F.max_depth + 8); int sp = 0;
    std::vector<std::pair<float, uint32_t>> dstack;          // by_dist: (entry distance, node link) singles
    bool found = false;
    uint32_t node = F.root_slab ? 1u : 0u;                   // link = node index << 1 | "load q5" (flat8.h B of an inner slot)
    group cur{ 0, 0, 0 };
    auto p2 = [](uint32_t e) { uint32_t b = e << 23; float f; std::memcpy(&f, &b, 4); return f; };
    for (;;) {
        if (cnt) cnt->n_inner++;
        const uint32_t* w = nodes + (size_t)(node >> 1) * 32; const float* pf = (const float*)w;
        const uint32_t q0w = w[3], base_b = w[4], leaf_base = w[5];
        const uint32_t imask = ctl::flat8_inner_mask(q0w), lmask = ctl::flat8_leaf_mask(q0w, base_b);
        const float ax = p2(q0w & 0xffu) * idx, ay = p2((q0w >> 8) & 0xffu) * idy, az = p2((q0w >> 16) & 0xffu) * idz;
        const float bx = std::fmaf(pf[0], idx, -oox), by = std::fmaf(pf[1], idy, -ooy), bz = std::fmaf(pf[2], idz, -ooz);
        const uint32_t* lox = w + 8, *loy = w + 10, *loz = w + 12, *hix = w + 14, *hiy = w + 16, *hiz = w + 18;
        const uint32_t *nx = sx ? hix : lox, *fx = sx ? lox : hix, *ny = sy ? hiy : loy, *fy = sy ? loy : hiy, *nz = sz ? hiz : loz, *fz = sz ? loz : hiz;
        float s_alpha = 0.0f, s_bn = -inf, s_bf = inf; const uint32_t zero2[2] = { 0u, 0u }; const uint32_t *s_nw = zero2, *s_fw = zero2;
        if ((node & 1u) && w[6] != 0u) {   // the node's oriented slab (csrc/flat_slab.h), a fourth slab axis
            const uint32_t nw = w[6];
            auto s6 = [](uint32_t v) { return (float)((int)(v & 63u) - (int)((v & 32u) << 1)); };
            uint32_t sb = nw & 0xfffc0000u; float step; std::memcpy(&step, &sb, 4);
            const float snx = s6(nw), sny = s6(nw >> 6), snz = s6(nw >> 12), base = pf[7];
            const float ex = ori.x - pf[0], ey = ori.y - pf[1], ez = ori.z - pf[2];
            const float sdot = std::fmaf(snz, ez, std::fmaf(sny, ey, snx * ex)), rdot = std::fmaf(snz, dir.z, std::fmaf(sny, dir.y, snx * dir.x));
            const float rr = 1.0f / (fabsf(rdot) > ooeps ? rdot : copysign_bits(ooeps, rdot));
            const float pad = (fabsf(ex) + fabsf(ey) + fabsf(ez)) * (31.0f * 1.9073486328125e-6f), u = base - sdot;
            const bool neg = rr < 0.0f;
            s_alpha = step * rr; s_bn = (neg ? u + pad : u - pad) * rr; s_bf = (neg ? u - pad : u + pad) * rr;
            s_nw = neg ? w + 22 : w + 20; s_fw = neg ? w + 20 : w + 22;
        }
        uint32_t hits = 0; float dd[8];
        for (int k = 0; k < 8; k++) {
            auto code = [&](const uint32_t* a) { return (float)((a[k >> 2] >> (8 * (k & 3))) & 0xffu); };
            const float tnx = std::fmaf(code(nx), ax, bx), tfx = std::fmaf(code(fx), ax, bx), tny = std::fmaf(code(ny), ay, by), tfy = std::fmaf(code(fy), ay, by);
            const float tnz = std::fmaf(code(nz), az, bz), tfz = std::fmaf(code(fz), az, bz);
            const float tns = std::fmaf(code(s_nw), s_alpha, s_bn), tfs = std::fmaf(code(s_fw), s_alpha, s_bf);
            const float cmin = fmax2(fmax2(fmax2(tnx, tny), fmax2(tnz, node_tmin)), tns), cmax = fmin2(fmin2(fmin2(tfx, tfy), fmin2(tfz, res.dist)), tfs);
            dd[k] = cmin;
            if (cmax >= cmin) hits |= 1u << k;
        }
        // leaf slots first, in visiting order
        for (uint32_t lo = ctl::flat8_to_order(hits & lmask, octinv); lo;) {
            const uint32_t bit = 31u - (uint32_t)__builtin_clz(lo); lo &= ~(1u << bit);
            const uint32_t slot = bit ^ octinv, entry = ctl::flat8_leaf_entry(leaf_base, lmask, slot);
            const uint32_t* e = leaves + (size_t)entry * 32;
            const uint32_t index = e[12], nodeIdx = e[13];
            if (cnt) cnt->n_tri++;
            ctl_woop_tri wt; std::memcpy(&wt, e, 48);
            M44 modl; std::memcpy(modl.d, g.node_inv_transforms[nodeIdx].m, 64);
            const V3 d = transformDir(modl, dir), o = transformPoint(modl, ori);   // TraceHelper.cu:526-560 (per entry here; per instance there)
            float t, u, v;
            if (woopIntersect(wt, o, d, tmin_tri, res.dist, t, u, v) && (!S.alpha_test || alphaSurvive(S, index >> 1, nodeIdx, u, v))) {
                res.node = nodeIdx; res.tri = index >> 1; res.u = u; res.v = v; res.dist = t; found = true;
                if (any_hit) return true;
            }
        }
        const uint32_t inner = hits & imask;
        if (by_dist) {
            std::pair<float, uint32_t> c[8]; int n = 0;
            for (uint32_t s = 0; s < 8; s++) if ((inner >> s) & 1u) c[n++] = { dd[s], (ctl::flat8_child_node(base_b, imask, s) << 1) | ((base_b >> (24 + s)) & 1u) };
            std::sort(c, c + n, [](const auto& a, const auto& b) { return a.first > b.first; });   // farthest first: the nearest is pushed last
            for (int i = 0; i < n; i++) dstack.push_back(c[i]);
            if (dstack.empty()) return found;
            node = dstack.back().second; dstack.pop_back();
            continue;
        }
        const uint32_t ordered = ctl::flat8_to_order(inner, octinv);
        if (ordered) { if (cur.hits) stack[++sp] = cur; cur = group{ base_b, imask, ordered }; }
        else if (!cur.hits) { if (sp == 0) return found; cur = stack[sp--]; }
        const uint32_t bit = 31u - (uint32_t)__builtin_clz(cur.hits); cur.hits &= ~(1u << bit);
        const uint32_t slot = bit ^ octinv;
        node = (ctl::flat8_child_node(cur.base_b, cur.imask, slot) << 1) | ((cur.base_b >> (24 + slot)) & 1u);
    }
}

inline bool traceRayFlat(const Scene& S, V3 ori, V3 dir, float tmin_tri, float tmax, bool any_hit, float node_tmin, Hit& res, TravCounts* cnt) {
    if (S.flat->format == CTL_FLAT_Q8) return traceRayFlat8(S, ori, dir, tmin_tri, tmax, any_hit, node_tmin, res, cnt);
    const ctl_scene_desc& g = S.d; const ctl_flat_bvh_desc& F = *S.flat;
    res.init(); res.dist = tmax;
    if (!F.n_nodes || !F.n_leaves) return false;
    const float ooeps = exp2f(-80.0f);
    const float idx = 1.0f / (fabsf(dir.x) > ooeps ? dir.x : copysign_bits(ooeps, dir.x)), idy = 1.0f / (fabsf(dir.y) > ooeps ? dir.y : copysign_bits(ooeps, dir.y)),
                idz = 1.0f / (fabsf(dir.z) > ooeps ? dir.z : copysign_bits(ooeps, dir.z));
    const float oox = ori.x * idx, ooy = ori.y * idy, ooz = ori.z * idz;
    const int sx = idx < 0.0f, sy = idy < 0.0f, sz = idz < 0.0f;
    const float* nodes = (const float*)F.nodes; const uint32_t* leaves = (const uint32_t*)F.leaves;
    std::vector<int> stack(4 * (size_t)F.max_depth + 8); int sp = 0; stack[0] = EntrypointSentinel;
    // ORC_STACK_CULL=1 (what-if): every pushed child carries its entry distance and a pop that lies behind the hit found meanwhile is dropped —
    // the product's -DCTL_STACK_DIST=1 build (csrc/traverse_flat.h), measured and not shipped
    std::vector<float> sdist(stack.size(), -INFINITY); static const bool stack_cull = getenv("ORC_STACK_CULL") != nullptr;
    auto pop = [&]() { for (;;) { const int n = stack[sp]; const float dn = sdist[sp]; sp--; if (!stack_cull || !(dn >= res.dist)) return n; } };
    int node = 0; bool found = false;
    while (node != EntrypointSentinel) {
        if (node >= 0) {
            if (cnt) cnt->n_inner++;
            if (cnt && cnt->node_log) cnt->node_log->push_back((uint32_t)node >> 2);
            if (cnt && g_slab_probe) { const uint32_t ni = (uint32_t)node >> 2; for (int b = 0; b < 8; b++) if (ni < g_top_probe_limits[b]) g_top_probe[b].fetch_add(1, std::memory_order_relaxed); }
            const float* p = nodes + (size_t)node * 4;
            float dd[4]; int c[4]; int width = 4;
            const float inf = INFINITY;
            if (F.format == CTL_FLAT_F4) {
                const float *nx = p + 4 * sx, *fx = p + 4 * (1 - sx), *ny = p + 4 * (2 + sy), *fy = p + 4 * (3 - sy), *nz = p + 4 * (4 + sz), *fz = p + 4 * (5 - sz);
                for (int k = 0; k < 4; k++) {
                    const float tnx = std::fmaf(nx[k], idx, -oox), tfx = std::fmaf(fx[k], idx, -oox), tny = std::fmaf(ny[k], idy, -ooy), tfy = std::fmaf(fy[k], idy, -ooy);
                    const float tnz = std::fmaf(nz[k], idz, -ooz), tfz = std::fmaf(fz[k], idz, -ooz);
                    const float cmin = fmax2(fmax2(tnx, tny), fmax2(tnz, node_tmin)), cmax = fmin2(fmin2(tfx, tfy), fmin2(tfz, res.dist));
                    dd[k] = (cmax >= cmin) ? cmin : inf; std::memcpy(&c[k], p + 24 + k, 4);
                }
            } else if (F.format == CTL_FLAT_Q4) {
                // the product's quantised 4-wide node (cudatracerlib_amd/csrc/flatten.h): with implied links (F.compact) bit 0 of an inner link says that
                // the node's last 16 B are an oriented slab — a fourth slab axis along a direction n of the node's own (csrc/flat_slab.h), restated here
                const bool slab = F.compact && (node & 1);
                p = nodes + (size_t)(node & ~3) * 4;
                uint32_t w[16]; std::memcpy(w, p, 64);
                const uint32_t meta = w[3];
                auto p2 = [](uint32_t e) { uint32_t b = e << 23; float f; std::memcpy(&f, &b, 4); return f; };
                const float ax = p2(meta & 0xffu) * idx, ay = p2((meta >> 8) & 0xffu) * idy, az = p2((meta >> 16) & 0xffu) * idz;
                const float bx = std::fmaf(p[0], idx, -oox), by = std::fmaf(p[1], idy, -ooy), bz = std::fmaf(p[2], idz, -ooz);
                const uint32_t nx = sx ? w[5] : w[4], fx = sx ? w[4] : w[5], ny = sy ? w[7] : w[6], fy = sy ? w[6] : w[7], nz = sz ? w[9] : w[8], fz = sz ? w[8] : w[9];
                float s_alpha = 0.0f, s_bn = -inf, s_bf = inf; uint32_t s_nw = 0, s_fw = 0;
                if (slab) {
                                        const uint32_t nw = w[12];
                    auto s6 = [](uint32_t v) { return (float)((int)(v & 63u) - (int)((v & 32u) << 1)); };   // 6-bit two's complement
                    uint32_t sb = nw & 0xfffc0000u; float step; std::memcpy(&step, &sb, 4);
                    const float snx = s6(nw), sny = s6(nw >> 6), snz = s6(nw >> 12), base = p[13];
                    const float ex = ori.x - p[0], ey = ori.y - p[1], ez = ori.z - p[2];
                    const float sdot = std::fmaf(snz, ez, std::fmaf(sny, ey, snx * ex)), rdot = std::fmaf(snz, dir.z, std::fmaf(sny, dir.y, snx * dir.x));
                    const float rr = 1.0f / (fabsf(rdot) > ooeps ? rdot : copysign_bits(ooeps, rdot));
                    const float pad = (fabsf(ex) + fabsf(ey) + fabsf(ez)) * (31.0f * 1.9073486328125e-6f), u = base - sdot;
                    const bool neg = rr < 0.0f;
                    s_alpha = step * rr; s_bn = (neg ? u + pad : u - pad) * rr; s_bf = (neg ? u - pad : u + pad) * rr;
                    s_nw = neg ? w[15] : w[14]; s_fw = neg ? w[14] : w[15];
                }
                int32_t links[4];
                if (F.compact) {
                    // flatten.h: link = base + nibble.  w0: first inner child * 4 | slab flag of slot 0, t1 in bits 26..29, t3 & 3 in bits 30..31; w1: t3 >> 2, t2 in bits 2..5, ~(first entry + 15) above
                    const uint32_t w0 = w[10], w1 = w[11], leafm = meta >> 28;
                    const uint32_t ib4 = w0 & 0x03fffffcu, nlb15 = (w1 >> 6) | 0xfc000000u;
                    const uint32_t t[4] = { 0u, (w0 >> 26) & 15u, (w1 >> 2) & 15u, (w0 >> 30) | ((w1 & 3u) << 2) };
                    links[0] = (leafm & 1u) ? (int32_t)(nlb15 + 15u) : (int32_t)(w0 & 0x03ffffffu);
                    for (int k = 1; k < 4; k++) links[k] = (int32_t)((((leafm >> k) & 1u) ? nlb15 : ib4) + t[k]);
                } else for (int k = 0; k < 4; k++) links[k] = (int32_t)w[12 + k];
                for (int k = 0; k < 4; k++) {
                    const float tnx = std::fmaf((float)((nx >> (8 * k)) & 0xffu), ax, bx), tfx = std::fmaf((float)((fx >> (8 * k)) & 0xffu), ax, bx);
                    const float tny = std::fmaf((float)((ny >> (8 * k)) & 0xffu), ay, by), tfy = std::fmaf((float)((fy >> (8 * k)) & 0xffu), ay, by);
                    const float tnz = std::fmaf((float)((nz >> (8 * k)) & 0xffu), az, bz), tfz = std::fmaf((float)((fz >> (8 * k)) & 0xffu), az, bz);
                    const float tns = std::fmaf((float)((s_nw >> (8 * k)) & 0xffu), s_alpha, s_bn), tfs = std::fmaf((float)((s_fw >> (8 * k)) & 0xffu), s_alpha, s_bf);
                    const float bmin = fmax2(fmax2(tnx, tny), fmax2(tnz, node_tmin)), bmax = fmin2(fmin2(tfx, tfy), fmin2(tfz, res.dist));
                    const float cmin = fmax2(bmin, tns), cmax = fmin2(bmax, tfs);
                    const bool exists = ((meta >> (24 + k)) & 1u) != 0;
                    if (cnt && g_slab_probe && slab && exists && ((meta >> (28 + k)) & 1u) && bmax >= bmin) { g_slab_tests.fetch_add(1, std::memory_order_relaxed); if (!(cmax >= cmin)) g_slab_rejects.fetch_add(1, std::memory_order_relaxed); }   // probe: leaf children of slab nodes the box lets in / the slab keeps out
                    dd[k] = ((cmax >= cmin) && exists) ? cmin : inf; c[k] = links[k];
                }
            } else {   // CTL_FLAT_F2: BVHNodeData
                width = 2;
                const float c0lox = std::fmaf(p[0], idx, -oox), c0hix = std::fmaf(p[1], idx, -oox), c0loy = std::fmaf(p[2], idy, -ooy), c0hiy = std::fmaf(p[3], idy, -ooy);
                const float c1lox = std::fmaf(p[4], idx, -oox), c1hix = std::fmaf(p[5], idx, -oox), c1loy = std::fmaf(p[6], idy, -ooy), c1hiy = std::fmaf(p[7], idy, -ooy);
                const float c0loz = std::fmaf(p[8], idz, -ooz), c0hiz = std::fmaf(p[9], idz, -ooz), c1loz = std::fmaf(p[10], idz, -ooz), c1hiz = std::fmaf(p[11], idz, -ooz);
                const float c0min = spanBegin(c0lox, c0hix, c0loy, c0hiy, c0loz, c0hiz, node_tmin), c0max = spanEnd(c0lox, c0hix, c0loy, c0hiy, c0loz, c0hiz, res.dist);
                const float c1min = spanBegin(c1lox, c1hix, c1loy, c1hiy, c1loz, c1hiz, node_tmin), c1max = spanEnd(c1lox, c1hix, c1loy, c1hiy, c1loz, c1hiz, res.dist);
                dd[0] = (c0max >= c0min) ? c0min : inf; dd[1] = (c1max >= c1min) ? c1min : inf; dd[2] = dd[3] = inf;
                std::memcpy(&c[0], p + 12, 4); std::memcpy(&c[1], p + 13, 4); c[2] = c[3] = EntrypointSentinel;
            }
            auto cswap = [&](int i, int j) { if (dd[j] < dd[i]) { std::swap(dd[i], dd[j]); std::swap(c[i], c[j]); } };
            if (width == 4) { cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2); } else cswap(0, 1);
            int n_hit = 0; for (int k = 0; k < 4; k++) if (dd[k] < inf) n_hit++;
            for (int i = n_hit - 1; i >= 1; i--) { stack[++sp] = c[i]; sdist[sp] = dd[i]; }
            node = n_hit ? c[0] : pop();
        } else {
            const uint32_t* e = leaves + (size_t)(uint32_t)(~node) * 32;   // 128 B: Woop rows a, b, c, {globalTri << 1 | last, node, 0, 0}, then a copy of the node's inverse transform
            const uint32_t index = e[12], nodeIdx = e[13];
            if (cnt) cnt->n_tri++;
            if (cnt && cnt->entry_log) cnt->entry_log->push_back((uint32_t)(~node));
            ctl_woop_tri w; std::memcpy(&w, e, 48);
            M44 modl; std::memcpy(modl.d, g.node_inv_transforms[nodeIdx].m, 64);
            const V3 d = transformDir(modl, dir), o = transformPoint(modl, ori);   // TraceHelper.cu:526-560 (per entry here; per instance there)
            float t, u, v;
            if (woopIntersect(w, o, d, tmin_tri, res.dist, t, u, v) && (!S.alpha_test || alphaSurvive(S, index >> 1, nodeIdx, u, v))) {
                res.node = nodeIdx; res.tri = index >> 1; res.u = u; res.v = v; res.dist = t; found = true;
                if (any_hit) return true;
            }
            node = (index & 1) ? pop() : node - 1;
        }
    }
    return found;
}

inline bool traceRay(const Scene& S, V3 ori, V3 dir, float tmin_tri, float tmax, bool any_hit, float node_tmin, Hit& res, TravCounts* cnt = nullptr) {
    if (S.flat) return traceRayFlat(S, ori, dir, tmin_tri, tmax, any_hit, node_tmin, res, cnt);
    const ctl_scene_desc& g = S.d;
    res.init(); res.dist = tmax;
    if (!g.n_nodes) return false;
    bool stop = false;
    auto nodeClb = [&](int nodeIdx) -> bool {
        if (stop) return false;
        if (cnt) cnt->n_inst++;
        const ctl_node& N = g.nodes[nodeIdx];
        const ctl_kernel_mesh& mesh = g.meshes[N.mesh_index];
        M44 modl; std::memcpy(modl.d, g.node_inv_transforms[nodeIdx].m, 64);
        V3 d = transformDir(modl, dir), o = transformPoint(modl, ori);
        auto triClb = [&](int triIdx) -> bool {
            if (stop) return false;
            bool found = false;
            for (int triAddr = triIdx;; triAddr++) {
                const ctl_woop_tri& w = g.woop[mesh.bvh_tri_offset / 3 + triAddr];
                uint32_t index = g.woop_index[mesh.bvh_index_offset + triAddr].index;
                if (cnt) cnt->n_tri++;
                float t, u, v;
                if (woopIntersect(w, o, d, tmin_tri, res.dist, t, u, v) && (!S.alpha_test || alphaSurvive(S, (index >> 1) + mesh.tri_offset, nodeIdx, u, v))) {
                    res.node = nodeIdx; res.tri = (index >> 1) + mesh.tri_offset; res.u = u; res.v = v; res.dist = t;
                    found = true;
                    if (any_hit) { stop = true; break; }
                }
                if (index & 1) break;
            }
            return found;
        };
        return tracerayTemplate(o, d, res.dist, node_tmin, triClb, (const float*)g.bvh_nodes, mesh.bvh_node_offset, 0, cnt, &stop);
    };
    // any-hit: `stop` ends both levels as intersectKernel<true> does (TraceHelper.cu:684-688)
    return tracerayTemplate(ori, dir, res.dist, node_tmin, nodeClb, (const float*)g.scene_bvh_nodes, 0, g.scene_start_node, cnt, &stop);
}
// per-thread traversal counters of a render in counting mode (orc_render_counts): [0] path rays, [1] occlusion rays
struct RenderCounts { TravCounts c[2]; uint64_t rays[2] = { 0, 0 }; };
inline RenderCounts*& renderCounts() { static thread_local RenderCounts* p = nullptr; return p; }
inline Hit traceRayClosest(const Scene& S, V3 ori, V3 dir, int kind = 0) {   // traceRay(const Ray&) TraceHelper.h:31-37
    RenderCounts* rc = renderCounts();
    if (rc) rc->rays[kind]++;
    Hit h; traceRay(S, ori, dir, S.d.ray_trace_eps, FLT_MAX, false, 0.0f, h, rc ? &rc->c[kind] : nullptr); if (!h.hasHit()) h.dist = FLT_MAX; return h;
}
// Engine/KernelDynamicScene.cu:70-80
inline bool occluded(const Scene& S, V3 o, V3 d, float tmin, float tmax) {
    Hit r2 = traceRayClosest(S, o, d, 1);
    bool end = r2.dist < tmax - S.d.ray_trace_eps;
    if (std::isinf(tmax) && !r2.hasHit()) end = false;
    return r2.dist > tmin + S.d.ray_trace_eps && end;
}

// --------------------------------------------------------------------------- TriangleData / fillDG
struct DG {   // Engine/DifferentialGeometry.h:11-47
    V3 P; Frame sys; V3 n; V3 dpdu, dpdv; V2 uv; V2 bary; uint8_t extraData;
    const ctl_mipmap* images = nullptr;   // g_SceneData.m_sTexData (ImageTexture::getTexture, Texture.cu:39-42)
    const ctl_rough_transmittance* rough_transmittance = nullptr;   // RoughTransmittanceManager's three tables (RoughTransmittance.cu:121-131)
    const ctl_material* materials = nullptr;   // g_SceneData.m_sMatData: nested BSDFs of coating / roughcoating / blend are entries of it
    // ray differentials of the first hit (DifferentialGeometry::computePartials; PathTracer.cu:60-61 — the wavefront tracer never computes them)
    bool hasUVPartials = false; float dudx = 0, dudy = 0, dvdx = 0, dvdy = 0;
    const struct MipPyramid* pyramids = nullptr;   // one per image: the levels behind level 0 (KernelMIPMap::m_sOffsets)
};
// Engine/TriangleData.cu:22-32 + 34-65
inline void triDataSetUV(ctl_triangle_data& T, V2 a, V2 b, V2 c) {
    auto pk = [](V2 v) { return (uint32_t)floatToHalf(v.x) | ((uint32_t)floatToHalf(v.y) << 16); };   // ushort2 {x,y} little endian
    T.uv[0] = pk(a); T.uv[1] = pk(b); T.uv[2] = pk(c);
}
inline void triDataSetData(ctl_triangle_data& T, V3 v0, V3 v1, V3 v2, V3 n0, V3 n1, V3 n2, bool quirk = false) {
    auto h = [&](uint32_t bits) { return halfToFloat((uint16_t)bits, quirk); };
    V2 t0{ h(T.uv[0]), h(T.uv[0] >> 16) }, t1{ h(T.uv[1]), h(T.uv[1] >> 16) }, t2{ h(T.uv[2]), h(T.uv[2] >> 16) };
    V3 dP1 = v1 - v0, dP2 = v2 - v0;
    V2 dUV1{ t1.x - t0.x, t1.y - t0.y }, dUV2{ t2.x - t0.x, t2.y - t0.y };
    float determinant = dUV1.x * dUV2.y - dUV1.y * dUV2.x;
    V3 dpdu, dpdv;
    if (determinant == 0) {
        V3 a, b, n = normalize(cross(dP1, dP2));
        coordinateSystem(n, a, b);
        dpdu = a; dpdv = b;
    } else {
        float invDet = 1.0f / determinant;
        dpdu = ((dUV2.y * dP1 - dUV1.y * dP2) * invDet);
        dpdv = ((-dUV2.x * dP1 + dUV1.x * dP2) * invDet);
    }
    uint32_t ax = floatToHalf(dpdu.x), ay = floatToHalf(dpdu.y), az = floatToHalf(dpdu.z);
    uint32_t bx = floatToHalf(dpdv.x), by = floatToHalf(dpdv.y), bz = floatToHalf(dpdv.z);
    T.nor_mat_extra[0] = (uint32_t)normalToUchar2(n0) | ((uint32_t)normalToUchar2(n1) << 16);
    T.nor_mat_extra[1] = (uint32_t)normalToUchar2(n2) | (T.nor_mat_extra[1] & 0xffff0000);
    T.dpdu_dpdv[0] = ax | (ay << 16); T.dpdu_dpdv[1] = az | (bx << 16); T.dpdu_dpdv[2] = by | (bz << 16);
}
inline uint32_t triMatIndex(const ctl_triangle_data& T, uint32_t off) { return ((T.nor_mat_extra[1] >> 16) & 0xff) + off; }   // TriangleData.h:40-44
// Engine/TriangleData.cu:75-103
inline void triDataFillDG(const ctl_triangle_data& T, const M44& localToWorld, DG& dg, bool quirk) {
    auto h = [&](uint32_t bits) { return halfToFloat((uint16_t)bits, quirk); };
    V3 na = uchar2ToNormal((uint16_t)T.nor_mat_extra[0]), nb = uchar2ToNormal((uint16_t)(T.nor_mat_extra[0] >> 16)), nc = uchar2ToNormal((uint16_t)T.nor_mat_extra[1]);
    float w = 1.0f - dg.bary.x - dg.bary.y, u = dg.bary.x, v = dg.bary.y;
    V3 n = normalize(u * na + v * nb + w * nc);
    const uint32_t* dpd = T.dpdu_dpdv;
    V3 dpdu(h(dpd[0]), h(dpd[0] >> 16), h(dpd[1]));
    V3 dpdv(h(dpd[1] >> 16), h(dpd[2]), h(dpd[2] >> 16));
    V3 s = dpdu - n * dot(n, dpdu);
    V3 t = cross(s, n);
    s = transformDir(localToWorld, s); t = transformDir(localToWorld, t);
    dg.sys = Frame(normalize(s), normalize(t), normalize(cross(t, s)));
    dg.dpdu = transformDir(localToWorld, dpdu);
    dg.dpdv = transformDir(localToWorld, dpdv);
    dg.n = normalize(cross(dg.dpdu, dg.dpdv));
    V2 ta{ h(T.uv[0]), h(T.uv[0] >> 16) }, tb{ h(T.uv[1]), h(T.uv[1] >> 16) }, tc{ h(T.uv[2]), h(T.uv[2] >> 16) };
    dg.uv = V2{ u * ta.x + v * tb.x + w * tc.x, u * ta.y + v * tb.y + w * tc.y };
    dg.extraData = (uint8_t)(T.nor_mat_extra[1] >> 24);
    if (dot(dg.n, dg.sys.n) < 0.0f) dg.n = -dg.n;
}
// Kernel/TraceHelper.cu:274-307 (host branch)
inline void fillDG(const Scene& S, V2 bary, uint32_t triIdx, uint32_t nodeIdx, DG& dg) {
    M44 l2w; std::memcpy(l2w.d, S.d.node_transforms[nodeIdx].m, 64);
    dg.bary = bary;
    dg.images = S.d.images; dg.rough_transmittance = S.d.rough_transmittance; dg.materials = S.d.materials;
    triDataFillDG(S.d.tri_data[triIdx], l2w, dg, S.half_host_quirk);
}

// --------------------------------------------------------------------------- KernelMIPMap, level 0 (Engine/MIPMap.cu, MIPMap_device.h)
inline Spec texelDecode(uint32_t v, uint32_t type) {
    unsigned x = v & 0xff, y = (v >> 8) & 0xff, z = (v >> 16) & 0xff, w = v >> 24;
    if (type == CTL_TEXEL_RGBE) {   // SpectrumConverter::RGBEToFloat3 (Math/Spectrum.h:557-565)
        if (w) { float e = ldexpf(1.0f, int(w) - (128 + 8)); return Spec(x * e, y * e, z * e); }
        return Spec(0.0f);
    }
    return Spec(float(x) / 255.0f, float(y) / 255.0f, float(z) / 255.0f);   // COLORREFToFloat3 (:528-532)
}
// MIPMap_device.h:34-55 (the MIRROR branch tests the parity of uv.x for both axes, as the reference does)
inline bool wrapCoordinates(V2 uv, V2 dim, uint32_t w, V2& loc) {
    switch (w) {
    case CTL_WRAP_REPEAT: loc = V2{ fracf(uv.x) * dim.x, fracf(1.0f - uv.y) * dim.y }; return true;
    case CTL_WRAP_CLAMP: loc = V2{ clampf(uv.x, 0.0f, 1.0f) * dim.x, clampf(1.0f - uv.y, 0.0f, 1.0f) * dim.y }; return true;
    case CTL_WRAP_MIRROR:
        loc.x = (int)uv.x % 2 == 0 ? fracf(uv.x) : 1.0f - fracf(uv.x);
        loc.y = (int)uv.x % 2 == 0 ? fracf(uv.y) : 1.0f - fracf(uv.y);
        loc = V2{ loc.x * dim.x, loc.y * dim.y }; return true;
    case CTL_WRAP_BLACK:
        if (uv.x < 0 || uv.x >= 1 || uv.y < 0 || uv.y >= 1) return false;
        loc = V2{ uv.x * dim.x, uv.y * dim.y }; return true;
    }
    return false;
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// KernelMIPMap::Texel(0, uv) (MIPMap.cu:21-44)
inline Spec mipTexel(const ctl_mipmap& M, V2 uv) {
    V2 l;
    if (!wrapCoordinates(uv, V2{ (float)M.width, (float)M.height }, M.wrap_mode, l)) return Spec(0.0f);
    int x = clampi((int)l.x, 0, (int)M.width - 1), y = clampi((int)l.y, 0, (int)M.height - 1);
    return texelDecode(M.texels[(size_t)y * M.width + x], M.texel_type);
}
// KernelMIPMap::triangle(0, uv) (MIPMap.cu:46-57)
inline Spec mipTriangle(const ctl_mipmap& M, V2 uv) {
    V2 s{ (float)M.width, (float)M.height }, is{ 1.0f / s.x, 1.0f / s.y };
    V2 l{ uv.x * s.x, uv.y * s.y };
    float ds = fracf(l.x), dt = fracf(l.y);
    return ((1.f - ds) * (1.f - dt)) * mipTexel(M, uv) +
           ((1.f - ds) * dt) * mipTexel(M, V2{ uv.x + 0, uv.y + is.y }) +
           (ds * (1.f - dt)) * mipTexel(M, V2{ uv.x + is.x, uv.y + 0 }) +
           (ds * dt) * mipTexel(M, V2{ uv.x + is.x, uv.y + is.y });
}
// KernelMIPMap::Sample(uv) (MIPMap.cu:116-121)
inline Spec mipSample(const ctl_mipmap& M, V2 uv) { return M.filter_mode == CTL_FILTER_POINT ? mipTexel(M, uv) : mipTriangle(M, uv); }
// KernelMIPMap::Sample(0.0f, x, y) (MIPMap.cu:155-172): width 0 -> level 0, clamped direct fetch
inline Spec mipFetch(const ctl_mipmap& M, int x, int y) {
    x = clampi(x, 0, (int)M.width - 1); y = clampi(y, 0, (int)M.height - 1);
    return texelDecode(M.texels[(size_t)y * M.width + x], M.texel_type);
}

// ---- the levels behind level 0 and the filtered lookup of a first hit with ray differentials
// SpectrumConverter::Float3ToCOLORREF / Float3ToRGBE (Math/Spectrum.h:521-555)
inline uint32_t float3ToRGBCOL(Spec c) {
    auto q = [](float x) { return (uint32_t)(uint8_t)(clampf(x, 0.0f, 1.0f) * 255.0f); };
    return q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (255u << 24);
}
inline uint32_t float3ToRGBE(Spec c) {
    float mx = fmax2(c.x, fmax2(c.y, c.z));
    if (mx < 1e-32) return 0;
    int e; mx = (float)std::frexp((double)mx, &e) * 256.0f / mx;
    return (uint32_t)(uint8_t)(c.x * mx) | ((uint32_t)(uint8_t)(c.y * mx) << 8) | ((uint32_t)(uint8_t)(c.z * mx) << 16) | ((uint32_t)(uint8_t)(e + 128) << 24);
}
// MIPMap::CompileToBinary (Engine/MIPMap.cpp:41-95): nLevels = 1 + log2(min(w, h)); level i = the 2x2 box average of level i-1, decoded, averaged and re-encoded
struct MipPyramid {
    std::vector<uint32_t> texels; uint32_t levels = 1; uint32_t offsets[16] = {};
    void build(const ctl_mipmap& M) {
        texels.assign(M.texels, M.texels + (size_t)M.width * M.height);
        uint32_t mn = M.width < M.height ? M.width : M.height; levels = 1; while ((mn >>= 1) && levels < 16) levels++;
        offsets[0] = 0;
        uint32_t off = M.width * M.height, pw = M.width;
        size_t prev = 0;
        for (uint32_t i = 1, j = M.width / 2, k = M.height / 2; i < levels; i++, j >>= 1, k >>= 1) {
            offsets[i] = off; texels.resize((size_t)off + (size_t)j * k);
            for (uint32_t t = 0; t < k; t++) for (uint32_t x = 0; x < j; x++) {
                auto ld = [&](uint32_t xx, uint32_t yy) { return texelDecode(texels[prev + (size_t)yy * pw + xx], M.texel_type); };
                Spec v = 0.25f * (ld(2 * x, 2 * t) + ld(2 * x + 1, 2 * t) + ld(2 * x, 2 * t + 1) + ld(2 * x + 1, 2 * t + 1));
                texels[(size_t)off + (size_t)t * j + x] = M.texel_type == CTL_TEXEL_RGBE ? float3ToRGBE(v) : float3ToRGBCOL(v);
            }
            prev = off; pw = j; off += j * k;
        }
    }
};
inline const float* mipWeightLut() {   // MIPMap.cpp:87-92, MTS_MIPMAP_LUT_SIZE = 64
    static float lut[64]; static bool init = false;
    if (!init) { for (int i = 0; i < 64; i++) { float r2 = (float)i / (float)(64 - 1); lut[i] = expf(-2.0f * r2) - expf(-2.0f); } init = true; }   // a table built on the host (libm in both oracle builds)
    return lut;
}
// KernelMIPMap::Texel(level, uv) (MIPMap.cu:21-44)
inline Spec mipTexelL(const ctl_mipmap& M, const MipPyramid& P, uint32_t level, V2 uv) {
    const int wl = (int)(M.width >> level), hl = (int)(M.height >> level);
    V2 l;
    if (!wrapCoordinates(uv, V2{ (float)wl, (float)hl }, M.wrap_mode, l)) return Spec(0.0f);
    int x = clampi((int)l.x, 0, wl - 1), y = clampi((int)l.y, 0, hl - 1);
    return texelDecode(P.texels[(size_t)P.offsets[level] + (size_t)y * wl + x], M.texel_type);
}
// KernelMIPMap::triangle(level, uv) (MIPMap.cu:46-57)
inline Spec mipTriangleL(const ctl_mipmap& M, const MipPyramid& P, uint32_t level, V2 uv) {
    level = level > P.levels - 1 ? P.levels - 1 : level;
    V2 s{ (float)(M.width >> level), (float)(M.height >> level) }, is{ 1.0f / s.x, 1.0f / s.y };
    float ds = fracf(uv.x * s.x), dt = fracf(uv.y * s.y);
    return ((1.f - ds) * (1.f - dt)) * mipTexelL(M, P, level, uv) + ((1.f - ds) * dt) * mipTexelL(M, P, level, V2{ uv.x + 0, uv.y + is.y }) +
           (ds * (1.f - dt)) * mipTexelL(M, P, level, V2{ uv.x + is.x, uv.y + 0 }) + (ds * dt) * mipTexelL(M, P, level, V2{ uv.x + is.x, uv.y + is.y });
}
// math::log2 on the host (Math/MathFunc.h:258-265): mlog(a) / mlog(2) — not log2f, which eval() calls by name in its anisotropic branch (MIPMap.cu:231, :266)
inline float mathLog2(float a) { return mlog(a) / mlog(2.0f); }
// KernelMIPMap::Sample(uv, width) (MIPMap.cu:140-153): the pyramid level from a footprint width, trilinear between two levels
inline Spec mipSampleWidth(const ctl_mipmap& M, const MipPyramid& P, V2 uv, float width) {
    const float level = (float)(P.levels - 1) + mathLog2(fmax2(width, 1e-8f));
    if (level < 0) return mipTriangleL(M, P, 0, uv);
    if (level >= (float)(P.levels - 1)) return mipTexelL(M, P, P.levels - 1, uv);
    const int iLevel = (int)floorf(level); const float delta = level - iLevel;
    return (1.f - delta) * mipTriangleL(M, P, (uint32_t)iLevel, uv) + delta * mipTriangleL(M, P, (uint32_t)(iLevel + 1), uv);
}
// KernelMIPMap::Sample(width, x, y) (MIPMap.cu:155-172): clamped direct fetch from the level the width selects
inline Spec mipFetchL(const ctl_mipmap& M, const MipPyramid& P, float width, int x, int y) {
    const float l = (float)(P.levels - 1) + mathLog2(fmax2(width, 1e-8f));
    const int level = (int)clampf(l, 0.0f, (float)(P.levels - 1)), wl = (int)(M.width >> level), hl = (int)(M.height >> level);
    x = clampi(x, 0, wl - 1); y = clampi(y, 0, hl - 1);
    return texelDecode(P.texels[(size_t)P.offsets[level] + (size_t)y * wl + x], M.texel_type);
}
// KernelMIPMap::evalEWA (MIPMap.cu:59-114)
inline Spec mipEvalEWA(const ctl_mipmap& M, const MipPyramid& P, uint32_t level, V2 uv, float A, float B, float C) {
    if (level >= P.levels) return mipTexelL(M, P, P.levels - 1, V2{ 0, 0 });
    V2 size{ (float)(M.width >> level), (float)(M.height >> level) };
    float u = uv.x * size.x - 0.5f, v = uv.y * size.y - 0.5f;
    V2 ratio{ size.x / (float)M.width, size.y / (float)M.height };
    A /= ratio.x * ratio.x; B /= ratio.x * ratio.y; C /= ratio.y * ratio.y;
    float invDet = 1.0f / (-B * B + 4.0f * A * C), deltaU = 2.0f * sqrtf(C * invDet), deltaV = 2.0f * sqrtf(A * invDet);
    int u0 = (int)ceilf(u - deltaU), u1 = (int)floorf(u + deltaU), v0 = (int)ceilf(v - deltaV), v1 = (int)floorf(v + deltaV);
    float As = A * 64, Bs = B * 64, Cs = C * 64;
    Spec result(0.0f); float denominator = 0.0f, ddq = 2 * As, uu0 = u0 - u;
    const float* lut = mipWeightLut();
    for (int vt = v0; vt <= v1; ++vt) {
        const float vv = vt - v;
        float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv, dq = As * (2 * uu0 + 1) + Bs * vv;
        for (int ut = u0; ut <= u1; ++ut) {
            if (q < 64) { unsigned qi = (unsigned)q; if (qi < 64) { const float w = lut[(int)q]; result = result + mipTexelL(M, P, level, V2{ (float)ut / size.x, (float)vt / size.y }) * w; denominator += w; } }
            q += dq; dq += ddq;
        }
    }
    if (denominator == 0) return mipTriangleL(M, P, level, uv);
    return sdiv(result, denominator);
}
// KernelMIPMap::eval(uv, d0, d1) (MIPMap.cu:193-278)
inline Spec mipEval(const ctl_mipmap& M, const MipPyramid& P, V2 uv, V2 d0, V2 d1) {
    const float dimx = (float)M.width, dimy = (float)M.height;
    float du0 = d0.x * dimx, dv0 = d0.y * dimy, du1 = d1.x * dimx, dv1 = d1.y * dimy, du = (du0 + du1) / 2.0f, dv = (dv0 + dv1) / 2.0f;
    if (M.filter_mode == CTL_FILTER_POINT) return mipTexelL(M, P, 0, uv);
    if (M.filter_mode == CTL_FILTER_BILINEAR) return mipTriangleL(M, P, 0, uv);
    if (M.filter_mode == CTL_FILTER_TRILINEAR) {
        float levela = mathLog2(dimx / fabsf(du)), levelb = mathLog2(dimy / fabsf(dv)), level = (float)P.levels - clampf((levela + levelb) / 2.0f, 1.0f, (float)P.levels);
        int iLevel = (int)floorf(level), iLevel2 = clampi(iLevel + 1, 0, (int)P.levels - 1);
        float p = level - iLevel;
        return p * mipTriangleL(M, P, (uint32_t)iLevel, uv) + (1 - p) * mipTriangleL(M, P, (uint32_t)iLevel2, uv);
    }
    float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
    float root = sqrtf((A - C) * (A - C) + B * B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root);
    float majorRadius = Aprime != 0 ? sqrtf(F / Aprime) : 0, minorRadius = Cprime != 0 ? sqrtf(F / Cprime) : 0;
    if (!(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        float level = mlog2(fmax2(majorRadius, 1e-4f)); int ilevel = (int)floorf(level);
        if (ilevel < 0) return mipTriangleL(M, P, 0, uv);
        float a = level - ilevel;
        return mipTriangleL(M, P, (uint32_t)ilevel, uv) * (1.0f - a) + mipTriangleL(M, P, (uint32_t)(ilevel + 1), uv) * a;
    }
    const float maxAnisotropy = 16;
    if (minorRadius * maxAnisotropy < majorRadius) {
        minorRadius = majorRadius / maxAnisotropy;
        float theta = 0.5f * matan(B / (A - C)), sinTheta = msin(theta), cosTheta = mcos(theta);
        float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta, sin2Theta = 2 * sinTheta * cosTheta;
        A = a2 * cosTheta2 + b2 * sinTheta2; B = (a2 - b2) * sin2Theta; C = a2 * sinTheta2 + b2 * cosTheta2; F = a2 * b2;
    }
    float scale = 1.0f / F; A *= scale; B *= scale; C *= scale;
    float level = fmax2(0.0f, mlog2(minorRadius)); int ilevel = (int)level; float a = level - ilevel;
    if (majorRadius < 1 || !(A > 0 && C > 0)) return mipTriangleL(M, P, (uint32_t)ilevel, uv);
    return mipEvalEWA(M, P, (uint32_t)ilevel, uv, A, B, C) * (1.0f - a) + mipEvalEWA(M, P, (uint32_t)(ilevel + 1), uv, A, B, C) * a;
}

// --------------------------------------------------------------------------- textures (SceneTypes/Texture.h: constant, checkerboard, image)
inline Spec texEval(const ctl_texture& t, const DG& dg) {
    if (t.type == CTL_TEX_CHECKER) {
        float u = dg.uv.x * t.uv_scale[0] + t.uv_offset[0], v = dg.uv.y * t.uv_scale[1] + t.uv_offset[1];
        auto modulo = [](int a, int b) { int r = a % b; return (r < 0) ? r + b : r; };   // MathFunc.h:120-124
        int x = 2 * modulo((int)(u * 2), 2) - 1, y = 2 * modulo((int)(v * 2), 2) - 1;      // Texture.h:136-146
        return (x * y == 1) ? Spec(t.value[0], t.value[1], t.value[2]) : Spec(t.value1[0], t.value1[1], t.value1[2]);
    }
    if (t.type == CTL_TEX_IMAGE) {   // ImageTexture::Evaluate(dg) without uv partials -> Evaluate(uv) (Texture.cu:6-29)
        if (t.image == 0xffffffffu || dg.images == nullptr) return Spec(0.0f);
        V2 uv{ t.uv_scale[0] * dg.uv.x + 0 * dg.uv.y + t.uv_offset[0], 0 * dg.uv.x + t.uv_scale[1] * dg.uv.y + t.uv_offset[1] };   // TextureMapping2D::TransformPoint (Texture.h:34-41)
        if (dg.hasUVPartials && dg.pyramids) {   // Texture.cu:15-29: mapping.differentiate (Texture.h:52-59, m12 = m21 = 0) -> KernelMIPMap::eval
            const float dsdx = t.uv_scale[0] * dg.dudx + 0 * dg.dvdx, dsdy = t.uv_scale[0] * dg.dudy + 0 * dg.dvdy;
            const float dtdx = 0 * dg.dudx + t.uv_scale[1] * dg.dvdx, dtdy = 0 * dg.dudy + t.uv_scale[1] * dg.dvdy;
            return mipEval(dg.images[t.image], dg.pyramids[t.image], uv, V2{ dsdx, dtdx }, V2{ dsdy, dtdy }) * Spec(t.value[0], t.value[1], t.value[2]);
        }
        return mipSample(dg.images[t.image], uv) * Spec(t.value[0], t.value[1], t.value[2]);
    }
    return Spec(t.value[0], t.value[1], t.value[2]);
}

inline float luminance(Spec s) { return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f; }   // Spectrum.cu:174-177
// KernelMIPMap::SampleAlpha (MIPMap.cu:123-138); the reference indexes the texel without clamping, the restatement clamps
inline float mipSampleAlpha(const ctl_mipmap& M, V2 uv) {
    V2 l;
    if (!wrapCoordinates(uv, V2{ (float)M.width, (float)M.height }, M.wrap_mode, l)) return 0.0f;
    if (M.texel_type == CTL_TEXEL_RGBE) return 1.0f;
    int x = clampi((int)l.x, 0, (int)M.width - 1), y = clampi((int)l.y, 0, (int)M.height - 1);
    return float(M.texels[(size_t)y * M.width + x] >> 24) / 255.0f;
}
// KernelMIPMap::evalGradient (MIPMap.cu:174-191)
inline void mipEvalGradient(const ctl_mipmap& M, V2 uv, Spec grad[2]) {
    const V2 dim{ (float)M.width, (float)M.height };
    float u = uv.x * dim.x - 0.5f, v = uv.y * dim.y - 0.5f;
    int xPos = (int)u, yPos = (int)v;   // math::Float2Int
    float dx = u - xPos, dy = v - yPos;
    const Spec p00 = mipTexel(M, V2{ (float)xPos / dim.x, (float)yPos / dim.y });
    const Spec p10 = mipTexel(M, V2{ ((float)xPos + 1) / dim.x, (float)yPos / dim.y });
    const Spec p01 = mipTexel(M, V2{ (float)xPos / dim.x, ((float)yPos + 1) / dim.y });
    const Spec p11 = mipTexel(M, V2{ ((float)xPos + 1) / dim.x, ((float)yPos + 1) / dim.y });
    Spec tmp = p01 + p10 - p11;
    grad[0] = (p10 + p00 * (dy - 1) - tmp * dy) * dim.x;
    grad[1] = (p01 + p00 * (dx - 1) - tmp * dx) * dim.y;
}
inline V2 texMapPoint(const ctl_texture& t, V2 uv) { return V2{ t.uv_scale[0] * uv.x + 0 * uv.y + t.uv_offset[0], 0 * uv.x + t.uv_scale[1] * uv.y + t.uv_offset[1] }; }   // TextureMapping2D::TransformPoint

// Material::SampleNormalMap (Engine/Material.cu:96-138; enableParallaxOcclusion is never set by the reference)
inline bool sampleNormalMap(const ctl_material& mat, DG& dg) {
    if (mat.map_kind == CTL_MAP_NORMAL) {
        Spec c = texEval(mat.map_tex, dg);
        V3 n = c;   // toLinearRGB of an RGB spectrum
        V3 nWorld = normalize(dg.sys.toWorld(n - V3(0.5f)));
        dg.sys.n = nWorld;
        dg.sys.t = normalize(cross(nWorld, dg.sys.s));
        dg.sys.s = normalize(cross(nWorld, dg.sys.t));
        return true;
    }
    if (mat.map_kind == CTL_MAP_HEIGHT && mat.map_tex.type == CTL_TEX_IMAGE && mat.map_tex.image != 0xffffffffu && dg.images) {
        V2 uv = texMapPoint(mat.map_tex, dg.uv);
        Spec grad[2];
        mipEvalGradient(dg.images[mat.map_tex.image], uv, grad);
        float dDispDu = luminance(grad[0]), dDispDv = luminance(grad[1]);
        V3 dpdu = dg.dpdu + dg.sys.n * (dDispDu - dot(dg.sys.n, dg.dpdu));
        V3 dpdv = dg.dpdv + dg.sys.n * (dDispDv - dot(dg.sys.n, dg.dpdv));
        dg.sys.n = normalize(cross(dpdu, dpdv));
        dg.sys.s = normalize(dpdu - dg.sys.n * dot(dg.sys.n, dpdu));
        dg.sys.t = normalize(cross(dg.sys.n, dg.sys.s));
        if (dot(dg.sys.n, dg.n) < 0) dg.sys.n = -dg.sys.n;
        return true;
    }
    return false;
}

// Material::AlphaTest (Engine/Material.cu:160-190); sample_fast (:141-158) = texEval at the interpolated uv
inline bool materialAlphaTest(const ctl_material& mat, V2 uv, const ctl_mipmap* images) {
    const uint32_t st = mat.alpha_state;
    if (st == CTL_ALPHA_DISABLED) return true;
    const ctl_texture& refl = mat.tex[0];   // bsdf.getTexture(0)
    const bool refl_img = refl.type == CTL_TEX_IMAGE, alpha_img = mat.alpha_tex.type == CTL_TEX_IMAGE;
    if ((st == CTL_ALPHA_MAP_ALPHA && alpha_img) || (st == CTL_ALPHA_REFLECTANCE_ALPHA && refl_img)) {
        const ctl_texture& t = st == CTL_ALPHA_MAP_ALPHA ? mat.alpha_tex : refl;
        float alpha = mipSampleAlpha(images[t.image], texMapPoint(t, uv));
        return alpha >= mat.alpha_test_scalar;
    }
    DG dg; dg.uv = uv; dg.images = images;
    Spec val = texEval((st & 4) ? refl : mat.alpha_tex, dg);
    if ((st & 3) == 1) return luminance(val) >= mat.alpha_test_scalar;
    if ((st & 3) == 3) {
        Spec d = val - Spec(mat.alpha_test_color[0], mat.alpha_test_color[1], mat.alpha_test_color[2]);
        return fmax2(fmax2(fabsf(d.x), fabsf(d.y)), fabsf(d.z)) <= mat.alpha_test_scalar;
    }
    return true;
}
inline bool alphaSurvive(const Scene& S, uint32_t tri, uint32_t nodeIdx, float u, float v) {
    const ctl_triangle_data& T = S.d.tri_data[tri];
    const ctl_material& mat = S.d.materials[triMatIndex(T, S.d.nodes[nodeIdx].material_offset)];
    if (mat.alpha_state == CTL_ALPHA_DISABLED) return true;
    auto h = [&](uint32_t bits) { return halfToFloat((uint16_t)bits, S.half_host_quirk); };
    // tri->getUVSetData(0, a, b, c) (Kernel/TraceHelper.cu:149): u from the HIGH half of each word, v from the low one — see shapeTriUV below: the alpha map, too, is looked
    // up with the surface's u and v exchanged
    V2 a{ h(T.uv[0] >> 16), h(T.uv[0]) }, b{ h(T.uv[1] >> 16), h(T.uv[1]) }, c{ h(T.uv[2] >> 16), h(T.uv[2]) };
    V2 uv{ u * a.x + v * b.x + (1 - u - v) * c.x, u * a.y + v * b.y + (1 - u - v) * c.y };
    return materialAlphaTest(mat, uv, S.d.images);
}

// --------------------------------------------------------------------------- sampler (Kernel/Sampler_device.h:59-113)
struct Sampler {
    const float* t1; const float* t2; unsigned idx; unsigned d1 = 0, d2 = 0;
    Sampler(const float* a, const float* b, unsigned i) : t1(a), t2(b), idx(i) {}
    float randomFloat() {
        unsigned e = d1 % CTL_SAMPLER_SEQUENCE_LENGTH, f = idx; float val = 0.0f;
        for (int i = 0; i < 2; i++) { val += t1[e * CTL_SAMPLER_NUM_SEQUENCES + f % CTL_SAMPLER_NUM_SEQUENCES]; f /= CTL_SAMPLER_NUM_SEQUENCES; }
        d1++; return fracf(val);
    }
    V2 randomFloat2() {
        unsigned e = d2 % CTL_SAMPLER_SEQUENCE_LENGTH, f = idx; float vx = 0.0f, vy = 0.0f;
        for (int i = 0; i < 2; i++) { const float* p = t2 + 2 * (e * CTL_SAMPLER_NUM_SEQUENCES + f % CTL_SAMPLER_NUM_SEQUENCES); vx += p[0]; vy += p[1]; f /= CTL_SAMPLER_NUM_SEQUENCES; }
        d2++; return V2{ fracf(vx), fracf(vy) };
    }
    void skip(unsigned off) { d1 += off; d2 += off; }
};

// --------------------------------------------------------------------------- sensor (SceneTypes/Sensor.cu:76-128)
// PerspectiveSensor / ThinLensSensor / OrthographicSensor / TelecentricSensor (SceneTypes/Sensor.cu: Update :76-96, :226-246, :408-427, :515-535)
struct SensorO {
    uint32_t type = CTL_SENSOR_PERSPECTIVE;
    M44 toWorld, sampleToCamera; V2 invRes; V3 dx, dy; float apertureRadius = 0, focusDistance = 0, screenScaleX = 1;
    void update(const ctl_sensor& s) {
        type = s.type;
        if (type < CTL_SENSOR_SPHERICAL || type > CTL_SENSOR_TELECENTRIC) throw std::runtime_error("oracle: unknown sensor type");
        std::memcpy(toWorld.d, s.to_world, 64);
        float aspect = s.resolution[0] / s.resolution[1];
        invRes = V2{ 1.0f / s.resolution[0], 1.0f / s.resolution[1] };
        if (type == CTL_SENSOR_SPHERICAL) return;
        const bool ortho = type == CTL_SENSOR_ORTHOGRAPHIC || type == CTL_SENSOR_TELECENTRIC;
        // float4x4::orthographic (float4x4.h:625-628) = Scale(1, 1, 1 / (far - near)) % Translate(0, 0, -near)
        M44 proj = ortho ? mul(scaleM(V3(1.0f, 1.0f, 1.0f / (s.far_depth - s.near_depth))), translateM(V3(0.0f, 0.0f, -s.near_depth))) : perspective(s.fov, s.near_depth, s.far_depth);
        M44 c2s = mul(mul(scaleM(V3(-0.5f, -0.5f * aspect, 1.0f)), translateM(V3(-1.0f, -1.0f / aspect, 0.0f))), proj);
        sampleToCamera = inverse(c2s);
        dx = transformPoint(sampleToCamera, V3(invRes.x, 0.0f, 0.0f)) - transformPoint(sampleToCamera, V3(0.0f));
        dy = transformPoint(sampleToCamera, V3(0.0f, invRes.y, 0.0f)) - transformPoint(sampleToCamera, V3(0.0f));
        apertureRadius = s.aperture_radius; focusDistance = s.focus_distance; screenScaleX = s.screen_scale[0] != 0.0f ? s.screen_scale[0] : 1.0f;
    }
    // sampleRayDifferential of the four types (Sensor.cu:130-144, :292-311, :440-450, :558-574); sampleRay (:116-128, :267-290, :429-438, :537-556) is its first ray,
    // except that OrthographicSensor::sampleRay starts on the plane z = 0 of the camera while its differential version starts at nearP
    void sampleRayDifferential(V2 pixelSample, V2 apertureSample, V3& o, V3& d, V3& oX, V3& dX, V3& oY, V3& dY, bool plain = false) const {
        if (type == CTL_SENSOR_SPHERICAL) {   // SphericalSensor::sampleRay (Sensor.cu:6-17); its sampleRayDifferential (Sensor.h:122-125) leaves rayX / rayY unset: the ray itself is used here
            float sinPhi = msin((1.0f - pixelSample.x * invRes.x) * 2 * PI), cosPhi = mcos((1.0f - pixelSample.x * invRes.x) * 2 * PI);
            float sinTheta = msin((1.0f - pixelSample.y * invRes.y) * PI), cosTheta = mcos((1.0f - pixelSample.y * invRes.y) * PI);
            o = transformPoint(toWorld, V3(0.0f)); d = transformDir(toWorld, V3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta));
            oX = oY = o; dX = dY = d;
            return;
        }
        V3 nearP = transformPoint(sampleToCamera, V3(pixelSample.x * invRes.x, pixelSample.y * invRes.y, 0.0f));
        if (type == CTL_SENSOR_PERSPECTIVE) {
            o = transformPoint(toWorld, V3(0.0f));   // toWorld.Translation() (float4x4.h:93-96)
            d = transformDir(toWorld, normalize(nearP));
            oX = oY = o; dX = transformDir(toWorld, normalize(nearP + dx)); dY = transformDir(toWorld, normalize(nearP + dy));
        } else if (type == CTL_SENSOR_THINLENS) {
            V2 tmp = squareToUniformDiskConcentric(apertureSample) * apertureRadius;
            V3 apertureP(tmp.x, tmp.y, 0.0f);
            float fDist = focusDistance / nearP.z;
            V3 focusP = nearP * fDist, focusPx = (nearP + dx) * fDist, focusPy = (nearP + dy) * fDist;
            o = transformPoint(toWorld, apertureP); d = transformDir(toWorld, normalize(focusP - apertureP));
            oX = oY = o; dX = transformDir(toWorld, normalize(focusPx - apertureP)); dY = transformDir(toWorld, normalize(focusPy - apertureP));
        } else if (type == CTL_SENSOR_ORTHOGRAPHIC) {
            o = transformPoint(toWorld, plain ? V3(nearP.x, nearP.y, 0.0f) : nearP); d = transformDir(toWorld, V3(0.0f, 0.0f, 1.0f));   // toWorld.Forward()
            oX = transformPoint(toWorld, nearP + dx); oY = transformPoint(toWorld, nearP + dy); dX = dY = d;
        } else {
            V2 diskSample = squareToUniformDiskConcentric(apertureSample) * (apertureRadius / screenScaleX);
            V3 focusP = nearP; focusP.z = focusDistance;
            V3 orig(diskSample.x + focusP.x, diskSample.y + focusP.y, 0.0f);
            o = transformPoint(toWorld, orig); d = normalize(transformDir(toWorld, focusP - orig));
            oX = transformPoint(toWorld, orig + dx); oY = transformPoint(toWorld, orig + dy); dX = dY = d;
        }
    }
    void sampleRay(V2 pixelSample, V2 apertureSample, V3& o, V3& d) const {
        if (type == CTL_SENSOR_THINLENS) {   // ThinLensSensor::sampleRay scales nearP by (focusDistance / nearP.z) in one product (Sensor.cu:279), the differential version through fDist
            V3 nearP = transformPoint(sampleToCamera, V3(pixelSample.x * invRes.x, pixelSample.y * invRes.y, 0.0f));
            V2 tmp = squareToUniformDiskConcentric(apertureSample) * apertureRadius;
            V3 apertureP(tmp.x, tmp.y, 0.0f), focusP = nearP * (focusDistance / nearP.z);
            o = transformPoint(toWorld, apertureP); d = transformDir(toWorld, normalize(focusP - apertureP));
            return;
        }
        V3 a, b, c, e; sampleRayDifferential(pixelSample, apertureSample, o, d, a, b, c, e, true);
    }
};
using PerspectiveSensor = SensorO;

// --------------------------------------------------------------------------- records (SceneTypes/Samples.h)
enum EMeasure { EInvalidMeasure = 0, ESolidAngle = 1, ELength = 2, EArea = 3, EDiscrete = 4 };
enum { EReflection = 0x2 | 0x20 | 0x80 | 0x8, EDiffuse = 0x2 | 0x4, EGlossy = 0x8 | 0x10, ESmooth = 0x2 | 0x4 | 0x8 | 0x10,
       EDelta = 0x1 | 0x20 | 0x40, EDelta1D = 0x80 | 0x100, EAll = ESmooth | EDelta | EDelta1D };
struct DirectRec {   // DirectSamplingRecord (Samples.h:133-149)
    V3 p, n; float pdf; int measure; V2 uv; V3 ref, refN, d; float dist;
    DirectRec() {}
    DirectRec(V3 p_, V3 n_) : p(p_), n(n_), measure(EArea), ref(p_), refN(n_) {}
};
struct BRec {   // BSDFSamplingRecord (Samples.h:173-190)
    DG dg; V3 wi, wo; float eta; unsigned typeMask, sampledType;
};

// --------------------------------------------------------------------------- lights
// Math/MonteCarlo.cu:7-14 (STL_lower_bound = first element not less than `sample`)
inline unsigned sampleReuse(const float* cdf, unsigned size, float& sample, float& pdf) {
    const float* entry = std::lower_bound(cdf, cdf + size + 1, sample);
    unsigned index = (unsigned)std::min(std::max(0, int(entry - cdf) - 1), int(size - 1));
    pdf = cdf[index + 1] - cdf[index];
    sample = (sample - cdf[index]) / pdf;
    return index;
}
// Engine/ShapeSet.cu:51-69
inline void shapeSamplePosition(const Scene& S, const ctl_light& L, DirectRec& pRec, V2 spatialSample) {
    const float* areaDistribution = (const float*)(S.d.anim + L.area_dist_index);
    const ctl_shape_tri* triangles = (const ctl_shape_tri*)(S.d.anim + L.triangles_index);
    float pdf; V2 sample = spatialSample;
    unsigned index = sampleReuse(areaDistribution, L.count, sample.y, pdf);
    const ctl_shape_tri& sn = triangles[index];
    V2 bary = squareToUniformTriangle(sample);
    V3 p0(sn.p[0][0], sn.p[0][1], sn.p[0][2]), p1(sn.p[1][0], sn.p[1][1], sn.p[1][2]), p2(sn.p[2][0], sn.p[2][1], sn.p[2][2]);
    pRec.p = bary.x * p0 + bary.y * p1 + (1.f - bary.x - bary.y) * p2;
    pRec.n = V3(sn.n[0], sn.n[1], sn.n[2]);
    pRec.pdf = 1.0f / L.sum_area;
    pRec.measure = EArea;
    pRec.uv = bary;
}
// getUV (Engine/ShapeSet.cu:25-31): TriangleData::getUVSetData(0, a, b, c), uv = u a + v b + w c
inline V2 shapeTriUV(const Scene& S, const ctl_shape_tri& sn, V2 bary) {
    const ctl_triangle_data& T = S.d.tri_data[sn.t_dat];
    auto h = [&](uint32_t bits) { return halfToFloat((uint16_t)bits, S.half_host_quirk); };
    // TriangleData::getUVSetData(0, a, b, c) (Engine/TriangleData.cu:25-32) takes u from the HIGH half of each word and v from the low one — the other way round than fillDG
    // (:91-97) reads them and than the constructor packs them: an area light's radiance texture (and an alpha map, alphaSurvive above) is looked up with the surface's u and v
    // exchanged.  The reference's own behaviour, pinned on its own code by tests/golden/scene_lights.npz (the checker panel); reproduced here and in csrc/shading.h / mipmap.h.
    V2 a{ h(T.uv[0] >> 16), h(T.uv[0]) }, b{ h(T.uv[1] >> 16), h(T.uv[1]) }, c{ h(T.uv[2] >> 16), h(T.uv[2]) };
    float u = bary.x, v = bary.y, w = 1 - u - v;
    return V2{ u * a.x + v * b.x + w * c.x, u * a.y + v * b.y + w * c.y };
}
// AlgebraHelper::Barycentric (Math/AlgebraHelper.h:46-59)
inline bool barycentric(V3 p, V3 a, V3 b, V3 c, float& u, float& v) {
    V3 v0 = b - a, v1 = c - a, v2 = p - a;
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    u = 1.0f - v - w;
    return 0 <= v && v <= 1 && 0 <= u && u <= 1 && 0 <= w && w <= 1;
}
// ShapeSet::getPosition (Engine/ShapeSet.cu:71-91): the first triangle of the set that contains the point
inline bool shapeGetPosition(const Scene& S, const ctl_light& L, V3 pos, V2* bary, V2* uv) {
    const ctl_shape_tri* triangles = (const ctl_shape_tri*)(S.d.anim + L.triangles_index);
    for (unsigned i = 0; i < L.count; i++) {
        const ctl_shape_tri& sn = triangles[i]; V2 b;
        if (barycentric(pos, V3(sn.p[0][0], sn.p[0][1], sn.p[0][2]), V3(sn.p[1][0], sn.p[1][1], sn.p[1][2]), V3(sn.p[2][0], sn.p[2][1], sn.p[2][2]), b.x, b.y)) {
            if (bary) *bary = b;
            if (uv) *uv = shapeTriUV(S, sn, b);
            return true;
        }
    }
    return false;
}
// ShapeSet::PdfTriangle (Engine/ShapeSet.cu:93-106)
inline float shapePdfTriangle(const Scene& S, const ctl_light& L, V3 pos) {
    const float* areaDistribution = (const float*)(S.d.anim + L.area_dist_index);
    const ctl_shape_tri* triangles = (const ctl_shape_tri*)(S.d.anim + L.triangles_index);
    for (unsigned i = 0; i < L.count; i++) {
        const ctl_shape_tri& sn = triangles[i]; V2 b;
        if (barycentric(pos, V3(sn.p[0][0], sn.p[0][1], sn.p[0][2]), V3(sn.p[1][0], sn.p[1][1], sn.p[1][2]), V3(sn.p[2][0], sn.p[2][1], sn.p[2][2]), b.x, b.y))
            return areaDistribution[i + 1] - areaDistribution[i];
    }
    return 0.0f;
}
inline bool lightNeedsUV(const ctl_light& L) { return L.rad_texture.type == CTL_TEX_CHECKER || L.rad_texture.type == CTL_TEX_IMAGE; }   // needsUVSample (Light.cu:50-53)
// m_rad_texture.Evaluate(dg) for a DifferentialGeometry that only carries P, bary and uv (Light.cu:72-80, :125-130)
inline Spec lightRadiance(const Scene& S, const ctl_light& L, V3 P, V2 bary, V2 uv) {
    if (!lightNeedsUV(L)) return Spec(L.radiance[0], L.radiance[1], L.radiance[2]);
    DG dg; dg.P = P; dg.bary = bary; dg.uv = uv; dg.images = S.d.images;
    return texEval(L.rad_texture, dg);
}
inline Frame lightFrame(const ctl_light& L) {   // Spot / Distant `Frame ToWorld`
    return Frame(V3(L.to_world[0], L.to_world[1], L.to_world[2]), V3(L.to_world[4], L.to_world[5], L.to_world[6]), V3(L.to_world[8], L.to_world[9], L.to_world[10]));
}
// SpotLight::falloffCurve (SceneTypes/Light.cu:327-336)
inline Spec spotFalloff(const ctl_light& L, V3 d) {
    const float cosTheta = Frame::cosTheta(d);
    if (cosTheta <= L.cos_cutoff_angle) return Spec(0.0f);
    if (cosTheta >= L.cos_beam_width) return Spec(1.0f);
    return Spec((L.cutoff_angle - macos(cosTheta)) * L.inv_transition_width);
}
inline float intervalToTent(float sample) {   // Math/Warp.h:13-27
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; } else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - std::sqrt(sample));
}
inline unsigned sampleReuse(const float* cdf, unsigned size, float& sample, float& pdf);
// InfiniteLight::internalSampleDirection (SceneTypes/Light.cu:420-463)
inline void envSampleDirection(const Scene& S, const ctl_light& L, V2 sample, V3& d, Spec& value, float& pdf) {
    const ctl_mipmap& map = S.d.images[L.env_image];
    const float* cdfRows = (const float*)(S.d.anim + L.cdf_rows_index), *cdfCols = (const float*)(S.d.anim + L.cdf_cols_index), *rowWeights = (const float*)(S.d.anim + L.row_weights_index);
    const float sizeX = (float)map.width, sizeY = (float)map.height;
    float qpdf;
    unsigned row = sampleReuse(cdfRows, (unsigned)sizeY, sample.y, qpdf),
             col = sampleReuse(cdfCols + row * (unsigned)(sizeX + 1), (unsigned)sizeX, sample.x, qpdf);
    V2 pos{ (float)col + intervalToTent(sample.x), (float)row + intervalToTent(sample.y) };
    int xPos = clampi(floor2int(pos.x), 0, (int)(sizeX - 1)), yPos = clampi(floor2int(pos.y), 0, (int)(sizeY - 1));
    float dx1 = pos.x - xPos, dx2 = 1.0f - dx1, dy1 = pos.y - yPos, dy2 = 1.0f - dy1;
    Spec value1 = mipFetch(map, xPos, yPos) * dx2 * dy2 + mipFetch(map, xPos + 1, yPos) * dx1 * dy2;
    Spec value2 = mipFetch(map, xPos, yPos + 1) * dx2 * dy1 + mipFetch(map, xPos + 1, yPos + 1) * dx1 * dy1;
    value = (value1 + value2) * Spec(L.env_scale[0], L.env_scale[1], L.env_scale[2]);
    pdf = (luminance(value1) * rowWeights[(int)clampf((float)yPos, 0.0f, sizeY - 1.0f)] +
           luminance(value2) * rowWeights[(int)clampf((float)(yPos + 1), 0.0f, sizeY - 1.0f)]) * L.normalization;
    const float pixX = 2 * PI / sizeX, pixY = PI / sizeY;   // m_pixelSize (Light.cpp:54)
    float sinPhi = msin(pixX * (pos.x + 0.5f)), cosPhi = mcos(pixX * (pos.x + 0.5f));
    float sinTheta = msin(pixY * (pos.y + 0.5f)), cosTheta = mcos(pixY * (pos.y + 0.5f));
    d = V3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= fmax2(fabsf(sinTheta), EPSILON);
}
inline V3 transformDirTranspose(const float* m, V3 d) {   // OrthogonalAffineMap::TransformDirectionTranspose (float4x4.h:424-427)
    return V3(dot(d, V3(m[0], m[4], m[8])), dot(d, V3(m[1], m[5], m[9])), dot(d, V3(m[2], m[6], m[10])));
}
// InfiniteLight::internalPdfDirection (SceneTypes/Light.cu:465-486)
inline float envPdfDirection(const Scene& S, const ctl_light& L, V3 d) {
    const ctl_mipmap& map = S.d.images[L.env_image];
    const float* rowWeights = (const float*)(S.d.anim + L.row_weights_index);
    const float sizeX = (float)map.width, sizeY = (float)map.height;
    V2 uv{ matan2(d.x, -d.z) * INV_TWOPI, safe_acos(d.y) * INV_PI };
    float u = uv.x * sizeX - 0.5f, v = uv.y * sizeY - 0.5f;
    int xPos = floor2int(u), yPos = floor2int(v);
    float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    Spec value1 = mipFetch(map, xPos, yPos) * dx2 * dy2 + mipFetch(map, xPos + 1, yPos) * dx1 * dy2;
    Spec value2 = mipFetch(map, xPos, yPos + 1) * dx2 * dy1 + mipFetch(map, xPos + 1, yPos + 1) * dx1 * dy1;
    float sinTheta = safe_sqrt(1 - d.y * d.y);
    return (luminance(value1) * rowWeights[clampi(yPos, 0, (int)sizeY - 1)] + luminance(value2) * rowWeights[clampi(yPos + 1, 0, (int)sizeY - 1)])
        * L.normalization / fmax2(fabsf(sinTheta), EPSILON);
}
// InfiniteLight::evalEnvironment(ray) (SceneTypes/Light.cu:488-501): Sample(uv, 0) = triangle(0, uv)
inline Spec envEval(const Scene& S, const ctl_light& L, V3 dir) {
    V3 v = transformDirTranspose(L.to_world, dir);
    V2 uv{ matan2(v.x, -v.z) * INV_TWOPI, safe_acos(v.y) * INV_PI };
    return mipTriangle(S.d.images[L.env_image], uv) * Spec(L.env_scale[0], L.env_scale[1], L.env_scale[2]);
}
// SceneTypes/Light.cu:83-137, :13-31 (point), :287-301 (spot), :224-245 (distant), :350-366 (infinite)
inline Spec lightSampleDirect(const Scene& S, const ctl_light& L, DirectRec& dRec, V2 sample) {
    if (L.type == CTL_LIGHT_POINT) {
        dRec.p = V3(L.position[0], L.position[1], L.position[2]);
        V3 dir = dRec.p - dRec.ref;
        dRec.dist = length(dir);
        float invDist = 1.0f / dRec.dist;
        dRec.d = dir * invDist; dRec.n = V3(0.0f); dRec.pdf = 1; dRec.measure = EDiscrete; dRec.uv = V2{ 0.5f, 0.5f };
        return Spec(L.radiance[0], L.radiance[1], L.radiance[2]) * (invDist * invDist);
    }
    if (L.type == CTL_LIGHT_SPOT) {   // SceneTypes/Light.cu:287-301
        dRec.p = V3(L.position[0], L.position[1], L.position[2]);
        V3 dir = dRec.p - dRec.ref;
        dRec.dist = length(dir);
        float invDist = 1.0f / dRec.dist;
        dRec.d = dir * invDist; dRec.n = V3(0.0f); dRec.pdf = 1; dRec.measure = EDiscrete; dRec.uv = V2{ 0.5f, 0.5f };
        return Spec(L.radiance[0], L.radiance[1], L.radiance[2]) * spotFalloff(L, lightFrame(L).toLocal(-dRec.d)) * (invDist * invDist);
    }
    if (L.type == CTL_LIGHT_DISTANT) {   // SceneTypes/Light.cu:224-245
        V3 d = lightFrame(L).toWorld(V3(0.0f, 0.0f, 1.0f));
        V3 diskCenter = d * L.bsphere_radius;
        float distance = dot(dRec.ref - diskCenter, d);
        if (distance < 0) return Spec(0.0f);
        dRec.p = dRec.ref - distance * d; dRec.d = -d; dRec.n = d; dRec.dist = distance;
        dRec.pdf = 1.0f; dRec.measure = EDiscrete;
        return Spec(L.radiance[0], L.radiance[1], L.radiance[2]);
    }
    if (L.type == CTL_LIGHT_INFINITE) {   // SceneTypes/Light.cu:350-366
        Spec value; V3 d; float pdf;
        envSampleDirection(S, L, sample, d, value, pdf);
        M44 wt; std::memcpy(wt.d, L.to_world, 64);
        d = transformDir(wt, d);
        dRec.pdf = pdf;
        dRec.p = V3(L.bsphere_center[0], L.bsphere_center[1], L.bsphere_center[2]) + d * L.bsphere_radius;
        dRec.n = -normalize(d);
        dRec.dist = L.bsphere_radius;
        dRec.d = normalize(d);
        dRec.measure = ESolidAngle;
        return sdiv(value, pdf);
    }
    V2 uv{ 0.0f, 0.0f }; float sc = 1;
    if (L.orthogonal) {   // the point of a random triangle's plane straight above / below the reference point (Light.cu:87-107)
        sc = PI;
        const float* areaDistribution = (const float*)(S.d.anim + L.area_dist_index);
        const ctl_shape_tri* triangles = (const ctl_shape_tri*)(S.d.anim + L.triangles_index);
        float sx = sample.x;
        const ctl_shape_tri& sn = triangles[sampleReuse(areaDistribution, L.count, sx, dRec.pdf)];   // ShapeSet::sampleTriangle (ShapeSet.cu:38-49)
        V3 p0(sn.p[0][0], sn.p[0][1], sn.p[0][2]), p1(sn.p[1][0], sn.p[1][1], sn.p[1][2]), p2(sn.p[2][0], sn.p[2][1], sn.p[2][2]);
        V3 n = normalize(cross(p1 - p0, p2 - p0));
        float lambda = dot(p0, n) - dot(dRec.ref, n);
        dRec.p = dRec.ref + lambda * n;
        if (!barycentric(dRec.p, p0, p1, p2, dRec.uv.x, dRec.uv.y)) { dRec.pdf = 0.0f; return Spec(0.0f); }
        dRec.n = n;
        dRec.pdf = 1.0f / float(L.count);
        dRec.measure = EArea;
        if (lightNeedsUV(L)) uv = shapeTriUV(S, sn, dRec.uv);
    } else {
        shapeSamplePosition(S, L, dRec, sample);
        if (lightNeedsUV(L)) {
            const ctl_shape_tri* triangles = (const ctl_shape_tri*)(S.d.anim + L.triangles_index);
            float pdfTri; V2 s2 = sample;
            uv = shapeTriUV(S, triangles[sampleReuse((const float*)(S.d.anim + L.area_dist_index), L.count, s2.y, pdfTri)], dRec.uv);
        }
    }
    V3 dir = dRec.p - dRec.ref;
    float distSquared = lenSqr(dir);
    dRec.dist = std::sqrt(distSquared);
    dRec.d = dir / dRec.dist;
    float dp = absdot(dRec.d, dRec.n);
    if (!L.orthogonal) {
        dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
        dRec.measure = ESolidAngle;
    } else dRec.measure = EDiscrete;
    if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0)
        return sdiv(lightRadiance(S, L, dRec.p, dRec.uv, uv), dRec.pdf) * sc;
    dRec.pdf = 0.0f;
    return Spec(0.0f);
}
// SceneTypes/Light.cu:139-159
inline float lightPdfDirect(const Scene& S, const ctl_light& L, const DirectRec& dRec) {
    if (L.type == CTL_LIGHT_POINT || L.type == CTL_LIGHT_SPOT || L.type == CTL_LIGHT_DISTANT) return dRec.measure == EDiscrete ? 1.0f : 0.0f;
    if (L.type == CTL_LIGHT_INFINITE) {   // SceneTypes/Light.cu:368-378
        float pdfSA = envPdfDirection(S, L, transformDirTranspose(L.to_world, dRec.d));
        if (dRec.measure == ESolidAngle) return pdfSA;
        else if (dRec.measure == EArea) return pdfSA * absdot(dRec.d, dRec.n) / (dRec.dist * dRec.dist);
        else return 0.0f;
    }
    if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0) {
        if (L.orthogonal) return dRec.measure == EDiscrete ? shapePdfTriangle(S, L, dRec.p) : 0.0f;
        float pdfPos = 1.0f / L.sum_area;
        if (dRec.measure == ESolidAngle) return pdfPos * (dRec.dist * dRec.dist) / absdot(dRec.d, dRec.n);
        else if (dRec.measure == EArea) return pdfPos;
        else return 0.0f;
    }
    return 0.0f;
}
// SceneTypes/Light.cu:67-81
inline Spec lightEval(const Scene& S, const ctl_light& L, V3 p, const Frame& sys, V3 d) {
    if (L.type != CTL_LIGHT_DIFFUSE) return Spec(0.0f);
    if (dot(sys.n, d) <= 0 || (L.orthogonal && dot(d, sys.n) < 1 - DeltaEpsilon)) return Spec(0.0f);
    V2 bary{ 0.0f, 0.0f }, uv{ 0.0f, 0.0f };
    if (lightNeedsUV(L)) shapeGetPosition(S, L, p, &bary, &uv);   // CTL_ASSERT in the reference: a point that is on no triangle keeps uv = 0
    return lightRadiance(S, L, p, bary, uv);
}
// Engine/KernelDynamicScene.cu:25-46
inline const ctl_light* sampleEmitter(const Scene& S, float& emPdf, V2& sample) {
    const ctl_scene_desc& g = S.d;
    if (g.num_lights == 0) return nullptr;
    unsigned idx = (unsigned)(std::upper_bound(g.light_cdf, g.light_cdf + g.num_lights, sample.x) - g.light_cdf);
    if (idx >= g.num_lights) idx = g.num_lights - 1;
    float fU = g.light_cdf[idx], fL = idx > 0 ? g.light_cdf[idx - 1] : 0.0f;
    sample.x = (sample.x - fL) / (fU - fL);
    emPdf = fU - fL;
    return g.lights + g.light_indices[idx];
}
inline float pdfEmitter(const Scene& S, const ctl_light* L) {
    unsigned idx = (unsigned)(L - S.d.lights);
    return S.d.light_cdf[idx] - (idx == 0 ? 0.0f : S.d.light_cdf[idx - 1]);
}

// --------------------------------------------------------------------------- microfacet (Engine/MicrofacetDistribution.{h,cu})
inline float hypot2(float a, float b) {   // Math/MathFunc.h:326-341
    float r;
    if (fabsf(a) > fabsf(b)) { r = b / a; r = fabsf(a) * std::sqrt(1.0f + r * r); }
    else if (b != 0.0f) { r = a / b; r = fabsf(b) * std::sqrt(1.0f + r * r); }
    else r = 0.0f;
    return r;
}
// math::erfinv / math::erf (Math/MathFunc.h:343-393): Giles' single-precision polynomial, A&S 7.1.26
inline float erfinvRef(float x) {
    float w = -mlog((1.0f - x) * (1.0f + x)), p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w; p = -4.39150654e-06f + p * w; p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w; p = -0.00417768164f + p * w; p = 0.246640727f + p * w; p = 1.50140941f + p * w;
    } else {
        w = std::sqrt(w) - 3;
        p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w; p = -0.00367342844f + p * w; p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w; p = 0.00943887047f + p * w; p = 1.00167406f + p * w; p = 2.83297682f + p * w;
    }
    return p * x;
}
inline float erfRef(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    const float sign = copysignf(1.0f, x);
    x = fabsf(x);
    const float t = 1.0f / (1.0f + p * x);
    const float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * mexp(-x * x);
    return sign * y;
}
struct Microfacet {
    int type; float alphaU, alphaV; bool sampleVis; float expU = 0, expV = 0;
    Microfacet(int t, float aU, float aV, bool sv) : type(t), alphaU(fmax2(aU, 1e-4f)), alphaV(fmax2(aV, 1e-4f)), sampleVis(sv) {
        if (type == CTL_MF_PHONG) { expU = fmax2(2.0f / (alphaU * alphaU) - 2.0f, 0.0f); expV = fmax2(2.0f / (alphaV * alphaV) - 2.0f, 0.0f); }
    }
    void scaleAlpha(float value) {   // MicrofacetDistribution.h:61-67: the Phong exponents follow the scaled roughness
        alphaU *= value; alphaV *= value;
        if (type == CTL_MF_PHONG) { expU = fmax2(2.0f / (alphaU * alphaU) - 2.0f, 0.0f); expV = fmax2(2.0f / (alphaV * alphaV) - 2.0f, 0.0f); }
    }
    bool isIso() const { return alphaU == alphaV; }
    float interpPhongExp(V3 v) const {   // MicrofacetDistribution.h interpolatePhongExponent
        const float sinTheta2 = Frame::sinTheta2(v);
        if (isIso() || sinTheta2 <= 2.93873587705571876e-39f /*RCPOVERFLOW*/) return expU;
        float invSinTheta2 = 1 / sinTheta2, cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
        return expU * cosPhi2 + expV * sinPhi2;
    }
    float eval(V3 m) const {   // MicrofacetDistribution.cu:6-42
        if (Frame::cosTheta(m) <= 0) return 0.0f;
        float cosTheta2 = m.z * m.z;
        float beckmannExponent = ((m.x * m.x) / (alphaU * alphaU) + (m.y * m.y) / (alphaV * alphaV)) / cosTheta2;
        float result;
        if (type == CTL_MF_BECKMANN) result = mexp(-beckmannExponent) / (PI * alphaU * alphaV * cosTheta2 * cosTheta2);
        else if (type == CTL_MF_GGX) { float root = (1 + beckmannExponent) * cosTheta2; result = 1.0f / (PI * alphaU * alphaV * root * root); }
        else { float e = interpPhongExp(m); result = std::sqrt((expU + 2) * (expV + 2)) * INV_TWOPI * mpow(Frame::cosTheta(m), e); }
        if (result < 1e-20f) result = 0;
        return result;
    }
    float projectRoughness(V3 v) const {
        float invSinTheta2 = 1 / Frame::sinTheta2(v);
        if (isIso() || invSinTheta2 <= 0) return alphaU;
        float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
        return std::sqrt(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
    }
    float smithG1(V3 v, V3 m) const {   // MicrofacetDistribution.cu:309-343
        if (dot(v, m) * Frame::cosTheta(v) <= 0) return 0.0f;
        const float tanTheta = fabsf(Frame::tanTheta(v));
        if (tanTheta == 0.0f) return 1.0f;
        float alpha = projectRoughness(v);
        if (type == CTL_MF_GGX) { const float root = alpha * tanTheta; return 2.0f / (1.0f + hypot2(1.0f, root)); }
        float a = 1.0f / (alpha * tanTheta);
        if (a >= 1.6f) return 1.0f;
        float aSqr = a * a;
        return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
    }
    float G(V3 wi, V3 wo, V3 m) const { return smithG1(wi, m) * smithG1(wo, m); }
    float pdfVisible(V3 wi, V3 m) const { if (Frame::cosTheta(wi) == 0) return 0.0f; return smithG1(wi, m) * absdot(wi, m) * eval(m) / fabsf(Frame::cosTheta(wi)); }
    float pdfAll(V3 m) const { return eval(m) * Frame::cosTheta(m); }
    float pdf(V3 wi, V3 m) const { return sampleVis ? pdfVisible(wi, m) : pdfAll(m); }
    void sampleFirstQuadrant(float u1, float& phi, float& exponent) const {   // MicrofacetDistribution.h:161-170
        phi = matan(std::sqrt((expU + 2.0f) / (expV + 2.0f)) * mtan(PI * u1 * 0.5f));
        const float sinPhi = msin(phi), cosPhi = mcos(phi);
        exponent = expU * cosPhi * cosPhi + expV * sinPhi * sinPhi;
    }
    V3 sampleAll(V2 sample, float& pdf) const {   // MicrofacetDistribution.cu:44-149
        float cosThetaM = 0.0f, sinPhiM, cosPhiM, alphaSqr;
        if (type == CTL_MF_PHONG) {   // :108-137
            float phiM, exponent;
            if (isIso()) { phiM = (2.0f * PI) * sample.y; exponent = expU; }
            else if (sample.y < 0.25f) sampleFirstQuadrant(4 * sample.y, phiM, exponent);
            else if (sample.y < 0.5f) { sampleFirstQuadrant(4 * (0.5f - sample.y), phiM, exponent); phiM = PI - phiM; }
            else if (sample.y < 0.75f) { sampleFirstQuadrant(4 * (sample.y - 0.5f), phiM, exponent); phiM += PI; }
            else { sampleFirstQuadrant(4 * (1 - sample.y), phiM, exponent); phiM = 2 * PI - phiM; }
            sinPhiM = msin(phiM); cosPhiM = mcos(phiM);
            cosThetaM = mpow(sample.x, 1.0f / (exponent + 2.0f));
            pdf = std::sqrt((expU + 2.0f) * (expV + 2.0f)) * INV_TWOPI * mpow(cosThetaM, exponent + 1.0f);
            if (pdf < 1e-20f) pdf = 0;
            float sinThetaP = std::sqrt(fmax2(0.0f, 1 - cosThetaM * cosThetaM));
            return V3(sinThetaP * cosPhiM, sinThetaP * sinPhiM, cosThetaM);
        }
        if (isIso()) { float a = (2.0f * PI) * sample.y; sinPhiM = msin(a); cosPhiM = mcos(a); alphaSqr = alphaU * alphaU; }
        else {
            float phiM = matan(alphaV / alphaU * mtan(PI + 2 * PI * sample.y)) + PI * floorf(2 * sample.y + 0.5f);
            sinPhiM = msin(phiM); cosPhiM = mcos(phiM);
            float cosSc = cosPhiM / alphaU, sinSc = sinPhiM / alphaV;
            alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
        }
        if (type == CTL_MF_BECKMANN) {
            float tanThetaMSqr = alphaSqr * -mlog(1.0f - sample.x);
            cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
            pdf = (1.0f - sample.x) / (PI * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
        } else {
            float tanThetaMSqr = alphaSqr * sample.x / (1.0f - sample.x);
            cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
            float temp = 1 + tanThetaMSqr / alphaSqr;
            pdf = INV_PI / (alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
        if (pdf < 1e-20f) pdf = 0;
        float sinThetaM = std::sqrt(fmax2(0.0f, 1 - cosThetaM * cosThetaM));
        return V3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }
    V2 sampleVisible11(float thetaI, V2 sample) const {   // MicrofacetDistribution.cu:185-307
        V2 slope;
        if (type == CTL_MF_BECKMANN) {   // :191-256: Newton / bisection on the CDF in the erf domain
            const float SQRT_PI_INV = 1 / std::sqrt(PI);
            if (thetaI < 1e-4f) { float r = std::sqrt(-mlog(1.0f - sample.x)); float a = 2 * PI * sample.y; return V2{ r * mcos(a), r * msin(a) }; }
            float tanThetaI = mtan(thetaI), cotThetaI = 1 / tanThetaI;
            float a = -1, c = erfRef(cotThetaI);
            float sample_x = sample.x > 1e-6f ? sample.x : 1e-6f;
            float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            float b = c - (1 + c) * mpow(1 - sample_x, fit);
            float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * mexp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = 0.5f * (a + c);
                float invErf = erfinvRef(b);
                float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * mexp(-invErf * invErf)) - sample_x;
                float derivative = normalization * (1 - invErf * tanThetaI);
                if (fabsf(value) < 1e-5f) break;
                if (value > 0) c = b; else a = b;
                b -= value / derivative;
            }
            slope.x = erfinvRef(b);
            slope.y = erfinvRef(2.0f * (sample.y > 1e-6f ? sample.y : 1e-6f) - 1.0f);
            return slope;
        }
        if (thetaI < 1e-4f) {
            float r = safe_sqrt(sample.x / (1 - sample.x)); float a = 2 * PI * sample.y;
            return V2{ r * mcos(a), r * msin(a) };
        }
        float tanThetaI = mtan(thetaI);
        float a = 1 / tanThetaI;
        float G1 = 2.0f / (1.0f + safe_sqrt(1.0f + 1.0f / (a * a)));
        float A = 2.0f * sample.x / G1 - 1.0f;
        if (fabsf(A) == 1) A -= copysign_bits(1.0f, A) * 1e-7f;
        float tmp = 1.0f / (A * A - 1.0f);
        float B = tanThetaI;
        float D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
        float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
        slope.x = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
        float Sg;
        if (sample.y > 0.5f) { Sg = 1.0f; sample.y = 2.0f * (sample.y - 0.5f); }
        else { Sg = -1.0f; sample.y = 2.0f * (0.5f - sample.y); }
        float z = (sample.y * (sample.y * (sample.y * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                  (sample.y * (sample.y * (sample.y * (sample.y * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
        slope.y = Sg * z * std::sqrt(1.0f + slope.x * slope.x);
        return slope;
    }
    V3 sampleVisible(V3 _wi, V2 sample) const {   // MicrofacetDistribution.cu:151-183
        V3 wi = normalize(V3(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
        float theta = 0, phi = 0;
        if (wi.z < 0.99999f) { theta = macos(wi.z); phi = matan2(wi.y, wi.x); }
        float sinPhi = msin(phi), cosPhi = mcos(phi);
        V2 slope = sampleVisible11(theta, sample);
        slope = V2{ cosPhi * slope.x - sinPhi * slope.y, sinPhi * slope.x + cosPhi * slope.y };
        slope.x *= alphaU; slope.y *= alphaV;
        float normalization = 1.0f / std::sqrt(slope.x * slope.x + slope.y * slope.y + (float)1.0);
        return V3(-slope.x * normalization, -slope.y * normalization, normalization);
    }
    V3 sample(V3 wi, V2 s, float& pdf) const {
        if (sampleVis) { V3 m = sampleVisible(wi, s); pdf = pdfVisible(wi, m); return m; }
        return sampleAll(s, pdf);
    }
};
inline V3 reflectAbout(V3 wi, V3 n) { return normalize(2 * dot(wi, n) * n - wi); }   // FresnelHelper.h:148-151
inline float avg3(Spec s) { float r = 0.0f; r += s.x; r += s.y; r += s.z; return r * (1.0f / 3); }   // Spectrum.h:180-190

} // namespace orc
#include "obsdf3.h"
#include "obsdf2.h"
#include "obsdf4.h"
namespace orc {

// --------------------------------------------------------------------------- BSDFs (SceneTypes/BSDF_Simple.cu)
inline bool bsdfHasComponent(const ctl_material& M, unsigned type) { return (type & M.combined_type) != 0; }

inline Spec bsdfSample(const ctl_material& M, BRec& bRec, float& pdf, V2 _sample) {
    if (M.bsdf_type >= CTL_BSDF_COATING) return bsdfComplexSample(M, bRec, pdf, _sample);
    switch (M.bsdf_type) {
    case CTL_BSDF_DIFFUSE: {   // BSDF_Simple.cu:7-36
        unsigned ct = M.combined_type;
        if (!(bRec.typeMask & ct) || (ct == CTL_EDiffuseReflection && Frame::cosTheta(bRec.wi) <= 0)) return Spec(0.0f);
        V2 sample = _sample;
        bRec.sampledType = ct;
        float sc = 1;
        if (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission)) {
            bRec.sampledType = sample.x < 0.5f ? CTL_EDiffuseReflection : CTL_EDiffuseTransmission;
            sample.x = sample.x < 0.5f ? sample.x * 2 : (sample.x - 0.5f) * 2;
            sc = 0.5f;
        }
        bRec.wo = squareToCosineHemisphere(sample);
        if ((ct == CTL_EDiffuseTransmission || (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission) && bRec.sampledType == CTL_EDiffuseTransmission)) && Frame::cosTheta(bRec.wi) > 0)
            bRec.wo.z *= -1;
        bRec.eta = 1.0f;
        pdf = fabsf(squareToCosineHemispherePdf(bRec.wo)) * sc;
        return texEval(M.tex[0], bRec.dg) * sc;
    }
    case CTL_BSDF_DIELECTRIC: {   // BSDF_Simple.cu:174-224 ; Dispersion.h sample_eta without dispersion -> eta = B + C/0.6
        bool sampleReflection = (bRec.typeMask & CTL_EDeltaReflection) != 0, sampleTransmission = (bRec.typeMask & CTL_EDeltaTransmission) != 0;
        Spec f_o(1.0f); float eta_pdf = 1.0f;
        float cosThetaT, eta = M.f[0] + M.f[1] / (600 / 1e3f), invEta = 1.0f / eta;
        float F = fresnelDielectricExt(Frame::cosTheta(bRec.wi), cosThetaT, eta);
        if (sampleTransmission && sampleReflection) {
            if (_sample.x <= F) {
                bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); bRec.eta = 1.0f; pdf = F;
                return texEval(M.tex[1], bRec.dg);
            } else {
                bRec.sampledType = CTL_EDeltaTransmission; bRec.wo = Frame::refract(bRec.wi, cosThetaT, eta, invEta);
                bRec.eta = cosThetaT < 0 ? eta : invEta; pdf = (1 - F) * eta_pdf;
                float factor = (cosThetaT < 0 ? invEta : eta);   // mode == ERadiance
                return f_o * texEval(M.tex[0], bRec.dg) * (factor * factor);
            }
        } else if (sampleReflection) {
            bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); bRec.eta = 1.0f; pdf = 1.0f;
            return texEval(M.tex[1], bRec.dg);
        } else if (sampleTransmission) {
            bRec.sampledType = CTL_EDeltaTransmission; bRec.wo = Frame::refract(bRec.wi, cosThetaT, eta, invEta);
            bRec.eta = cosThetaT < 0 ? eta : invEta; pdf = 1.0f * eta_pdf;
            float factor = (cosThetaT < 0 ? invEta : eta);
            return f_o * texEval(M.tex[0], bRec.dg) * (factor * factor * (1 - F));
        }
        return Spec(0.0f);
    }
    case CTL_BSDF_CONDUCTOR: {   // BSDF_Simple.cu:617-630
        bool sampleReflection = (bRec.typeMask & CTL_EDeltaReflection) != 0;
        if (!sampleReflection || Frame::cosTheta(bRec.wi) <= 0) return Spec(0.0f);
        bRec.sampledType = CTL_EDeltaReflection; bRec.wo = Frame::reflect(bRec.wi); bRec.eta = 1.0f; pdf = 1;
        return texEval(M.tex[0], bRec.dg) * fresnelConductorExact(Frame::cosTheta(bRec.wi), Spec(M.f[0], M.f[1], M.f[2]), Spec(M.f[3], M.f[4], M.f[5]));
    }
    case CTL_BSDF_ROUGHCONDUCTOR: {   // BSDF_Simple.cu:662-705
        if (Frame::cosTheta(bRec.wi) < 0 || !(bRec.typeMask & CTL_EGlossyReflection)) return Spec(0.0f);
        Microfacet distr((int)M.u[0], avg3(texEval(M.tex[1], bRec.dg)), avg3(texEval(M.tex[2], bRec.dg)), M.u[1] != 0);
        const V3 m = distr.sample(bRec.wi, _sample, pdf);
        if (pdf == 0) return Spec(0.0f);
        bRec.wo = reflectAbout(bRec.wi, m); bRec.eta = 1.0f; bRec.sampledType = CTL_EGlossyReflection;
        if (Frame::cosTheta(bRec.wo) <= 0) return Spec(0.0f);
        const Spec F = fresnelConductorExact(dot(bRec.wi, m), Spec(M.f[0], M.f[1], M.f[2]), Spec(M.f[3], M.f[4], M.f[5])) * texEval(M.tex[0], bRec.dg);
        float weight;
        if (distr.sampleVis) weight = distr.smithG1(bRec.wo, m);
        else weight = distr.eval(m) * distr.G(bRec.wi, bRec.wo, m) * dot(bRec.wi, m) / (pdf * Frame::cosTheta(bRec.wi));
        pdf /= 4.0f * dot(bRec.wo, m);
        return F * weight;
    }
    default: return bsdf2Sample(M, bRec, pdf, _sample);
    }
}

inline Spec bsdfF(const ctl_material& M, const BRec& bRec, int measure = ESolidAngle) {
    if (M.bsdf_type >= CTL_BSDF_COATING) return bsdfComplexF(M, bRec, measure);
    if (measure == EDiscrete) return bsdfFDiscrete(M, bRec);   // only nested evaluation asks for the discrete measure
    switch (M.bsdf_type) {
    case CTL_BSDF_DIFFUSE: {   // BSDF_Simple.cu:38-56
        unsigned ct = M.combined_type;
        if (!(bRec.typeMask & ct) || measure != ESolidAngle) return Spec(0.0f);
        bool validRefl = ct == CTL_EDiffuseReflection && Frame::cosTheta(bRec.wi) > 0 && Frame::cosTheta(bRec.wo) > 0;
        bool validTrans = ct == CTL_EDiffuseTransmission && Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) < 0;
        Spec s = texEval(M.tex[0], bRec.dg) * (INV_PI * fabsf(Frame::cosTheta(bRec.wo)));
        if (validRefl || validTrans) return s;
        else if (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission)) return s * 0.5f;
        return Spec(0.0f);
    }
    case CTL_BSDF_DIELECTRIC: case CTL_BSDF_CONDUCTOR:
        return Spec(0.0f);   // measure == ESolidAngle on every call site of the path (BSDF_Simple.cu:226-252, 632-646)
    case CTL_BSDF_ROUGHCONDUCTOR: {   // BSDF_Simple.cu:707-740
        if (measure != ESolidAngle || Frame::cosTheta(bRec.wi) < 0 || Frame::cosTheta(bRec.wo) < 0 || !(bRec.typeMask & CTL_EGlossyReflection)) return Spec(0.0f);
        V3 H = normalize(bRec.wo + bRec.wi);
        Microfacet distr((int)M.u[0], avg3(texEval(M.tex[1], bRec.dg)), avg3(texEval(M.tex[2], bRec.dg)), M.u[1] != 0);
        const float D = distr.eval(H);
        if (D == 0) return Spec(0.0f);
        const Spec F = fresnelConductorExact(dot(bRec.wi, H), Spec(M.f[0], M.f[1], M.f[2]), Spec(M.f[3], M.f[4], M.f[5])) * texEval(M.tex[0], bRec.dg);
        const float G = distr.G(bRec.wi, bRec.wo, H);
        float value = D * G / (4.0f * Frame::cosTheta(bRec.wi));
        return F * value;
    }
    default: return bsdf2F(M, bRec, measure);
    }
}

inline float bsdfPdf(const ctl_material& M, const BRec& bRec, int measure = ESolidAngle) {
    if (M.bsdf_type >= CTL_BSDF_COATING) return bsdfComplexPdf(M, bRec, measure);
    if (measure == EDiscrete) return bsdfPdfDiscrete(M, bRec);
    switch (M.bsdf_type) {
    case CTL_BSDF_DIFFUSE: {   // BSDF_Simple.cu:58-75
        unsigned ct = M.combined_type;
        if (!(bRec.typeMask & ct) || measure != ESolidAngle) return 0.0f;
        bool validRefl = ct == CTL_EDiffuseReflection && Frame::cosTheta(bRec.wi) > 0 && Frame::cosTheta(bRec.wo) > 0;
        bool validTrans = ct == CTL_EDiffuseTransmission && Frame::cosTheta(bRec.wi) * Frame::cosTheta(bRec.wo) < 0;
        float f = fabsf(squareToCosineHemispherePdf(bRec.wo));
        if (validRefl || validTrans) return f;
        else if (ct == (CTL_EDiffuseReflection | CTL_EDiffuseTransmission)) return f * 0.5f;
        return 0.0f;
    }
    case CTL_BSDF_DIELECTRIC: case CTL_BSDF_CONDUCTOR: return 0.0f;
    case CTL_BSDF_ROUGHCONDUCTOR: {   // BSDF_Simple.cu:742-763
        if (measure != ESolidAngle || Frame::cosTheta(bRec.wi) < 0 || Frame::cosTheta(bRec.wo) < 0 || !(bRec.typeMask & CTL_EGlossyReflection)) return 0.0f;
        V3 H = normalize(bRec.wo + bRec.wi);
        Microfacet distr((int)M.u[0], avg3(texEval(M.tex[1], bRec.dg)), avg3(texEval(M.tex[2], bRec.dg)), M.u[1] != 0);
        if (distr.sampleVis) return distr.eval(H) * distr.smithG1(bRec.wi, H) / (4.0f * Frame::cosTheta(bRec.wi));
        return distr.pdf(bRec.wi, H) / (4 * absdot(bRec.wo, H));
    }
    default: return bsdf2Pdf(M, bRec, measure);
    }
}

// --------------------------------------------------------------------------- hit -> material / bRec (Kernel/TraceResult.cu)
inline const ctl_material& hitMat(const Scene& S, const Hit& h) {   // TraceResult.cu:67-86
    return S.d.materials[triMatIndex(S.d.tri_data[h.tri], S.d.nodes[h.node].material_offset)];
}
inline uint32_t hitLightIndex(const Scene& S, const Hit& h) {        // TraceResult.cu:52-58
    uint32_t nli = hitMat(S, h).node_light_index;
    if (nli == UINT32_MAX) return UINT32_MAX;
    return S.d.nodes[h.node].lights[nli];
}
// TraceResult.cu:11-43.  wi is taken in the frame BEFORE the normal / height map perturbs it, as the reference does (:30-34)
inline void getBsdfSample(const Scene& S, const Hit& h, V3 rayO, V3 rayD, BRec& bRec) {
    bRec.eta = 1.0f; bRec.sampledType = 0; bRec.typeMask = EAll;
    bRec.dg.hasUVPartials = false;      // TraceResult::fillDG (Kernel/TraceResult.cu:13-21)
    bRec.dg.P = rayO + rayD * h.dist;   // Ray::operator()(t) (Math/Ray.h)
    fillDG(S, V2{ h.u, h.v }, h.tri, h.node, bRec.dg);
    bRec.wi = bRec.dg.sys.toLocal(-rayD);
    sampleNormalMap(hitMat(S, h), bRec.dg);
    if (hitMat(S, h).two_sided && bRec.wi.z < 0) {
        bRec.dg.n = -bRec.dg.n; bRec.dg.sys.n = -bRec.dg.sys.n; bRec.wi.z *= -1.0f;
    }
}

// --------------------------------------------------------------------------- direct lighting (Kernel/TraceAlgorithms.cu:44-101)
inline Spec estimateDirect(const Scene& S, BRec bRec, const ctl_material& mat, const ctl_light* light, float light_pdf, unsigned flags, Sampler& rng, uint64_t* rays) {
    DirectRec dRec(bRec.dg.P, bRec.dg.sys.n);
    Spec value = lightSampleDirect(S, *light, dRec, rng.randomFloat2());
    Spec retVal(0.0f);
    if (!isZero(value)) {
        bRec.wo = bRec.dg.sys.toLocal(dRec.d);
        bRec.typeMask = flags;
        Spec bsdfVal = bsdfF(mat, bRec);
        if (!isZero(bsdfVal)) {
            if (rays) (*rays)++;
            if (!occluded(S, dRec.ref, dRec.d, 0, dRec.dist)) {
                float weight = 1.0f;
                if (dRec.measure != EDiscrete) {
                    const float bsdfPdf_ = bsdfPdf(mat, bRec);
                    const float directPdf = (dRec.measure == EArea ? dRec.pdf * dRec.dist / fabsf(dot(dRec.n, dRec.d)) : dRec.pdf) * light_pdf;
                    weight = powerHeuristic(1, directPdf, 1, bsdfPdf_);
                }
                retVal = value * bsdfVal * weight;
            }
        }
    }
    return retVal;
}
inline Spec uniformSampleOneLight(const Scene& S, const BRec& bRec, const ctl_material& mat, Sampler& rng, uint64_t* rays) {
    if (!S.d.num_lights) return Spec(0.0f);
    V2 sample = rng.randomFloat2();
    float pdf;
    const ctl_light* light = sampleEmitter(S, pdf, sample);
    if (light == nullptr) return Spec(0.0f);
    return sdiv(estimateDirect(S, bRec, mat, light, pdf, EAll & ~EDelta, rng, rays), pdf);
}

// --------------------------------------------------------------------------- PathTrace<DIRECT> (Integrators/PathTracer.cu:10-113), no volumes
// DifferentialGeometry::computePartials (Engine/DifferentialGeometry.cu:9-90); rox / roy: origins of the x / y differential rays (= the ray's own for perspective sensors)
inline void computePartials(DG& dg, V3 rox, V3 rxd, V3 roy, V3 ryd) {
    dg.hasUVPartials = true;
    if (dot(dg.dpdu, dg.dpdu) == 0 && dot(dg.dpdv, dg.dpdv) == 0) { dg.dudx = dg.dvdx = dg.dudy = dg.dvdy = 0.0f; return; }
    const float pp = dot(dg.n, dg.P), pox = dot(dg.n, rox), poy = dot(dg.n, roy), prx = dot(dg.n, rxd), pry = dot(dg.n, ryd);
    if (prx == 0 || pry == 0) { dg.dudx = dg.dvdx = dg.dudy = dg.dvdy = 0.0f; return; }
    const float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
    const float absX = fabsf(dg.n.x), absY = fabsf(dg.n.y), absZ = fabsf(dg.n.z);
    int axes[2];
    if (absX > absY && absX > absZ) { axes[0] = 1; axes[1] = 2; } else if (absY > absZ) { axes[0] = 0; axes[1] = 2; } else { axes[0] = 0; axes[1] = 1; }
    const float dpduA[3] = { dg.dpdu.x, dg.dpdu.y, dg.dpdu.z }, dpdvA[3] = { dg.dpdv.x, dg.dpdv.y, dg.dpdv.z };
    const float A[2][2] = { { dpduA[axes[0]], dpdvA[axes[0]] }, { dpduA[axes[1]], dpdvA[axes[1]] } };
    const V3 px = rox + rxd * tx, py = roy + ryd * ty;
    const float pA[3] = { dg.P.x, dg.P.y, dg.P.z }, pxA[3] = { px.x, px.y, px.z }, pyA[3] = { py.x, py.y, py.z };
    const float Bx[2] = { pxA[axes[0]] - pA[axes[0]], pxA[axes[1]] - pA[axes[1]] }, By[2] = { pyA[axes[0]] - pA[axes[0]], pyA[axes[1]] - pA[axes[1]] };
    auto solve = [&](const float b[2], float x[2]) {   // AlgebraHelper::solveLinearSystem2x2 (Math/AlgebraHelper.h:11-24), RCPOVERFLOW = 2.93873587705571876e-39f
        const float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
        if (fabsf(det) <= 2.93873587705571876e-39f) return false;
        const float inverse = 1.0f / det;
        x[0] = (A[1][1] * b[0] - A[0][1] * b[1]) * inverse; x[1] = (A[0][0] * b[1] - A[1][0] * b[0]) * inverse;
        return true;
    };
    float x[2];
    if (solve(Bx, x)) { dg.dudx = x[0]; dg.dvdx = x[1]; } else { dg.dudx = 1; dg.dvdx = 0; }
    if (solve(By, x)) { dg.dudy = x[0]; dg.dvdy = x[1]; } else { dg.dudy = 0; dg.dvdy = 1; }
}

struct RayDiff { V3 ox, dx, oy, dy; };   // the sensor's x / y differential rays (sampleRayDifferential)
// diff non-null = the megakernel integrator's first-hit texture filtering
// (PathTracer.cu:60-61), null = no partials (what the wavefront tracer does)
// debugging aid of the parity fuzz (tools/fuzz_diag.py): when set for this thread, pathTrace appends one record of 26 floats per vertex —
// depth, triangle, node, material index, BSDF model, light index (-1), f.rgb, pdf, sampled type, cf.rgb and cl.rgb AFTER the vertex, hit distance, u, v, the ray that found the vertex (origin, direction)
inline std::vector<float>*& pathLog() { static thread_local std::vector<float>* p = nullptr; return p; }
// What the PRODUCT's kernels add for a path whose throughput became exactly zero (a test's aid, not the reference): they stop such a path at once (it cannot contribute
// any more; shade_kernel.inc, megakernel.hip) and count the radiance collected so far, where the reference traces it on — and DROPS the whole sample when the dead path later
// meets a NaN (Image::AddSample).  The path functions note the radiance at the first zero throughput here; orc_render adds it to a side image when the sample is then dropped
// (orc_set_zero_stop_image), so that a test can hold the kernels' frame to `oracle frame + side image` pixel by pixel instead of masking the pixels whose weights differ.
struct ZeroStop { bool have = false; bool pending = false; Spec cl; };
inline ZeroStop& zeroStop() { static thread_local ZeroStop z; return z; }
inline Spec pathTrace(const Scene& S, bool DIRECT, V3 ro, V3 rd, Sampler& rnd, int maxPathLength, int rrStartDepth, uint64_t* rays, const RayDiff* diff = nullptr, bool omitLastNEE = false) {
    Spec cl(0.0f), cf(1.0f);
    int depth = 0; bool specularBounce = false;
    BRec bRec; Hit r2; r2.init();
    float brdf_scattering_pdf = 0; V3 last_nor;
    zeroStop() = ZeroStop();
    while (depth++ < maxPathLength) {
        r2 = traceRayClosest(S, ro, rd);
        if (rays) (*rays)++;
        const V3 log_o = ro, log_d = rd;
        if (r2.hasHit()) {
            getBsdfSample(S, r2, ro, rd, bRec);
            if (depth == 1 && diff && S.pyramids) { bRec.dg.pyramids = S.pyramids; computePartials(bRec.dg, diff->ox, diff->dx, diff->oy, diff->dy); }
            const ctl_material& mat = hitMat(S, r2);
            uint32_t li = hitLightIndex(S, r2);
            if (li != UINT32_MAX) {
                float misWeight = 1.0f;
                if (!DIRECT || depth == 1 || specularBounce) misWeight = 1.0f;
                else {
                    DirectRec dRec(ro, last_nor);   // DirectSamplingRecFromRay (TraceAlgorithms.cu:33-42)
                    dRec.p = bRec.dg.P; dRec.n = bRec.dg.n; dRec.d = rd; dRec.dist = r2.dist; dRec.measure = ESolidAngle;
                    const ctl_light* light = S.d.lights + li;
                    float direct_pdf = lightPdfDirect(S, *light, dRec) * pdfEmitter(S, light);
                    misWeight = powerHeuristic(1, brdf_scattering_pdf, 1, direct_pdf);
                }
                cl = cl + misWeight * cf * lightEval(S, S.d.lights[li], bRec.dg.P, bRec.dg.sys, -rd);
            }
            Spec f = bsdfSample(mat, bRec, brdf_scattering_pdf, rnd.randomFloat2());
            last_nor = bRec.dg.sys.n;
            // omitLastNEE (a test's what-if, not the reference): PathTrace without the next-event estimation of its last vertex — the term pathIterateKernel never takes
            if (DIRECT && bsdfHasComponent(mat, ESmooth) && !(omitLastNEE && depth == maxPathLength)) cl = cl + cf * uniformSampleOneLight(S, bRec, mat, rnd, rays);
            specularBounce = (bRec.sampledType & EDelta) != 0;
            cf = cf * f;
            if (!zeroStop().have && isZero(cf)) { zeroStop().have = true; zeroStop().cl = cl; }   // where the kernels end the path (next-event estimation of this vertex included, as theirs is)
            ro = bRec.dg.P; rd = bRec.dg.sys.toWorld(bRec.wo);   // BSDFSamplingRecord::getOutgoing (Samples.cu)
            if (pathLog()) {
                const float rec[26] = { (float)depth, (float)r2.tri, (float)r2.node, (float)(&mat - S.d.materials), (float)mat.bsdf_type, li == UINT32_MAX ? -1.0f : (float)li, f.x, f.y, f.z,
                                        brdf_scattering_pdf, (float)bRec.sampledType, cf.x, cf.y, cf.z, cl.x, cl.y, cl.z, r2.dist, r2.u, r2.v, log_o.x, log_o.y, log_o.z, log_d.x, log_d.y, log_d.z };
                pathLog()->insert(pathLog()->end(), rec, rec + 26);
            }
        }
        if (!r2.hasHit()) break;
        if (depth > rrStartDepth && !specularBounce) {
            if (rnd.randomFloat() >= vmax(cf)) break;
            cf = sdiv(cf, vmax(cf));
        }
    }
    if (!r2.hasHit() && S.d.env_map_index != 0xffffffffu) {   // PathTracer.cu:99-111; EvalEnvironment == 0 without an environment map
        const ctl_light* light = S.d.lights + S.d.env_map_index;
        float misWeight = 1.0f;
        if (!DIRECT || depth == 1 || specularBounce) misWeight = 1.0f;
        else {
            DirectRec dRec(ro, last_nor);   // DirectSamplingRecFromRay(r, dist, last_nor, Vec3f(), NormalizedT<Vec3f>()), measure ESolidAngle
            dRec.p = V3(0.0f); dRec.n = V3(0.0f); dRec.d = rd; dRec.dist = r2.dist; dRec.measure = ESolidAngle;
            float direct_pdf = lightPdfDirect(S, *light, dRec) * pdfEmitter(S, light);
            misWeight = powerHeuristic(1, brdf_scattering_pdf, 1, direct_pdf);
        }
        cl = cl + misWeight * cf * envEval(S, *light, rd);
    } else if (!r2.hasHit()) {
        // without an environment map the line is still executed: cl += 1 * cf * Spectrum(0) (EvalEnvironment, KernelDynamicScene.cu:54-60) — nothing for a finite throughput, a NaN
        // for a NaN / infinite one: the escaped path of a BSDF evaluated outside its domain poisons its sample, and Image::AddSample drops it (found reading for the round-5 fuzz)
        cl = cl + 1.0f * cf * Spec(0.0f);
    }
    return cl;
}

// KernelDynamicScene::sampleEmitterDirect (Engine/KernelDynamicScene.cu:98-117): the emitter choice re-uses (and re-scales) the sample's first coordinate.
// `object` receives the chosen light when the sample is valid (dRec.object there).
inline Spec sampleEmitterDirect(const Scene& S, DirectRec& dRec, V2 sample, const ctl_light** object = nullptr) {
    dRec.pdf = 0; if (object) *object = nullptr;
    float emPdf = 0.0f;
    const ctl_light* emitter = sampleEmitter(S, emPdf, sample);
    if (!emitter) return Spec(0.0f);
    Spec value = lightSampleDirect(S, *emitter, dRec, sample);
    if (dRec.pdf != 0) { dRec.pdf *= emPdf; value = sdiv(value, emPdf); if (object) *object = emitter; return value; }
    return Spec(0.0f);
}
// ---- pathIterateKernel<NEXT_EVENT_EST> (Integrators/PseudoRealtime/WavefrontPathTracer.cu:51-164) followed along ONE path: the reference wavefront tracer's OWN
// per-path rules, which differ from PathTrace<DIRECT> at equal parameters (the product's PathSemantics = Wavefront):
//   * Russian roulette at pathDepth >= RRStartDepth, BEFORE the BSDF is sampled and whatever the last lobe was (:102-109);
//   * at the last bounce (pathDepth + 1 == maxPathDepth) emission is added and nothing is sampled: no next-event estimation there (:79, :111);
//   * next-event estimation through sampleEmitterDirect with ONE 2-D sample — the emitter choice re-uses its first coordinate (KernelDynamicScene.cu:98-117) —
//     drawn AFTER the BSDF sample (:113, :120); the MIS weight is taken for delta emitters as well (:127-129); the shadow ray is queued whatever f() returned;
//   * the shadow test of the next iteration: unoccluded iff the ray's closest hit lies at t >= dDist (1 - eps) (:62-70);
//   * the previous vertex' normal travels as two bytes (NormalizedFloat3ToUchar2, :137 -> :94, :151);
//   * u16bary: the hit's barycentrics pass through the 16-bit pair of the traversal result (Kernel/TraceHelper.cu:722-731, :44-51).
// The sampler is this oracle's deterministic one (index = film pixel, consumed along the path) — the reference indexes it by the racing queue slot and skips
// by the pass count (:58-59), which no two runs reproduce.  W = the sensor's importance weight (1 for the sensors here).
inline Spec pathTraceWavefront(const Scene& S, bool NEE, V3 ro, V3 rd, Sampler& rng, int maxPathDepth, int RRStartDepth, uint64_t* rays, bool u16bary) {
    Spec L(0.0f), throughput(1.0f), directF(0.0f);
    bool specular_bounce = true, have_shadow = false; float bsdf_pdf = 0.0f, dDist = 0.0f; uint16_t prev_normal = 0; V3 sh_o, sh_d;
    BRec bRec;
    zeroStop() = ZeroStop();
    for (int pathDepth = 0; pathDepth < maxPathDepth; pathDepth++) {
        Hit res = traceRayClosest(S, ro, rd);
        if (rays) (*rays)++;
        if (u16bary && res.hasHit()) {   // traversalResult: (uint16_t)(b * UINT16_MAX) -> Vec2f(x_disc, y_disc) / UINT16_MAX
            const uint16_t xd = (uint16_t)(res.u * 65535), yd = (uint16_t)(res.v * 65535);
            res.u = (float)xd / 65535.0f; res.v = (float)yd / 65535.0f;
        }
        if (NEE && pathDepth > 0 && have_shadow) {   // :62-73: the shadow ray queued by the previous vertex was traced (closest hit) together with this path ray
            if (rays) (*rays)++;
            const Hit sh = traceRayClosest(S, sh_o, sh_d, 1);
            if (sh.dist >= dDist * (1 - S.d.ray_trace_eps)) L = L + directF;
            have_shadow = false; directF = Spec(0.0f);
        }
        if (zeroStop().pending) { zeroStop().pending = false; zeroStop().have = true; zeroStop().cl = L; }   // the kernels' sample of a path that died at the previous vertex: its last shadow ray resolved
        bool path_terminated = (pathDepth + 1 == maxPathDepth);
        if (res.hasHit()) {
            getBsdfSample(S, res, ro, rd, bRec);
            const ctl_material& mat = hitMat(S, res);
            const uint32_t li = hitLightIndex(S, res);
            if (li != UINT32_MAX) {   // :86-99
                float misWeight = 1.0f;
                if (!NEE || pathDepth == 0 || specular_bounce) misWeight = 1.0f;
                else {
                    DirectRec dRec(ro, uchar2ToNormal(prev_normal));
                    dRec.p = bRec.dg.P; dRec.n = bRec.dg.n; dRec.d = rd; dRec.dist = res.dist; dRec.measure = ESolidAngle;
                    const ctl_light* light = S.d.lights + li;
                    const float direct_pdf = lightPdfDirect(S, *light, dRec) * pdfEmitter(S, light);
                    misWeight = powerHeuristic(1, bsdf_pdf, 1, direct_pdf);
                }
                L = L + misWeight * lightEval(S, S.d.lights[li], bRec.dg.P, bRec.dg.sys, -rd) * throughput;
            }
            bool surviveRR = true;   // :101-109
            if (pathDepth >= RRStartDepth) {
                if (rng.randomFloat() < vmax(throughput)) throughput = sdiv(throughput, vmax(throughput));
                else surviveRR = false;
            }
            if (pathDepth + 1 != maxPathDepth && surviveRR) {
                const Spec f = bsdfSample(mat, bRec, bsdf_pdf, rng.randomFloat2());
                specular_bounce = (bRec.sampledType & EDelta) != 0;
                const V3 new_o = bRec.dg.P, new_d = bRec.dg.sys.toWorld(bRec.wo);
                if (NEE && bsdfHasComponent(mat, ESmooth)) {
                    DirectRec dRec(bRec.dg.P, bRec.dg.sys.n);
                    const Spec value = sampleEmitterDirect(S, dRec, rng.randomFloat2());
                    if (!isZero(value)) {
                        bRec.typeMask = EAll & ~EDelta;
                        bRec.wo = bRec.dg.sys.toLocal(dRec.d);
                        const Spec bsdfVal = bsdfF(mat, bRec);
                        const float bsdfPdf_ = bsdfPdf(mat, bRec);
                        const float directPdf = dRec.measure == EArea ? dRec.pdf * dRec.dist / fabsf(dot(dRec.n, dRec.d)) : dRec.pdf;   // PdfAtoW (MonteCarlo.h)
                        const float weight = powerHeuristic(1, directPdf, 1, bsdfPdf_);
                        directF = throughput * value * bsdfVal * weight;
                        dDist = dRec.dist; sh_o = bRec.dg.P; sh_d = dRec.d; have_shadow = true;
                    }
                }
                prev_normal = normalToUchar2(bRec.dg.sys.n);
                throughput = throughput * f;
                if (!zeroStop().have && !zeroStop().pending && isZero(throughput)) zeroStop().pending = true;   // the kernels end the path here (shade_kernel.inc: alive = !is_zero(cf))
                ro = new_o; rd = new_d;
            } else path_terminated = true;
        } else {   // :143-157
            path_terminated = true;
            if (S.d.env_map_index != 0xffffffffu) {
                const ctl_light* light = S.d.lights + S.d.env_map_index;
                float misWeight = 1.0f;
                if (!(!NEE || pathDepth == 0 || specular_bounce)) {
                    DirectRec dRec(ro, uchar2ToNormal(prev_normal));
                    dRec.p = V3(0.0f); dRec.n = V3(0.0f); dRec.d = rd; dRec.dist = res.dist; dRec.measure = ESolidAngle;
                    const float direct_pdf = lightPdfDirect(S, *light, dRec) * pdfEmitter(S, light);
                    misWeight = powerHeuristic(1, bsdf_pdf, 1, direct_pdf);
                }
                L = L + misWeight * throughput * envEval(S, *light, rd);
            } else L = L + 1.0f * throughput * Spec(0.0f);   // EvalEnvironment == Spectrum(0) without a map (:156 is executed all the same): a NaN / infinite throughput poisons the sample
        }
        if (path_terminated) break;   // I.AddSample (:159-162)
    }
    return L;
}

// ---- PathTraceRegularization<DIRECT> (Integrators/PathTracer.cu:115-173): the PathTracer plugin with Regularization = true
// Light::samplePosition of the emitters the mollified connection uses (SceneTypes/Light.cu:33-40 point, :304-311 spot, :246-258 distant); area and
// environment emitters are skipped by the caller (PathTracer.cu:139), so only their sample consumption matters
inline Spec lightSamplePosition(const ctl_light& L, V2 sample, V3& p) {
    if (L.type == CTL_LIGHT_POINT || L.type == CTL_LIGHT_SPOT) { p = V3(L.position[0], L.position[1], L.position[2]); return Spec(L.radiance[0], L.radiance[1], L.radiance[2]) * (4 * PI); }
    if (L.type == CTL_LIGHT_DISTANT) {
        const V2 q = squareToUniformDiskConcentric(sample);
        const Frame F = lightFrame(L);
        const V3 perpOffset = F.toWorld(V3(q.x, q.y, 0) * L.bsphere_radius), d = F.toWorld(V3(0.0f, 0.0f, 1.0f));
        p = d * L.bsphere_radius + perpOffset;
        const float surfaceArea = PI * L.bsphere_radius * L.bsphere_radius, invSurfaceArea = 1.0f / surfaceArea;   // Light.h:159-165
        return sdiv(Spec(L.radiance[0], L.radiance[1], L.radiance[2]), invSurfaceArea);                                 // m_power (Light.cu:208-212)
    }
    p = V3(0.0f); return Spec(0.0f);
}
// UniformSampleAllLights (Kernel/TraceAlgorithms.cu:75-90), nSamples = 1
inline Spec uniformSampleAllLights(const Scene& S, const BRec& bRec, const ctl_material& mat, Sampler& rng, uint64_t* rays) {
    Spec L(0.0f);
    for (unsigned i = 0; i < S.d.num_lights; i++) L = L + estimateDirect(S, bRec, mat, S.d.lights + S.d.light_indices[i], 1.0f, EAll & ~EDelta, rng, rays) / 1.0f;
    return L;
}
// InfiniteLight::evalEnvironment(ray, rX, rY) (SceneTypes/Light.cu:496-518)
inline Spec envEvalDifferential(const Scene& S, const ctl_light& L, V3 dir, V3 dirX, V3 dirY) {
    V3 v = transformDirTranspose(L.to_world, dir);
    V2 uv{ matan2(v.x, -v.z) * INV_TWOPI, safe_acos(v.y) * INV_PI };
    V3 dvdx = transformDirTranspose(L.to_world, dirX) - v, dvdy = transformDirTranspose(L.to_world, dirY) - v;
    float t1 = INV_TWOPI / (v.x * v.x + v.z * v.z), t2 = -INV_PI / fmax2(safe_sqrt(1.0f - v.y * v.y), 1e-4f);
    V2 dudx{ t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y }, dudy{ t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y };
    return mipEval(S.d.images[L.env_image], S.pyramids[L.env_image], uv, dudx, dudy) * Spec(L.env_scale[0], L.env_scale[1], L.env_scale[2]);
}
inline Spec pathTraceRegularization(const Scene& S, bool DIRECT, V3 ro, V3 rd, const RayDiff& diff, Sampler& rnd, float g_fRMollifier, int maxPathLength, int rrStartDepth, uint64_t* rays) {
    Hit r2; r2.init();
    Spec cl(0.0f), cf(1.0f);
    int depth = 0; bool specularBounce = false;
    BRec bRec;
    zeroStop() = ZeroStop();   // (k_path_trace_regularization makes no zero-throughput cut)
    for (;;) {   // while (traceRay(r, &r2) && depth++ < maxPathLength)
        r2 = traceRayClosest(S, ro, rd);
        if (rays) (*rays)++;
        if (!(r2.hasHit() && depth++ < maxPathLength)) break;
        getBsdfSample(S, r2, ro, rd, bRec);
        if (depth == 1 && S.pyramids) { bRec.dg.pyramids = S.pyramids; computePartials(bRec.dg, diff.ox, diff.dx, diff.oy, diff.dy); }
        const ctl_material& mat = hitMat(S, r2);
        if (!DIRECT || (depth == 1 || specularBounce)) {
            uint32_t li = hitLightIndex(S, r2);
            if (li != UINT32_MAX) cl = cl + cf * lightEval(S, S.d.lights[li], bRec.dg.P, bRec.dg.sys, -rd);   // TraceResult::Le
        }
        float pdf_unused;
        Spec f = bsdfSample(mat, bRec, pdf_unused, rnd.randomFloat2());
        if (DIRECT) {
            if (bsdfHasComponent(mat, EDelta)) {   // connect the sampled direction of a BSDF with delta lobes to a point / spot / distant emitter inside a cone that shrinks with the passes
                V2 sample = rnd.randomFloat2();
                if (S.d.num_lights) {
                    float emPdf; const ctl_light* l = sampleEmitter(S, emPdf, sample);   // KernelDynamicScene::sampleEmitterPosition (KernelDynamicScene.cu:156-168)
                    V3 lp; Spec l_s = sdiv(lightSamplePosition(*l, sample, lp), emPdf);
                    float lDist = length(lp - bRec.dg.P);
                    V3 lDir = (lp - bRec.dg.P) / lDist;
                    if (!(l->type == CTL_LIGHT_DIFFUSE || l->type == CTL_LIGHT_INFINITE)) {
                        if (rays) (*rays)++;
                        if (!occluded(S, bRec.dg.P, lDir, 0, lDist)) {
                            float eps = matan(g_fRMollifier / lDist);
                            float normalization = 1.0f / (2 * PI * (1 - mcos(eps)));
                            float l_dot_o = dot(lDir, bRec.dg.sys.toWorld(bRec.wo));
                            float indicator = macos(l_dot_o) <= eps ? 1.0f : 0.0f;
                            cl = cl + cf * f * l_s * (normalization * indicator);
                        }
                    }
                }
            } else cl = cl + cf * uniformSampleAllLights(S, bRec, mat, rnd, rays);
        }
        specularBounce = (bRec.sampledType & EDelta) != 0;
        cf = cf * f;
        if (depth > rrStartDepth) {
            if (rnd.randomFloat() < vmax(cf)) cf = sdiv(cf, vmax(cf));
            else break;
        }
        ro = bRec.dg.P; rd = bRec.dg.sys.toWorld(bRec.wo);
        r2.init();
    }
    // PathTracer.cu:168-171: the environment is added for the LAST ray whether it escaped or not (a path cut by the roulette or by the
    // depth limit adds the environment radiance seen along the ray that produced its last hit); reproduced as written
    if (S.d.env_map_index != 0xffffffffu) {
        const ctl_light& env = S.d.lights[S.d.env_map_index];
        if (!r2.hasHit() && depth == 0) cl = cf * (S.pyramids ? envEvalDifferential(S, env, rd, diff.dx, diff.dy) : envEval(S, env, rd));
        else cl = cl + cf * envEval(S, env, rd);
    } else if (!r2.hasHit() && depth == 0) cl = cf * Spec(0.0f);   // EvalEnvironment == Spectrum(0) without a map
    else cl = cl + cf * Spec(0.0f);
    return cl;
}

// Engine/Image.cu:22-44 (host branch)
inline void addSample(ctl_pixel_data* img, int W, int H, float sx, float sy, Spec L) {
    L = V3(fmax2(0.0f, L.x), fmax2(0.0f, L.y), fmax2(0.0f, L.z));   // Spectrum::clampNegative (Spectrum.h:241-244): max(0, s) = (0 > s) ? 0 : s — a NaN stays a NaN and the sample is dropped below
                                                                     // (the operands the other way round would turn it into 0 and count the sample: found by the parity fuzz, round 5)
    int x = floor2int(sx), y = floor2int(sy);
    bool bad = std::isnan(L.x) || std::isnan(L.y) || std::isnan(L.z) || std::isinf(L.x) || std::isinf(L.y) || std::isinf(L.z);
    if (x < 0 || x >= W || y < 0 || y >= H || bad) return;
    ctl_pixel_data& r = img[(size_t)y * W + x];
    r.rgb[0] += L.x; r.rgb[1] += L.y; r.rgb[2] += L.z; r.weight_sum += 1.0f;
}

} // namespace orc
