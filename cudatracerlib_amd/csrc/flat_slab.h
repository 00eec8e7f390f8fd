// flat_slab.h — the oriented slab of a bottom node of the flattened BVH (flatten.h): a FOURTH slab axis next to x, y, z.
//
// Why: 8.44 M of the 8.54 M leaves of the bench scene hold one triangle, and 70 % of the 128-B leaf entries a ray fetches are rays that cross
// the triangle's axis-aligned box but not the triangle's plane inside it (counted by the oracle).  An axis-aligned box cannot say that;
// a slab along the triangle's normal can.  A node whose children are (mostly) leaves stores ONE direction n — picked by the builder among its
// triangles' normals — and per child the interval of  D(x) = n . (x - origin)  its triangles cover.  The node step treats it exactly like a box axis:
//     t = (D - s) / r      with  s = n . (o - origin),  r = n . d
// so a child costs two byte conversions and two FMAs more, its entry / exit distances are clipped by the slab, and a ray that misses the slab inside
// the box never fetches the leaf entry.  The slab only culls: every entry that is looked at is still decided by the reference's arithmetic.
//
// Layout (the last 16 B of a flat4_node, read only for nodes whose parent link says so):
//     word 0: bits 0..17: n as three 6-bit signed integers (largest |component| = 31); bits 18..31: the top 14 bits of the float `step` (5 mantissa bits)
//     word 1: base (float)  —  D of code 0
//     word 2: lo codes, byte c = child c        D interval of child c = [base + step * lo_c, base + step * hi_c]
//     word 3: hi codes                          inner children: 0 .. 255 = the whole node; missing children: 255 .. 0
//
// Conservative under the kernel's own fp32 evaluation: the builder pads every interval by (a) the distance by which the fp32 object-space Woop test
// can accept a point off the exact triangle (2^-20 of: instance scale x object-space magnitude + world magnitude — about 8 x its round-off), (b) the
// node-extent share of the evaluation error, and rounds the codes outwards (also under the fp32 evaluation base + step * code); the kernel pads by
// kSlabRayPad * |o - origin|_1, the ray-dependent share (the error of s and of t * r grows with the distance between the ray origin and the node).
// tests/test_flat_slab.py decodes every child's interval of a tree and holds its triangles' vertices inside it by the pad, and runs the whole chain
// against the exact test on millions of (ray, entry) pairs.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define CTL_SLAB_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define CTL_SLAB_HD inline
#endif

namespace ctl {

constexpr int kSlabSubtreeDefault = 32;                            // an inner child with at most this many triangles under it gets its own interval (0: leaf children only); measured in DESIGN.md §3
constexpr int kSlabSubtreeMax = 64;
constexpr int kSlabNMax = 31;                                   // largest |component| of the integer normal
constexpr float kSlabRayPad = 31.0f * 1.9073486328125e-6f;     // kSlabNMax * 2^-19: 8 x the worst-case fp32 error of s + t r per unit of |o - origin|_1

struct slab_ray {        // what a node step needs of a node's slab for one ray
    float alpha;         // step / r
    float beta_n, beta_f;  // (base -+ pad - s) / r : entry / exit side
    uint32_t near_w, far_w;   // code words in entry / exit order
    bool neg;                 // r < 0: entry side = the hi codes (8-wide nodes pick their two code words per side themselves)
};

CTL_SLAB_HD float slab_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// guarded reciprocal as for the box slabs (TraceHelper.cu:417-420: |x| < 2^-80 is replaced by +-2^-80)
CTL_SLAB_HD float slab_rcp(float r) {
    const float ooeps = 8.271806125530277e-25f;
    uint32_t b; memcpy(&b, &r, 4);
    const float g = (r < 0 ? -r : r) > ooeps ? r : slab_u2f((b & 0x80000000u) | 0x17800000u);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(g);
#else
    return 1.0f / g;
#endif
}

// nw / base / lo_w / hi_w: the four words of the slab quarter; origin: the node's origin; (ox..dz): the ray
CTL_SLAB_HD void slab_setup(uint32_t nw, float base, uint32_t lo_w, uint32_t hi_w, float orgx, float orgy, float orgz,
                            float ox, float oy, float oz, float dx, float dy, float dz, slab_ray& R) {
    const float nx = (float)((int32_t)(nw << 26) >> 26), ny = (float)((int32_t)(nw << 20) >> 26), nz = (float)((int32_t)(nw << 14) >> 26);
    const float step = slab_u2f(nw & 0xfffc0000u);
    const float ex = ox - orgx, ey = oy - orgy, ez = oz - orgz;
    const float s = __builtin_fmaf(nz, ez, __builtin_fmaf(ny, ey, nx * ex));
    const float r = __builtin_fmaf(nz, dz, __builtin_fmaf(ny, dy, nx * dx));
    const float rr = slab_rcp(r);
    const float pad = (__builtin_fabsf(ex) + __builtin_fabsf(ey) + __builtin_fabsf(ez)) * kSlabRayPad;
    const bool neg = rr < 0.0f;
    const float u = base - s;
    R.alpha = step * rr;
    R.beta_n = (neg ? u + pad : u - pad) * rr;
    R.beta_f = (neg ? u - pad : u + pad) * rr;
    R.near_w = neg ? hi_w : lo_w; R.far_w = neg ? lo_w : hi_w; R.neg = neg;
}
// entry / exit distance of child k's slab
CTL_SLAB_HD float slab_near(const slab_ray& R, int k) { return __builtin_fmaf((float)((R.near_w >> (8 * k)) & 0xffu), R.alpha, R.beta_n); }
CTL_SLAB_HD float slab_far(const slab_ray& R, int k) { return __builtin_fmaf((float)((R.far_w >> (8 * k)) & 0xffu), R.alpha, R.beta_f); }

}  // namespace ctl
