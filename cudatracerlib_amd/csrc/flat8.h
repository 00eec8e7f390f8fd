// flat8.h — the 8-wide node of the flattened world-space BVH ("Q8", flatten.h) and the ONE decode of it that the builder (flatten.cpp), the traversal
// kernels (traverse_flat8.h) and the test oracle's mirror (oracle/ocore.h) share.
//
// Why 8-wide (round 4).  The 4-wide kernel sat at the knee between VALU issue and memory round trips: a node step is ~250 VALU instructions of which 88 test
// boxes — the rest is the step's fixed part (loop, pop, link decode, ordering network, three pushes) — and every step ends in one wait for the slowest lane's line.  An 8-wide
// node halves the steps per ray and with them BOTH the fixed instructions and the dependent round trips; what makes it affordable on a 64-wide SIMT machine is that nothing
// in the step depends on the order of the children any more:
//   * slots are OCTANT-ORDERED at build time (Ylitie, Karras, Laine: "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017, §3.2 — the published
//     idea; layout and code here are this project's own): child c of a node goes to the slot s whose sign vector (bit k of s set = "+" on axis k) best matches the offset of its
//     centroid from the node's, so a ray visits the slots in the order of decreasing (s XOR octinv), octinv = the ray's three direction signs.  No distance sort: the hit
//     children of a step are ONE byte, and the traversal stack holds ONE 8-byte group {first inner child, inner mask, hit byte} per level instead of up to three links per step;
//   * a leaf slot holds exactly ONE triangle entry (98.8 % of the 4-wide tree's leaves did already): entry = first entry of the node + rank of the slot among the leaf slots,
//     so a step's leaf hits are one more byte and the (ray, entry) pairs they stand for need no per-slot counts.
//
// Layout, 128 B = one memory-side line, line-aligned; a step loads the first 80 B (five 16-B loads per lane), 96 B when the link that led here says so:
//   q0  origin[3] (f32) | e[3] (biased exponents of the per-axis quantisation step), imask (bit s: slot s is an INNER child)
//   q1  base_b: bits 0..23 node index of the first inner child, bits 24..31 "B" (bit s: an inner slot -> the child node has leaf children or a slab: its q5 is loaded too;
//                a non-inner slot -> the slot is a LEAF) | leaf_base: first leaf entry of the node | slab_n, slab_base: flat_slab.h word 0 and base (slab_n == 0: no slab)
//   q2  qlo_x[8] qlo_y[8]      byte (s & 3) of word (s >> 2) = slot s; child box = origin + 2^(e-127) * code, conservative; an EMPTY slot has the inverted box lo = 255, hi = 0
//   q3  qlo_z[8] qhi_x[8]
//   q4  qhi_y[8] qhi_z[8]
//   q5  slab_lo[8] slab_hi[8]  interval codes of the node's oriented slab (flat_slab.h) per slot; whole range 0..255 for children without an interval of their own
//   q6, q7 unused (never loaded)
// Inner children of a node are consecutive nodes in slot order, the entries of its leaf slots consecutive entries in slot order.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CTL_F8_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define CTL_F8_HD inline
#endif

namespace ctl {

struct flat8_node {
    float origin[3];
    uint8_t e[3]; uint8_t imask;
    uint32_t base_b, leaf_base, slab_n; float slab_base;
    uint32_t qlo_x[2], qlo_y[2], qlo_z[2], qhi_x[2], qhi_y[2], qhi_z[2];
    uint32_t slab_lo[2], slab_hi[2];
    uint32_t pad[8];
};
static_assert(sizeof(flat8_node) == 128, "8-wide node is one 128-B line");

constexpr uint32_t kFlat8MaxNodes = 1u << 24;          // node indices are 24 bits (base_b)
constexpr int kFlat8StackGroups = 64;                   // sibling groups a lane's traversal stack can hold (one per level): the tree depth is checked against it at upload
constexpr uint32_t kFlat8None = 0x76543210u;           // host-side explicit link of an empty slot (as in the other formats)

CTL_F8_HD uint32_t flat8_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_popcount(x);
#else
    x = x - ((x >> 1) & 0x55555555u); x = (x & 0x33333333u) + ((x >> 2) & 0x33333333u); return (((x + (x >> 4)) & 0x0f0f0f0fu) * 0x01010101u) >> 24;
#endif
}
CTL_F8_HD uint32_t flat8_below(uint32_t slot) { return (1u << slot) - 1u; }                                   // mask of the slots before `slot`
CTL_F8_HD uint32_t flat8_inner_mask(uint32_t q0w) { return q0w >> 24; }                                      // q0w: the word {e[3], imask}
CTL_F8_HD uint32_t flat8_leaf_mask(uint32_t q0w, uint32_t base_b) { return (base_b >> 24) & ~(q0w >> 24); }   // leaf slots
CTL_F8_HD uint32_t flat8_heavy_mask(uint32_t q0w, uint32_t base_b) { return (base_b >> 24) & (q0w >> 24); }   // inner slots whose node is loaded with its q5
CTL_F8_HD uint32_t flat8_child_node(uint32_t base_b, uint32_t imask, uint32_t slot) { return (base_b & 0x00ffffffu) + flat8_popc(imask & flat8_below(slot)); }
CTL_F8_HD uint32_t flat8_leaf_entry(uint32_t leaf_base, uint32_t lmask, uint32_t slot) { return leaf_base + flat8_popc(lmask & flat8_below(slot)); }
// a ray's octant word: bit k set when its direction is NOT negative on axis k.  Slots are visited in decreasing (slot ^ octinv): the "-" side of an axis first when the ray runs towards "+".
CTL_F8_HD uint32_t flat8_octinv(float dx, float dy, float dz) { return (dx < 0.0f ? 0u : 1u) | (dy < 0.0f ? 0u : 2u) | (dz < 0.0f ? 0u : 4u); }
// a byte of per-slot bits (bit s = slot s) -> visiting order (bit p = the slot with s ^ octinv == p; the highest set bit is visited first); its own inverse.
// Works on several bytes of a word at once (the kernels permute the inner-hit byte and the leaf-hit byte together).
CTL_F8_HD uint32_t flat8_to_order(uint32_t bits, uint32_t octinv) {
    if (octinv & 1u) bits = ((bits & 0x55555555u) << 1) | ((bits >> 1) & 0x55555555u);
    if (octinv & 2u) bits = ((bits & 0x33333333u) << 2) | ((bits >> 2) & 0x33333333u);
    if (octinv & 4u) bits = ((bits & 0x0f0f0f0fu) << 4) | ((bits >> 4) & 0x0f0f0f0fu);
    return bits;
}

}  // namespace ctl
