// flatten.h — single-level world-space acceleration structure over every instanced triangle of a two-level scene (see flatten.cpp).
//
// The flattened structure is a CULLING structure only: its boxes live in world space, but every leaf entry carries the
// OBJECT-space Woop rows of its triangle and the node that instances it, and the traversal kernel evaluates it with the
// reference's two-level arithmetic (instance inverse transform, object-space Woop test, exact division — Kernel/TraceHelper.cu
// :526-560, :646-682).  The reported (t, u, v, triangle, node) are therefore the reference's bit for bit; the tree only decides
// which entries are looked at.
#pragma once
#include "../../include/ctl_amd.h"
#include "flat8.h"
#include <vector>
#include <cstddef>

namespace ctl {

// leaf entry, 128 B = one L2 line holding everything the exact test of one instanced triangle needs, so that a leaf step is ONE dependent
// fetch: the mesh's own Woop rows (TriIntersectorData, Engine/TriIntersectorData.h:10-40), unchanged; index = globalTri << 1 | lastEntryOfLeaf;
// node = the instancing node; inv = rows 0..2 of the node's inverse transform and w33 = its element (3,3) (float4x4.h:402-406 divides by it).
// Measured alternatives on synthetic-SM: 64-B entries + a second, dependent fetch of the instance record 2.17 Grays/s,
// 16-B entries pointing into the meshes' shared Woop stream 2.16, these 128-B entries 2.25.
struct flat_leaf { float a[4], b[4], c[4]; uint32_t index; uint32_t node; uint32_t pad[2]; float inv[12]; float w33; uint32_t pad2[3]; };
static_assert(sizeof(flat_leaf) == 128, "flat leaf entry is one 128-B line");

// node formats (one per flat_scene; the traversal kernels are instantiated per format)
enum flat_format : int {
    kFlatQ4 = 0,   // 4-wide, 64 B, child boxes quantised to 8 bits against the node's own box
    kFlatF4 = 1,   // 4-wide, 128 B (one L2 line), fp32 child boxes stored plane-major
    kFlatF2 = 2,   // 2-wide, 64 B, fp32 child boxes: the reference's BVHNodeData layout (Engine/TriIntersectorData.h:42-117)
    kFlatQ8 = 3,   // 8-wide, 128 B (one L2 line, 96 B used), 8-bit child boxes, one-triangle leaf slots, octant-ordered slots (flat8.h)
};

// Q4.  Child box c, axis k:  lo = origin[k] + 2^(e[k]-127) * qlo[k][c],  hi = origin[k] + 2^(e[k]-127) * qhi[k][c]  (conservative).
// Everything a traversal step needs sits in the first 48 B (three 16-B loads per lane instead of four): the child links are implied by the
// layout — the inner children of a node are consecutive nodes, the entries of its leaf children are consecutive leaf entries, both in slot
// order — and spelled so that a link is ONE addition (round 3; the first form, a base and per-slot entry counts, cost the node step 45 VALU
// instructions and a dozen exec-mask branches of prefix sums):
//   links[0]  bits 0..25: first inner child's node index * 4 | slab flag of slot 0 (bit 0)     bits 26..29: t1     bits 30..31: t3 & 3
//   links[1]  bits 0..1: t3 >> 2     bits 2..5: t2     bits 6..31: ~(first leaf entry of the node + 15), low 26 bits
//   t_k of an inner child  = (inner children in slots < k) << 2 | slab flag of slot k      ->  link = inner base * 4 + t_k
//   t_k of a leaf child    = 15 - (entries of the leaf children in slots < k)              ->  link = ~(leaf base + 15) + t_k = ~(its first entry)
// (slot 0: t_0 = slab flag or 15, implied; a leaf has at most four entries; fewer than 2^24 nodes and 2^26 - 15 entries).  The slab flag says that the
// child node carries an ORIENTED SLAB (flat_slab.h) in its last 16 B; the traversal hands that bit down in the child link (bit 0 of an inner link), so a
// lane knows BEFORE it fetches a node whether to load 48 or 64 B.  The last 16 B are that slab when the tree has implied links, else the explicit links
// child[4] (the kernels then read them and no node has a slab); host code and the test oracle find the explicit links of every tree in flat_scene::child_links.
struct flat4_node {
    float origin[3];
    uint8_t e[3];          // biased float exponents of the per-axis quantisation step
    uint8_t mask;          // bit c set <=> child c exists; bit 4 + c set (with bit c) <=> child c is a leaf.  A missing child has an INVERTED box (lo = 255, hi = 0), so the kernel's slab test alone rejects it,
                           // and its link repeats a sibling's (flatten.cpp: a slot that round-off lets through must lead somewhere harmless) — which is why an empty slot of a node without inner children carries a leaf bit
    uint32_t qlo_x, qhi_x, qlo_y, qhi_y, qlo_z, qhi_z;   // byte c = child c
    uint32_t links[2];
    union {
        int32_t child[4];  // explicit links (trees without implied links): >= 0: node index * 4 (float4 units); < 0: ~firstLeafEntry; 0x76543210: none
        struct { uint32_t slab_n; float slab_base; uint32_t slab_lo, slab_hi; };   // flat_slab.h: normal (3 x int8) + step exponent, D of code 0, per-child interval codes (byte c = child c)
    };
};
static_assert(sizeof(flat4_node) == 64, "quantised wide node is one 64-B fetch group");
// the links a traversal step derives from the layout (what node_step_q4 computes): inner child = node index * 4 | slab flag, leaf child = ~first entry
inline void flat4_link_nibbles(const uint32_t w0, const uint32_t w1, uint32_t t[4]) { t[0] = 0; t[1] = (w0 >> 26) & 15u; t[2] = (w1 >> 2) & 15u; t[3] = (w0 >> 30) | ((w1 & 3u) << 2); }
inline void flat4_implied_links(const flat4_node& n, int32_t c[4]) {
    const uint32_t w0 = n.links[0], w1 = n.links[1], leafm = n.mask >> 4;
    const uint32_t ib4 = w0 & 0x03fffffcu, nlb15 = (w1 >> 6) | 0xfc000000u;
    uint32_t t[4]; flat4_link_nibbles(w0, w1, t);
    c[0] = (leafm & 1u) ? (int32_t)(nlb15 + 15u) : (int32_t)(w0 & 0x03ffffffu);
    for (int k = 1; k < 4; k++) c[k] = (int32_t)((((leafm >> k) & 1u) ? nlb15 : ib4) + t[k]);
}
// the two link words of a node from its explicit links (child: node index * 4, ~first entry or 0x76543210), the entry count of every leaf child and the slab flags of the inner ones;
// false when the node or entry numbers do not fit
inline bool flat4_encode_links(const int32_t child[4], const uint32_t counts[4], uint32_t slab_flags, uint32_t links[2]) {
    uint32_t inner_base4 = 0, leaf_base = 0; bool have_inner = false, have_leaf = false;
    uint32_t t[4] = { 0, 0, 0, 0 }, ni = 0, nl = 0;
    for (int k = 0; k < 4; k++) {
        if (child[k] == 0x76543210) continue;
        if (child[k] >= 0) {
            if (!have_inner) { inner_base4 = (uint32_t)child[k]; have_inner = true; }
            if ((uint32_t)child[k] != inner_base4 + 4u * ni) return false;        // inner children are consecutive nodes
            t[k] = (ni << 2) | ((slab_flags >> k) & 1u); ni++;
        } else {
            if (!have_leaf) { leaf_base = (uint32_t)~child[k]; have_leaf = true; }
            if ((uint32_t)~child[k] != leaf_base + nl || counts[k] < 1 || counts[k] > 4) return false;   // leaf children own consecutive entries
            t[k] = 15u - nl; nl += counts[k];
        }
    }
    if (inner_base4 >= (1u << 26) || (inner_base4 & 3u) || (uint64_t)leaf_base + 15u >= (1ull << 26)) return false;
    links[0] = inner_base4 | (child[0] >= 0 && child[0] != 0x76543210 ? (t[0] & 1u) : 0u) | (t[1] << 26) | ((t[3] & 3u) << 30);
    links[1] = (t[3] >> 2) | (t[2] << 2) | ((~(leaf_base + 15u)) << 6);
    return true;
}

// F4.  Plane-major so that a lane picks the near / far plane of all four children by ADDRESS (the sign of its ray direction
// selects lo or hi), not by four selects per axis: lo_x[4] hi_x[4] lo_y[4] hi_y[4] lo_z[4] hi_z[4] child[4] pad[4].
// A missing child has an inverted box (lo = +FLT_MAX, hi = -FLT_MAX) and is never entered.
struct flat4f_node {
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];
    int32_t child[4];      // >= 0: node index * 8 (float4 units); < 0: ~firstLeafEntry; 0x76543210: none
    uint32_t pad[4];
};
static_assert(sizeof(flat4f_node) == 128, "fp32 wide node is one 128-B L2 line");

struct flat_scene {
    int format = kFlatQ4;
    std::vector<flat8_node> nodes_q8;     // kFlatQ8; node 0 is the root
    std::vector<flat4_node> nodes;        // kFlatQ4; node 0 is the root
    std::vector<flat4f_node> nodes_f4;    // kFlatF4
    std::vector<ctl_bvh_node> nodes_f2;   // kFlatF2 (child >= 0: node index * 4)
    std::vector<flat_leaf> leaves;
    int max_depth = 0;                    // of the stored tree
    bool compact_links = true;            // Q4: every node's implied links (flat4_node::links) are valid; false -> the kernels read child[]
    std::vector<int32_t> child_links;     // Q4: 4 explicit links per node (as flat4_node::child), host side only
    bool root_slab = false;               // Q4: the root node itself carries a slab (single-node trees)
    size_t slab_nodes = 0;                // Q4: nodes that carry a slab
    size_t split_refs = 0;                // references added by early split clipping (flatten.cpp; a triangle with k references has k leaf entries)
    size_t node_bytes() const { return nodes.size() * sizeof(flat4_node) + nodes_f4.size() * sizeof(flat4f_node) + nodes_f2.size() * sizeof(ctl_bvh_node) + nodes_q8.size() * sizeof(flat8_node); }
    int stack_need() const { return (format == kFlatF2 || format == kFlatQ8) ? max_depth + 2 : 3 * max_depth + 2; }   // traversal-stack entries a ray can need (Q8: one sibling group per level)
};

// false when the scene has no triangles or more than `max_triangles` instanced triangles
bool flatten_scene(const ctl_scene_desc& d, flat_scene& out, size_t max_triangles, int format = kFlatQ4);
int default_flat_format();   // kFlatQ4 unless $CTL_FLAT_FORMAT says q4 / f4 / f2 (measurement knob)

}  // namespace ctl
