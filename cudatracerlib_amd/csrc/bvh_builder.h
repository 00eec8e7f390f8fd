// bvh_builder.h — host BVH2 construction producing the reference's node encoding
// (Engine/TriIntersectorData.h:42-117 node layout; child encoding of SplitBVHBuilder.cpp:163-203:
//  inner child = nodeIndex*4 (float4 units), leaf = ~firstLeafEntry, missing child = 0x76543210).
// The tree itself is this project's own binned-SAH build (object splits, 32 bins, parallel over subtrees);
// the closest hit of a ray does not depend on the tree shape.
#pragma once
#include "../../include/ctl_amd.h"
#include <vector>
#include <cstdint>

namespace ctl {

struct aabb {
    float lo[3], hi[3];
    void reset() { for (int i = 0; i < 3; i++) { lo[i] = 3.402823466e+38f; hi[i] = -3.402823466e+38f; } }
    void grow(const float* p) { for (int i = 0; i < 3; i++) { if (p[i] < lo[i]) lo[i] = p[i]; if (p[i] > hi[i]) hi[i] = p[i]; } }
    void grow(const aabb& b) { for (int i = 0; i < 3; i++) { if (b.lo[i] < lo[i]) lo[i] = b.lo[i]; if (b.hi[i] > hi[i]) hi[i] = b.hi[i]; } }
    float area() const { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return 2.0f * (dx * dy + dy * dz + dz * dx); }
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
};

struct bvh_result {
    std::vector<ctl_bvh_node> nodes;     // node 0 is the root
    std::vector<uint32_t> leaf_prims;    // primitive ids in leaf order; a leaf ~k covers leaf_prims[k..] up to its last flag
    std::vector<uint8_t> leaf_last;      // 1 on the last entry of each leaf
    int root;                            // 0 for an inner root, ~0 when the whole set is one leaf and wrap_single_leaf == false
    int max_depth;
};

// prim_boxes: one box per primitive.  max_leaf: 8 for meshes (BVHBuilderHelper.cpp:119), 1 for the scene BVH
// (BVHRebuilder.cpp:380-383).  wrap_single_leaf: emit the (leaf, 0x76543210) root node the reference emits for a
// one-leaf mesh (SplitBVHBuilder.cpp:176-189).  max_depth_limit bounds the traversal stack.
// reinsertion_passes > 0: that many passes of insertion-based re-optimisation over the finished tree (bvh_builder.cpp reinserter), each over the largest
// `reinsertion_fraction` of the nodes; a result deeper than max_depth_limit is discarded.
void build_bvh(const std::vector<aabb>& prim_boxes, int max_leaf, bool wrap_single_leaf, int max_depth_limit, bvh_result& out, float node_cost = 1.0f,
               int reinsertion_passes = 0, float reinsertion_fraction = 1.0f);

// The reference's mesh build restated (sbvh_builder.cpp): SplitBVHBuilder with spatial splits over the triangles
// (positions[3 * n_vert], indices[3 * n_tri] or nullptr for a soup).  leaf_prims may name a triangle more than once.
void build_sbvh(const float* positions, const uint32_t* indices, uint32_t n_tri, int max_leaf, bvh_result& out);

} // namespace ctl

namespace ctl {
// 4-wide node obtained by collapsing the BVH2.  mode 0: greedy (repeatedly open the inner child with the largest surface area).  mode 1: SAH-optimal dynamic
// programme over (subtree, slots) with `node_cost` per wide-node visit and 1 per leaf entry; it may also merge a subtree of <= max_leaf primitives into ONE
// leaf, which rewrites R.leaf_last (the primitives of a subtree are consecutive in R.leaf_prims).
// child >= 0: index into the wide-node array; child < 0: ~firstLeafEntry (entries of R.leaf_prims up to the next leaf_last flag); n = children used.
struct wide4_node { aabb box; aabb cbox[4]; int child[4]; int n; };
void collapse_bvh4(bvh_result& R, std::vector<wide4_node>& out, int& max_depth, int mode = 1, float node_cost = 0.75f, int max_leaf = 4);
// 8-wide node for the Q8 format (flat8.h): greedy collapse (open the inner child with the largest surface area until eight slots are used), then the children are
// placed into OCTANT-ORDERED slots: slot s stands for the sign vector (bit k set = "+" on axis k), and the children go to the slots that maximise
// sum over children of  sign(s) . (child centroid - node centroid)  (greedy matching) — a ray then visits the slots in decreasing (s ^ octinv).  child[s] as in wide4_node, 0x76543210 = empty slot.
struct wide8_node { aabb box; aabb cbox[8]; int child[8]; };
void collapse_bvh8(const bvh_result& R, std::vector<wide8_node>& out, int& max_depth);
} // namespace ctl
