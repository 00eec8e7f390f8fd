// flatten.h — world-space single-level re-layout of a two-level scene (see flatten.cpp).
#pragma once
#include "../../include/ctl_amd.h"
#include <vector>
#include <cstddef>

namespace ctl {

struct flat_leaf { float a[4], b[4], c[4]; uint32_t index; uint32_t node; uint32_t pad[2]; };   // 64 B: Woop rows + (globalTri << 1 | last) + node
static_assert(sizeof(flat_leaf) == 64, "flat leaf entry is one 64-B fetch group");

// 4-wide node with child boxes quantised to 8 bits relative to the node's own box: 64 B = ONE fetch group per visit for
// four children (the reference's BVH2 node spends the same 64 B on two).  Child box c, axis k:
//   lo = origin[k] + 2^e[k] * qlo[k][c],  hi = origin[k] + 2^e[k] * qhi[k][c]   (conservative: lo <= true lo, hi >= true hi)
struct flat4_node {
    float origin[3];
    uint8_t e[3];          // biased float exponents of the per-axis quantisation step (the step is 2^(e-127))
    uint8_t mask;          // bit c set <=> child c exists
    uint32_t qlo_x, qhi_x, qlo_y, qhi_y, qlo_z, qhi_z;   // byte c = child c
    int32_t child[4];      // >= 0: node index * 4 (float4 units); < 0: ~firstLeafEntry
    uint32_t pad[2];
};
static_assert(sizeof(flat4_node) == 64, "wide node is one 64-B fetch group");

// 8-wide node, 128 B = ONE L2 line per visit.  tools/gather_probe.hip: an MI355X gathers ~56-60 G random records/s from HBM
// (95-127 G/s from L2) whether a record is 16, 64 or 128 bytes — traversal time is the NUMBER of records fetched, so a node
// should fill the line it costs.  Eight children per visit need ~20 node fetches per ray where the 4-wide node needs ~34.
// Child c, axis k: lo = origin[k] + 2^(e[k]-127) * qlo[k][c] (conservative, as flat4_node).  96 B are used; the device loads 6 float4.
struct flat8_node {
    float origin[3];
    uint8_t e[3];
    uint8_t mask;                  // bit c set <=> child c exists
    uint32_t qlo_x[2], qhi_x[2];   // byte c of the 8-byte group = child c
    uint32_t qlo_y[2], qhi_y[2];
    uint32_t qlo_z[2], qhi_z[2];
    int32_t child[8];              // >= 0: node index * 8 (float4 units); < 0: ~firstLeafEntry
    uint32_t pad[8];
};
static_assert(sizeof(flat8_node) == 128, "8-wide node is one 128-B line");

struct flat_scene {
    int width = 4;                     // 4: nodes, 8: nodes8
    std::vector<flat8_node> nodes8;
    std::vector<flat4_node> nodes;     // node 0 is the root
    std::vector<flat_leaf> leaves;
    int max_depth = 0;                 // of the 4-wide tree
};

// false when the scene has no triangles or more than `max_triangles` instanced triangles
bool flatten_scene(const ctl_scene_desc& d, flat_scene& out, size_t max_triangles, int width = 8);

}  // namespace ctl
