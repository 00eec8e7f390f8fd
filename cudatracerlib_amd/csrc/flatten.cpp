// flatten.cpp — optional re-layout at scene upload: ONE world-space BVH over all instanced triangles ("flattening
// static instances into one BVH is allowed by the API and likely necessary").  The MI355X has 288 GB of HBM: 128 B per instanced
// triangle buys a traversal that never restarts at a mesh root per instance and has two instead of four kinds of work per wave.
// The tree only culls: a leaf entry keeps the mesh's own object-space Woop rows, its node id and the node's inverse transform, and the
// kernel evaluates it with the reference's instance-transform + Woop arithmetic, so (t, u, v, triangle, node) equal the two-level traversal bit for bit
// (flatten.h).  World-space vertices are recovered from the Woop rows in double only to compute the boxes, which are padded.
#include "flatten.h"
#include "knobs.h"
#include "flat_slab.h"
#include "bvh_builder.h"
#include "scene_cache.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace ctl {

namespace {

// builder settings of the world-space BVH; the environment overrides are measurement knobs (tools/flat_build_probe.sh).  SAH node cost 0.5:
// a 4-wide node covers two levels of the binary tree in one fetch + one loop iteration, a leaf entry costs one of each per triangle.
// Measured on synthetic-SM: node cost 1 -> 34.0 nodes + 8.45 triangles per ray, 0.5 -> 35.3 + 6.28 and +1 % rays/s; leaf size 2 / 4 / 8: no difference
int flat_max_leaf() { static const int v = [] { const char* e = knob_env("CTL_FLAT_MAX_LEAF"); const int x = e ? atoi(e) : 0; return x >= 1 && x <= 16 ? x : 4; }(); return v; }
// collapse of the binary tree into 4-wide nodes: 1 = SAH-optimal dynamic programme (bvh_builder.h), 0 = greedy; its node cost is in leaf-entry tests:
// k_intersect spends ~277 lane-slots on a node step and ~365 on a leaf-entry step at its measured lane utilisation
int flat_collapse_mode() { static const int v = [] { const char* e = knob_env("CTL_FLAT_COLLAPSE"); return e ? atoi(e) : 0; }(); return v; }
float flat_collapse_node_cost() { static const float v = [] { const char* e = knob_env("CTL_FLAT_COLLAPSE_NODE_COST"); const float x = e ? (float)atof(e) : 0.0f; return x > 0.0f ? x : 0.75f; }(); return v; }
// early split clipping (flatten_scene): longest side a triangle reference may have, in medians of the scene's triangle boxes (0 = off); $CTL_FLAT_SPLIT
float flat_split_ratio() { static const float v = [] { const char* e = knob_env("CTL_FLAT_SPLIT"); const float x = e ? (float)atof(e) : -1.0f; return x >= 0.0f ? x : 4.0f; }(); return v; }   // with the gain rule below, on the GPU (profiles/r04_split_clipping.log): 4: synthetic-sm-hard 3381 Mrays/s (no clipping: 1383), 8: 3316, 16: 3203; synthetic-SM and -bathroom unchanged
// ... and a split is kept only where the two parts' boxes have less than this fraction of the part's box surface (1 = always); $CTL_FLAT_SPLIT_GAIN
float flat_split_gain() { static const float v = [] { const char* e = knob_env("CTL_FLAT_SPLIT_GAIN"); const float x = e ? (float)atof(e) : -1.0f; return x > 0.0f ? x : 0.65f; }(); return v; }   // 0.6 .. 0.7 equal; 0.8 .. 1.0 also split axis-aligned floors and walls: synthetic-SM - 7 %, synthetic-sm-hard 2800 .. 2960
// insertion-based re-optimisation of the BVH2 before the collapse (bvh_builder.cpp reinserter): passes, and the share of the nodes (largest first) a pass visits
int flat_reinsert_passes() { static const int v = [] { const char* e = knob_env("CTL_FLAT_REINSERT"); const int x = e ? atoi(e) : -1; return x >= 0 && x <= 64 ? x : 16; }(); return v; }
float flat_reinsert_fraction() { static const float v = [] { const char* e = knob_env("CTL_FLAT_REINSERT_FRACTION"); const float x = e ? (float)atof(e) : 0.0f; return x > 0.0f && x <= 1.0f ? x : 0.03f; }(); return v; }
int flat_slot_order() { static const int v = [] { const char* e = knob_env("CTL_FLAT_SLOT_ORDER"); const int x = e ? atoi(e) : -1; return x >= 0 && x <= 2 ? x : 0; }(); return v; }
float flat_node_cost() { static const float v = [] { const char* e = knob_env("CTL_FLAT_NODE_COST"); const float x = e ? (float)atof(e) : 0.0f; return x > 0.0f ? x : 0.5f; }(); return v; }

// 4x4 inverse in double (cofactor expansion)
bool inv4(const double m[16], double out[16]) {
    double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0 || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * id;
    return true;
}

// vertices of a Woop triangle (TriIntersectorData::getData, Engine/TriIntersectorData.cu:20-32) in double
bool woop_vertices(const ctl_woop_tri& w, double v[3][3]) {
    const double m[16] = { w.b[0], w.b[1], w.b[2], w.b[3], w.c[0], w.c[1], w.c[2], w.c[3], w.a[0], w.a[1], w.a[2], -(double)w.a[3], 0, 0, 0, 1 };
    double inv[16];
    if (!inv4(m, inv)) return false;
    for (int k = 0; k < 3; k++) { v[2][k] = inv[k * 4 + 3]; v[0][k] = v[2][k] + inv[k * 4 + 0]; v[1][k] = v[2][k] + inv[k * 4 + 1]; }
    return true;
}
// world-space length that one unit of object-space round-off of this triangle can reach: its largest object-space coordinate times the
// largest row sum of the instance's linear part
double woop_slack(const double v[3][3], const float* M) {
    double m = 0; for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) m = std::max(m, std::fabs(v[j][k]));
    double rs = 0; for (int r = 0; r < 3; r++) rs = std::max(rs, std::fabs((double)M[r * 4]) + std::fabs((double)M[r * 4 + 1]) + std::fabs((double)M[r * 4 + 2]));
    return m * rs;
}
// phase timing on stderr when CTL_VERBOSE is set
struct phase_timer {
    const bool on = std::getenv("CTL_VERBOSE") != nullptr; std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) { if (!on) return; const auto n = std::chrono::steady_clock::now(); std::fprintf(stderr, "[ctl flatten] %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count()); t = n; }
};
// static split of [0, n) over the host threads
template <typename F> void parallel_for(size_t n, const F& f, size_t min_chunk = 4096) {
    const size_t nt = std::min<size_t>(std::min(64u, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, n / min_chunk));
    if (nt <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++) th.emplace_back([&, t]() { f(n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
}
// child links and leaf ranges of a tree read back from the cache: every link must land inside the arrays before they reach the GPU
bool flat_links_valid(const flat_scene& F) {
    const size_t nl = F.leaves.size();
    auto ok = [&](int32_t c, size_t n_nodes, int unit) {
        if (c == 0x76543210) return true;
        if (c >= 0) return c % unit == 0 && (size_t)(c / unit) < n_nodes;
        return (size_t)(~c) < nl;
    };
    if (F.format == kFlatQ4) {
        if (F.child_links.size() != F.nodes.size() * 4) return false;
        for (size_t i = 0; i < F.nodes.size(); i++) {
            const flat4_node& n = F.nodes[i];
            int32_t imp[4]; if (F.compact_links) flat4_implied_links(n, imp);
            for (int c = 0; c < 4; c++) {
                const int32_t k = F.child_links[i * 4 + c];
                if (((n.mask >> c) & 1) && !ok(k, F.nodes.size(), 4)) return false;
                // the kernels follow the IMPLIED links (and the slab flag they carry) when the tree is compact: they must be the explicit ones
                if (F.compact_links && ((n.mask >> c) & 1) && (imp[c] & ~(k >= 0 ? 1 : 0)) != k) return false;
                if (!F.compact_links && ((n.mask >> c) & 1) && n.child[c] != k) return false;
                // a slot without a child repeats the link of one that exists (flatten_scene's last pass): a link the kernels may follow, so it is checked like the others
                if (!((n.mask >> c) & 1)) {
                    const int32_t mine = F.compact_links ? imp[c] : n.child[c]; bool found = false;
                    for (int e = 0; e < 4; e++) if ((n.mask >> e) & 1) { const int32_t theirs = F.compact_links ? imp[e] : n.child[e]; if (mine == theirs || (mine >= 0 && theirs >= 0 && (mine & ~3) == (theirs & ~3))) found = true; }
                    if (!found && (n.mask & 15)) return false;
                }
            }
        }
    }
    else if (F.format == kFlatQ8) {
        // every link the kernels derive (flat8.h) must land inside the arrays, and must be the explicit one host code sees
        if (F.child_links.size() != F.nodes_q8.size() * 8 || F.nodes_q8.size() >= kFlat8MaxNodes) return false;
        for (size_t i = 0; i < F.nodes_q8.size(); i++) {
            const flat8_node& n = F.nodes_q8[i];
            const uint32_t q0w = (uint32_t)n.e[0] | ((uint32_t)n.e[1] << 8) | ((uint32_t)n.e[2] << 16) | ((uint32_t)n.imask << 24);
            const uint32_t im = flat8_inner_mask(q0w), lm = flat8_leaf_mask(q0w, n.base_b);
            for (uint32_t s = 0; s < 8; s++) {
                const int32_t k = F.child_links[i * 8 + s];
                if ((im >> s) & 1u) { const uint32_t c = flat8_child_node(n.base_b, im, s); if (c >= F.nodes_q8.size() || k != (int32_t)c) return false; }
                else if ((lm >> s) & 1u) { const uint32_t e = flat8_leaf_entry(n.leaf_base, lm, s); if (e >= nl || k != ~(int32_t)e || !(F.leaves[e].index & 1u)) return false; }
                else if (k != (int32_t)kFlat8None) return false;
            }
            // an empty slot that round-off lets through is read as entry leaf_base + rank: one past the node's own entries at most, which must exist (the upload appends a closing entry)
            if ((size_t)n.leaf_base + flat8_popc(lm) > nl) return false;
        }
    }
    else if (F.format == kFlatF4) { for (const auto& n : F.nodes_f4) for (int c = 0; c < 4; c++) if (!ok(n.child[c], F.nodes_f4.size(), 8)) return false; }
    else for (const auto& n : F.nodes_f2) if (!ok(n.child0, F.nodes_f2.size(), 4) || !ok(n.child1, F.nodes_f2.size(), 4)) return false;
    if (F.node_bytes() == 0) return false;
    return nl > 0 && (F.leaves[nl - 1].index & 1u);   // the last entry closes its leaf
}
struct slab_ctri { double w[3][3]; double slack; int c; };   // a triangle under child c of the node whose slab is being chosen
float round_down(double x) { float f = (float)x; return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f; }
float round_up(double x) { float f = (float)x; return ((double)f < x) ? std::nextafterf(f, INFINITY) : f; }

}  // namespace

int default_flat_format() {
    static const int v = [] {
        const char* e = knob_env("CTL_FLAT_FORMAT");
        if (e && (!std::strcmp(e, "f4") || !std::strcmp(e, "F4"))) return (int)kFlatF4;
        if (e && (!std::strcmp(e, "f2") || !std::strcmp(e, "F2"))) return (int)kFlatF2;
        if (e && (!std::strcmp(e, "q8") || !std::strcmp(e, "Q8"))) return (int)kFlatQ8;
        return (int)kFlatQ4;
    }();
    return v;
}

bool flatten_scene(const ctl_scene_desc& d, flat_scene& out, size_t max_triangles, int format) {
    out.nodes.clear(); out.nodes_q8.clear(); out.nodes_f4.clear(); out.nodes_f2.clear(); out.leaves.clear(); out.compact_links = true; out.split_refs = 0;
    out.format = (format == kFlatF4 || format == kFlatF2 || format == kFlatQ8) ? format : kFlatQ4;
    phase_timer pt;
    // leaf-entry range of every mesh (the woop stream is shared; a mesh ends where the next one starts)
    std::vector<std::pair<uint32_t, uint32_t>> starts;
    for (uint32_t m = 0; m < d.n_meshes; m++) starts.emplace_back(d.meshes[m].bvh_tri_offset / 3, m);
    std::sort(starts.begin(), starts.end());
    std::vector<uint32_t> mesh_first(d.n_meshes), mesh_last(d.n_meshes);
    for (size_t r = 0; r < starts.size(); r++) { mesh_first[starts[r].second] = starts[r].first; mesh_last[starts[r].second] = (r + 1 < starts.size()) ? starts[r + 1].first : d.n_woop; }
    // unique triangles per mesh: (mesh-local triangle id -> one leaf entry); spatial splits reference a triangle from several leaves
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> mesh_tris(d.n_meshes);
    size_t total = 0;
    for (uint32_t k = 0; k < d.n_nodes; k++) {
        const uint32_t m = d.nodes[k].mesh_index;
        if (mesh_tris[m].empty()) {
            auto& v = mesh_tris[m];
            for (uint32_t w = mesh_first[m]; w < mesh_last[m]; w++) v.emplace_back(d.woop_index[d.meshes[m].bvh_index_offset + (w - mesh_first[m])].index >> 1, w);
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first == b.first; }), v.end());
        }
        total += mesh_tris[m].size();
    }
    if (total == 0 || total > max_triangles) return false;
    // flattened-BVH cache (scene_cache.h): the result depends on the leaf streams, the instance list and the node format only
    std::string key;
    if (!cache_dir().empty()) {
        content_hash H; const uint32_t version = 17;
        H.add_value(version); { const char* e = knob_env("CTL_FLAT_SLAB_SUBTREE"); H.add_value(e ? atoi(e) : kSlabSubtreeDefault); } { const char* e = knob_env("CTL_FLAT_FORCE_EXPLICIT"); H.add_value(e ? atoi(e) : 0); } H.add_value(flat_collapse_mode()); { const char* e = knob_env("CTL_FLAT_SLAB_USEFUL"); H.add_value(e ? atof(e) : 0.6); } H.add_value(flat_collapse_node_cost()); { const char* e = knob_env("CTL_FLAT_BFS_TOP"); H.add_value(e ? atol(e) : 65536L); } H.add_value((int)sizeof(flat_leaf)); H.add_value(flat_max_leaf()); H.add_value(flat_node_cost()); H.add_value(flat_split_ratio()); H.add_value(flat_split_gain()); H.add_value(flat_reinsert_passes()); H.add_value(flat_reinsert_fraction()); H.add_value(flat_slot_order()); H.add_value(out.format); H.add_value(d.n_meshes); H.add_value(d.n_nodes); H.add_value(d.n_woop);
        H.add(d.woop, (size_t)d.n_woop * sizeof(ctl_woop_tri)); H.add(d.woop_index, (size_t)d.n_woop * sizeof(ctl_woop_index));
        H.add(d.meshes, (size_t)d.n_meshes * sizeof(ctl_kernel_mesh));
        for (uint32_t k = 0; k < d.n_nodes; k++) { H.add_value(d.nodes[k].mesh_index); H.add(d.node_transforms[k].m, 64); }
        key = H.hex();
        cache_reader rd("flat", key);
        int c_format = -1, c_depth = 0, c_compact = 0, c_root_slab = 0; uint64_t c_slab_nodes = 0, c_split_refs = 0;
        if (rd.found() && rd.value(c_format) && rd.value(c_depth) && rd.value(c_compact) && rd.vector(out.nodes) && rd.vector(out.nodes_q8) && rd.vector(out.nodes_f4) && rd.vector(out.nodes_f2) && rd.vector(out.leaves) && rd.vector(out.child_links) && rd.value(c_root_slab) && rd.value(c_slab_nodes) && rd.value(c_split_refs) && rd.verify() &&
            c_format == out.format && ((out.compact_links = c_compact != 0), flat_links_valid(out))) {
            out.max_depth = c_depth; out.root_slab = c_root_slab != 0; out.slab_nodes = (size_t)c_slab_nodes; out.split_refs = (size_t)c_split_refs; pt.lap("cache hit");
            return true;
        }
        out.nodes.clear(); out.nodes_q8.clear(); out.nodes_f4.clear(); out.nodes_f2.clear(); out.leaves.clear(); out.child_links.clear(); out.compact_links = true;
    }
    struct wtri { uint32_t tri, node, woop, local; };   // local: index into mesh_local[mesh of node] (a triangle split below is referenced by several entries)
    // object-space vertices of every mesh's triangles once (degenerate ones can never be hit and are dropped), then per node in parallel
    struct ltri { double v[3][3]; uint32_t tri, woop; };
    std::vector<std::vector<ltri>> mesh_local(d.n_meshes);
    for (uint32_t m = 0; m < d.n_meshes; m++) {
        mesh_local[m].reserve(mesh_tris[m].size());
        for (auto& e : mesh_tris[m]) { ltri l; l.tri = e.first; l.woop = e.second; if (woop_vertices(d.woop[e.second], l.v)) mesh_local[m].push_back(l); }
    }
    std::vector<size_t> node_first(d.n_nodes + 1, 0);
    for (uint32_t k = 0; k < d.n_nodes; k++) node_first[k + 1] = node_first[k] + mesh_local[d.nodes[k].mesh_index].size();
    std::vector<wtri> tris(node_first[d.n_nodes]);
    std::vector<aabb> boxes(node_first[d.n_nodes]);
    if (tris.empty()) return false;
    parallel_for(d.n_nodes, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            const ctl_node& N = d.nodes[k]; const ctl_kernel_mesh& km = d.meshes[N.mesh_index];
            const float* M = d.node_transforms[k].m;
            size_t o = node_first[k];
            for (const ltri& l : mesh_local[N.mesh_index]) {
                wtri& t = tris[o]; t.tri = km.tri_offset + l.tri; t.node = (uint32_t)k; t.woop = l.woop; t.local = (uint32_t)(o - node_first[k]);
                aabb& b = boxes[o]; b.reset(); o++;
                for (int j = 0; j < 3; j++) {
                    for (int r = 0; r < 3; r++) {
                        const double w = (double)M[r * 4] * l.v[j][0] + (double)M[r * 4 + 1] * l.v[j][1] + (double)M[r * 4 + 2] * l.v[j][2] + (double)M[r * 4 + 3];
                        const float lo = round_down(w), hi = round_up(w); if (lo < b.lo[r]) b.lo[r] = lo; if (hi > b.hi[r]) b.hi[r] = hi;
                    }
                }
                // The kernel decides a hit with the fp32 object-space Woop test, whose accepted region differs from the exact triangle by
                // round-off: pad the box by a few units in the last place of its largest coordinate (and of its extent) — and of the triangle's
                // OBJECT-space magnitude carried through the instance transform (a mesh far from its own origin, a strongly scaled instance: the round-off
                // of the object-space test scales with those, not with the world box)
                const float off = (float)woop_slack(l.v, M);
                for (int r = 0; r < 3; r++) {
                    const float mag = std::max(std::max(std::max(std::fabs(b.lo[r]), std::fabs(b.hi[r])), b.hi[r] - b.lo[r]), off);
                    const float e = mag * 9.5367431640625e-7f + 1e-30f;   // 8 ulp
                    b.lo[r] -= e; b.hi[r] += e;
                }
            }
        }
    }, 1);
    pt.lap("world triangles");
    // Early split clipping: a triangle much larger than its neighbours (a floor under two thousand instances, a beam across a hall of small triangles) would otherwise sit in a leaf
    // near the root whose box every ray crosses.  Such a triangle is entered as several REFERENCES — the same entry, each with the box of the part of the triangle inside one cell of a
    // recursive spatial-median subdivision of its box (exact polygon clipping in double, rounded outwards, padded like the whole triangle) — until no reference is longer than
    // flat_split_ratio() x the MEDIAN length of the scene's triangle boxes: the scene's own scale, so a beam through fine geometry ends up in finer pieces than a floor under coarse
    // one.  The total is held to + 30 % references by doubling that length.  The traversal tests the whole triangle wherever a reference leads it (same hit, same bits; a second
    // encounter of the closest hit does not pass t < t_hit).  Measured with the oracle's counting traversal (tools/bvh_quality_probe.py): DESIGN.md §3.
    if (flat_split_ratio() > 0.0f) {
        const size_t n0 = boxes.size(), budget = n0 * 3 / 10 + 64;
        auto longest = [](const aabb& b) { return std::max(std::max(b.hi[0] - b.lo[0], b.hi[1] - b.lo[1]), b.hi[2] - b.lo[2]); };
        std::vector<float> ext(n0);
        for (size_t g = 0; g < n0; g++) ext[g] = longest(boxes[g]);
        std::nth_element(ext.begin(), ext.begin() + n0 / 2, ext.end());
        const double median = ext[n0 / 2];
        struct piece { aabb box; uint32_t src; };
        std::vector<piece> extra; std::vector<aabb> first;   // first[i]: the box that replaces boxes[big[i]]
        std::vector<uint32_t> big;
        double side = 0.0;   // longest side of the scene: references are never made shorter than 1 / 4096 of it (a scene whose median triangle is a point)
        { aabb sb; sb.reset(); for (const aabb& b : boxes) for (int r = 0; r < 3; r++) { sb.lo[r] = std::min(sb.lo[r], b.lo[r]); sb.hi[r] = std::max(sb.hi[r], b.hi[r]); } side = longest(sb); }
        for (double lmax = std::max(std::max(median * flat_split_ratio(), side / 4096.0), 1e-30);; lmax *= 2.0) {
            extra.clear(); first.clear(); big.clear();
            bool over = false;
            for (size_t g = 0; g < n0 && !over; g++) {
                const aabb& b = boxes[g];
                if (longest(b) <= lmax) continue;
                const wtri& t = tris[g]; const ltri& l = mesh_local[d.nodes[t.node].mesh_index][t.local]; const float* M = d.node_transforms[t.node].m;
                double w[3][3];
                for (int j = 0; j < 3; j++) for (int r = 0; r < 3; r++) w[j][r] = (double)M[r * 4] * l.v[j][0] + (double)M[r * 4 + 1] * l.v[j][1] + (double)M[r * 4 + 2] * l.v[j][2] + (double)M[r * 4 + 3];
                const float off = (float)woop_slack(l.v, M);
                struct poly { double p[9][3]; int n; double lo[3], hi[3]; };
                std::vector<poly> todo(1), done;
                poly& P0 = todo[0]; P0.n = 3;
                for (int j = 0; j < 3; j++) for (int r = 0; r < 3; r++) P0.p[j][r] = w[j][r];
                for (int r = 0; r < 3; r++) { P0.lo[r] = std::min(std::min(w[0][r], w[1][r]), w[2][r]); P0.hi[r] = std::max(std::max(w[0][r], w[1][r]), w[2][r]); }
                while (!todo.empty()) {
                    poly P = todo.back(); todo.pop_back();
                    int ax = 0; for (int r = 1; r < 3; r++) if (P.hi[r] - P.lo[r] > P.hi[ax] - P.lo[ax]) ax = r;
                    if (P.hi[ax] - P.lo[ax] <= lmax || P.n > 7 || done.size() + todo.size() >= 16384) { done.push_back(P); continue; }
                    const double mid = 0.5 * (P.lo[ax] + P.hi[ax]);
                    const size_t todo_before = todo.size();
                    for (int side_k = 0; side_k < 2; side_k++) {   // Sutherland-Hodgman against x[ax] <= mid / >= mid
                        poly Q; Q.n = 0;
                        for (int i = 0; i < P.n; i++) {
                            const double* a = P.p[i]; const double* c = P.p[(i + 1) % P.n];
                            const bool ia = side_k ? a[ax] >= mid : a[ax] <= mid, ic = side_k ? c[ax] >= mid : c[ax] <= mid;
                            if (ia) { for (int r = 0; r < 3; r++) Q.p[Q.n][r] = a[r]; Q.n++; }
                            if (ia != ic) { const double u = (mid - a[ax]) / (c[ax] - a[ax]); for (int r = 0; r < 3; r++) Q.p[Q.n][r] = r == ax ? mid : a[r] + u * (c[r] - a[r]); Q.n++; }
                        }
                        if (Q.n < 3) continue;
                        for (int r = 0; r < 3; r++) { Q.lo[r] = Q.hi[r] = Q.p[0][r]; for (int i = 1; i < Q.n; i++) { Q.lo[r] = std::min(Q.lo[r], Q.p[i][r]); Q.hi[r] = std::max(Q.hi[r], Q.p[i][r]); } }
                        // the interpolated corners carry round-off of the order of the triangle's size: keep a part's box inside the parent part's box, and widen it by that round-off
                        for (int r = 0; r < 3; r++) { const double e = 4e-16 * (std::fabs(P.lo[r]) + std::fabs(P.hi[r]) + (P.hi[r] - P.lo[r])); Q.lo[r] = std::max(P.lo[r], Q.lo[r] - e); Q.hi[r] = std::min(P.hi[r], Q.hi[r] + e); }
                        todo.push_back(Q);
                    }
                    // keep the split only where it removes empty space: the two parts' boxes together must have less than flat_split_gain() x the surface of the part's box.  Halving a
                    // beam that crosses its box diagonally quarters each box (sum 0.5); halving an axis-aligned rectangle of a floor removes nothing (sum 1.0) and only adds references
                    if (todo.size() == todo_before + 2) {
                        auto half_area = [](const poly& B) { const double x = B.hi[0] - B.lo[0], y = B.hi[1] - B.lo[1], z = B.hi[2] - B.lo[2]; return x * y + y * z + z * x; };
                        if (half_area(todo[todo_before]) + half_area(todo[todo_before + 1]) >= flat_split_gain() * half_area(P)) { todo.resize(todo_before); done.push_back(P); }
                    } else if (todo.size() == todo_before + 1) { todo.resize(todo_before); done.push_back(P); }   // everything on one side of the plane (a sliver): no split
                    else if (todo.size() == todo_before) done.push_back(P);
                }
                if (done.size() < 2) continue;
                big.push_back((uint32_t)g);
                for (size_t i = 0; i < done.size(); i++) {
                    aabb pb;
                    for (int r = 0; r < 3; r++) {
                        pb.lo[r] = round_down(done[i].lo[r]); pb.hi[r] = round_up(done[i].hi[r]);
                        const float mag = std::max(std::max(std::max(std::fabs(b.lo[r]), std::fabs(b.hi[r])), b.hi[r] - b.lo[r]), off);   // the WHOLE triangle's padding: the test that accepts a hit is the whole triangle's
                        const float e = mag * 9.5367431640625e-7f + 1e-30f;
                        pb.lo[r] = std::max(b.lo[r], pb.lo[r] - e); pb.hi[r] = std::min(b.hi[r], pb.hi[r] + e);
                    }
                    if (i == 0) first.push_back(pb); else extra.push_back(piece{ pb, (uint32_t)g });
                }
                if (extra.size() > budget) over = true;
            }
            if (!over) break;
        }
        for (size_t i = 0; i < big.size(); i++) boxes[big[i]] = first[i];
        tris.reserve(n0 + extra.size()); boxes.reserve(n0 + extra.size());
        for (const piece& e : extra) { tris.push_back(tris[e.src]); boxes.push_back(e.box); }
        out.split_refs = extra.size();
        pt.lap("split large triangles");
    }
    bvh_result R;
    build_bvh(boxes, out.format == kFlatQ8 ? 1 : flat_max_leaf(), true, 60, R, flat_node_cost(), flat_reinsert_passes(), flat_reinsert_fraction());   // Q8: every leaf slot is ONE entry (flat8.h)
    pt.lap("build BVH2");
    int wdepth = 0;
    if (out.format == kFlatF2) {
        // the binary tree as built, in the reference's node layout; the two children of a node are stored next to each other (one
        // 128-B line) so that a ray entering both pays one line
        std::vector<int> new_id(R.nodes.size(), -1), order; order.reserve(R.nodes.size());
        std::vector<std::pair<int, int>> stack;   // (node, depth)
        new_id[0] = 0; order.push_back(0); stack.emplace_back(0, 1);
        while (!stack.empty()) {
            const auto [me, dep] = stack.back(); stack.pop_back();
            wdepth = std::max(wdepth, dep);
            int kids[2], nk = 0;
            for (int c : { R.nodes[me].child0, R.nodes[me].child1 }) if (c >= 0 && c != 0x76543210) kids[nk++] = c / 4;
            if (nk && (order.size() & 1)) order.push_back(-1);   // children start at an even index: a sibling pair is one 128-B line (the array is line-aligned)
            for (int c = 0; c < nk; c++) { new_id[kids[c]] = (int)order.size(); order.push_back(kids[c]); }
            for (int c = nk - 1; c >= 0; c--) stack.emplace_back(kids[c], dep + 1);
        }
        out.nodes_f2.resize(order.size());
        for (size_t i = 0; i < order.size(); i++) {
            if (order[i] < 0) { std::memset(&out.nodes_f2[i], 0, sizeof(ctl_bvh_node)); out.nodes_f2[i].child0 = out.nodes_f2[i].child1 = 0x76543210; continue; }   // padding
            ctl_bvh_node n = R.nodes[order[i]];
            if (n.child0 >= 0 && n.child0 != 0x76543210) n.child0 = new_id[n.child0 / 4] * 4;
            if (n.child1 >= 0 && n.child1 != 0x76543210) n.child1 = new_id[n.child1 / 4] * 4;
            // a missing child (one-leaf scenes) gets an inverted box, so that it is never entered
            const float big = 3.402823466e+38f;   // a = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y), b = c1 likewise, c = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
            if (n.child0 == 0x76543210) { n.a[0] = n.a[2] = n.c[0] = big; n.a[1] = n.a[3] = n.c[1] = -big; }
            if (n.child1 == 0x76543210) { n.b[0] = n.b[2] = n.c[2] = big; n.b[1] = n.b[3] = n.c[3] = -big; }
            out.nodes_f2[i] = n;
        }
    } else if (out.format == kFlatQ8) {
        // collapse to 8-wide nodes with octant-ordered slots (bvh_builder.h), memory order as for the 4-wide tree: the top breadth-first, depth-first clusters below,
        // the inner children of a node consecutive in slot order
        std::vector<wide8_node> W;
        collapse_bvh8(R, W, wdepth);
        if (W.size() >= kFlat8MaxNodes) return flatten_scene(d, out, max_triangles, kFlatQ4);   // node indices are 24 bits: a tree beyond that is stored 4-wide (whose links turn explicit past its own limits)
        {
            std::vector<int> new_id(W.size(), -1), order; order.reserve(W.size());
            std::vector<int> stack; new_id[0] = 0; order.push_back(0);
            static const size_t bfs_top = [] { const char* e = knob_env("CTL_FLAT_BFS_TOP"); return e ? (size_t)atol(e) : (size_t)65536; }();
            std::vector<int> frontier; frontier.push_back(0);
            for (size_t head = 0; head < frontier.size() && order.size() < bfs_top; head++) {
                const int me = frontier[head]; frontier[head] = -1;
                for (int c = 0; c < 8; c++) if (W[me].child[c] >= 0 && W[me].child[c] != 0x76543210) { const int k = W[me].child[c]; new_id[k] = (int)order.size(); order.push_back(k); frontier.push_back(k); }
            }
            for (size_t i = frontier.size(); i-- > 0;) if (frontier[i] >= 0) stack.push_back(frontier[i]);
            while (!stack.empty()) {
                const int me = stack.back(); stack.pop_back();
                int kids[8], nk = 0;
                for (int c = 0; c < 8; c++) if (W[me].child[c] >= 0 && W[me].child[c] != 0x76543210) kids[nk++] = W[me].child[c];
                for (int c = 0; c < nk; c++) { new_id[kids[c]] = (int)order.size(); order.push_back(kids[c]); }
                for (int c = nk - 1; c >= 0; c--) stack.push_back(kids[c]);
            }
            std::vector<wide8_node> W2(W.size());
            for (size_t i = 0; i < order.size(); i++) { W2[i] = W[order[i]]; for (int c = 0; c < 8; c++) if (W2[i].child[c] >= 0 && W2[i].child[c] != 0x76543210) W2[i].child[c] = new_id[W2[i].child[c]]; }
            W.swap(W2);
        }
        out.nodes_q8.resize(W.size()); out.child_links.assign(W.size() * 8, (int32_t)kFlat8None);
        parallel_for(W.size(), [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                const wide8_node& w = W[i]; flat8_node& f = out.nodes_q8[i];
                std::memset(&f, 0, sizeof(f));
                uint32_t* q[3][2] = { { f.qlo_x, f.qhi_x }, { f.qlo_y, f.qhi_y }, { f.qlo_z, f.qhi_z } };
                uint32_t first_inner = 0; bool have_inner = false, consecutive = true; uint32_t ni = 0;
                for (int c = 0; c < 8; c++) {
                    const int k = w.child[c];
                    if (k == 0x76543210) continue;
                    out.child_links[i * 8 + c] = k;   // leaf links are rewritten below (entries in node order)
                    if (k >= 0) { if (!have_inner) { first_inner = (uint32_t)k; have_inner = true; } if ((uint32_t)k != first_inner + ni) consecutive = false; ni++; f.imask |= (uint8_t)(1u << c); }
                    else f.base_b |= 1u << (24 + c);   // B of a non-inner slot: a leaf
                }
                if (!consecutive) f.base_b = 0xffffffffu;   // cannot happen (memory order above); flat_links_valid refuses the tree
                else f.base_b |= first_inner;
                for (int k = 0; k < 3; k++) {
                    f.origin[k] = w.box.lo[k];
                    const double ext = (double)w.box.hi[k] - (double)w.box.lo[k];
                    int e = 1;
                    if (ext > 0) { int ex; std::frexp(ext / 255.0, &ex); e = ex + 127; if (e < 1) e = 1; if (e > 254) e = 254; }
                    f.e[k] = (uint8_t)e;
                    const double step = std::ldexp(1.0, e - 127);
                    for (int c = 0; c < 8; c++) {
                        long lo = 255, hi = 0;   // empty slot: inverted box, never entered
                        if (w.child[c] != 0x76543210) {
                            lo = (long)std::floor(((double)w.cbox[c].lo[k] - (double)f.origin[k]) / step);
                            hi = (long)std::ceil(((double)w.cbox[c].hi[k] - (double)f.origin[k]) / step);
                            // conservative under the device's fp32 evaluation origin + step * q (one rounding) too
                            while (lo > 0 && (float)((double)f.origin[k] + step * (double)lo) > w.cbox[c].lo[k]) lo--;
                            while (hi < 255 && (float)((double)f.origin[k] + step * (double)hi) < w.cbox[c].hi[k]) hi++;
                            lo = std::min(255L, std::max(0L, lo)); hi = std::min(255L, std::max(0L, hi));
                        }
                        q[k][0][c >> 2] |= (uint32_t)lo << (8 * (c & 3)); q[k][1][c >> 2] |= (uint32_t)hi << (8 * (c & 3));
                    }
                }
                f.slab_lo[0] = f.slab_lo[1] = 0u; f.slab_hi[0] = f.slab_hi[1] = 0xffffffffu;
            }
        });
    } else {
        // collapse to 4-wide nodes
        std::vector<wide4_node> W;
        collapse_bvh4(R, W, wdepth, flat_collapse_mode(), flat_collapse_node_cost(), std::min(flat_max_leaf(), 4));
        if (const int so = flat_slot_order()) {
            // Slot order.  The closest-hit traversal orders the entered children by distance; the any-hit traversal (shadow rays) takes them in SLOT order, so the slot order is its
            // visiting order: 1 = largest box first (the child a random ray most likely meets), 2 = smallest first (measurement), 0 = as the collapse left them
            for (wide4_node& w : W) {
                int idx[4] = { 0, 1, 2, 3 };
                std::stable_sort(idx, idx + w.n, [&](int a, int b) { const float x = w.cbox[a].area(), y = w.cbox[b].area(); return so == 1 ? x > y : x < y; });
                wide4_node t = w;
                for (int k = 0; k < w.n; k++) { w.cbox[k] = t.cbox[idx[k]]; w.child[k] = t.child[idx[k]]; }
            }
        }
        {   // memory order: the inner children of a node sit next to each other, subtrees stay clustered: a ray that enters a node
            // usually enters one or two of its children next, and neighbouring lines share DRAM pages / L2 sets
            std::vector<int> new_id(W.size(), -1), order; order.reserve(W.size());
            std::vector<int> stack; new_id[0] = 0; order.push_back(0);
            // The first 65 536 nodes (4 MiB, one XCD's L2) breadth-first — the top of the tree, which every ray walks, as one contiguous block — and depth-first
            // clusters below that frontier.  Measured on synthetic-SM against the all-depth-first order (2164 / 2159 Mrays/s): 4096 nodes 2183, 32 768: 2173,
            // 262 144: 2179, everything breadth-first 2176 (profiles/r02r_node_order_ab.log).  $CTL_FLAT_BFS_TOP overrides (measurement knob, part of the cache key)
            static const size_t bfs_top = [] { const char* e = knob_env("CTL_FLAT_BFS_TOP"); return e ? (size_t)atol(e) : (size_t)65536; }();
            std::vector<int> frontier; frontier.push_back(0);
            for (size_t head = 0; head < frontier.size() && order.size() < bfs_top; head++) {
                const int me = frontier[head]; frontier[head] = -1;
                for (int c = 0; c < W[me].n; c++) if (W[me].child[c] >= 0) { const int k = W[me].child[c]; new_id[k] = (int)order.size(); order.push_back(k); frontier.push_back(k); }
            }
            for (size_t i = frontier.size(); i-- > 0;) if (frontier[i] >= 0) stack.push_back(frontier[i]);   // not yet expanded, first one on top
            while (!stack.empty()) {
                const int me = stack.back(); stack.pop_back();
                int kids[4], nk = 0;
                for (int c = 0; c < W[me].n; c++) if (W[me].child[c] >= 0) kids[nk++] = W[me].child[c];
                for (int c = 0; c < nk; c++) { new_id[kids[c]] = (int)order.size(); order.push_back(kids[c]); }
                for (int c = nk - 1; c >= 0; c--) stack.push_back(kids[c]);
            }
            std::vector<wide4_node> W2(W.size());
            for (size_t i = 0; i < order.size(); i++) { W2[i] = W[order[i]]; for (int c = 0; c < W2[i].n; c++) if (W2[i].child[c] >= 0) W2[i].child[c] = new_id[W2[i].child[c]]; }
            W.swap(W2);
        }
        if (out.format == kFlatF4) {
            out.nodes_f4.resize(W.size());
            parallel_for(W.size(), [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; i++) {
                    const wide4_node& w = W[i]; flat4f_node& f = out.nodes_f4[i];
                    std::memset(&f, 0, sizeof(f));
                    float* lo[3] = { f.lo_x, f.lo_y, f.lo_z }; float* hi[3] = { f.hi_x, f.hi_y, f.hi_z };
                    for (int c = 0; c < 4; c++) {
                        const bool have = c < w.n;
                        for (int k = 0; k < 3; k++) { lo[k][c] = have ? w.cbox[c].lo[k] : 3.402823466e+38f; hi[k][c] = have ? w.cbox[c].hi[k] : -3.402823466e+38f; }
                        f.child[c] = have ? (w.child[c] >= 0 ? w.child[c] * 8 : w.child[c]) : 0x76543210;
                    }
                }
            });
        } else {
            // quantise the child boxes conservatively against the node's own box (one exponent per axis)
            out.nodes.resize(W.size());
            parallel_for(W.size(), [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                const wide4_node& w = W[i]; flat4_node& f = out.nodes[i];
                std::memset(&f, 0, sizeof(f));
                uint32_t* q[3][2] = { { &f.qlo_x, &f.qhi_x }, { &f.qlo_y, &f.qhi_y }, { &f.qlo_z, &f.qhi_z } };
                for (int k = 0; k < 3; k++) {
                    f.origin[k] = w.box.lo[k];
                    const double ext = (double)w.box.hi[k] - (double)w.box.lo[k];
                    int e = 1;   // smallest normal exponent
                    if (ext > 0) { int ex; std::frexp(ext / 255.0, &ex); e = ex + 127; /* 2^ex >= ext/255 */ if (e < 1) e = 1; if (e > 254) e = 254; }
                    f.e[k] = (uint8_t)e;
                    const double step = std::ldexp(1.0, e - 127);
                    for (int c = 0; c < w.n; c++) {
                        long lo = (long)std::floor(((double)w.cbox[c].lo[k] - (double)f.origin[k]) / step);
                        long hi = (long)std::ceil(((double)w.cbox[c].hi[k] - (double)f.origin[k]) / step);
                        // the device evaluates origin + step * q in fp32 (one rounding): keep the box conservative under that rounding too
                        while (lo > 0 && (float)((double)f.origin[k] + step * (double)lo) > w.cbox[c].lo[k]) lo--;
                        while (hi < 255 && (float)((double)f.origin[k] + step * (double)hi) < w.cbox[c].hi[k]) hi++;
                        lo = std::min(255L, std::max(0L, lo)); hi = std::min(255L, std::max(0L, hi));
                        *q[k][0] |= (uint32_t)lo << (8 * c); *q[k][1] |= (uint32_t)hi << (8 * c);
                    }
                }
                for (int c = 0; c < 4; c++) {
                    if (c < w.n) { f.mask |= (uint8_t)(1u << c); f.child[c] = w.child[c] >= 0 ? w.child[c] * 4 : w.child[c]; }
                    else {
                        // a missing child gets an INVERTED box (lo = 255, hi = 0 on every axis): whatever the ray, its entry plane lies behind its exit plane
                        // (|step / dir| is a normal float >= 2^-126, so 255 steps always differ from 0 steps), and the kernel needs no "child exists" test per slot
                        f.child[c] = 0x76543210;
                        for (int k = 0; k < 3; k++) { *q[k][0] |= 255u << (8 * c); *q[k][1] &= ~(255u << (8 * c)); }
                    }
                }
            }
            });
        }
    }
    pt.lap("node layout");
    // leaf entries.  Wide formats: in node order, the leaf children of a node one after the other in slot order (flat4_node's implied links).
    std::vector<uint32_t> entry_src;   // new entry -> position in R.leaf_prims
    if (out.format == kFlatF2) { entry_src.resize(R.leaf_prims.size()); for (size_t i = 0; i < entry_src.size(); i++) entry_src[i] = (uint32_t)i; }
    else if (out.format == kFlatQ8) {
        entry_src.reserve(R.leaf_prims.size());
        for (size_t i = 0; i < out.nodes_q8.size(); i++) {
            flat8_node& n = out.nodes_q8[i];
            n.leaf_base = (uint32_t)entry_src.size();
            for (int c = 0; c < 8; c++) {
                int32_t& k = out.child_links[i * 8 + c];
                if (k >= 0) continue;   // inner child, or kFlat8None (positive)
                const uint32_t e = (uint32_t)~k;
                // the BVH2 was built with one primitive per leaf; coincident boxes (duplicate triangles, split references with identical boxes) can still share a leaf once
                // the builder's depth limit is reached: such a scene takes the 4-wide format, whose leaves hold any number of entries (as a tree beyond 2^24 nodes does)
                if (!R.leaf_last[e]) return flatten_scene(d, out, max_triangles, kFlatQ4);
                k = ~(int32_t)entry_src.size(); entry_src.push_back(e);
            }
        }
    }
    else {
        entry_src.reserve(R.leaf_prims.size());
        auto relink = [&](int32_t* child, int n_children, uint32_t* counts, uint8_t* mask) {   // leaf children: entries appended in slot order, link = ~first new entry
            for (int c = 0; c < n_children; c++) {
                if (counts) counts[c] = 0;
                if (child[c] == 0x76543210 || child[c] >= 0) continue;
                const uint32_t first = (uint32_t)entry_src.size();
                for (uint32_t e = (uint32_t)~child[c];; e++) { entry_src.push_back(e); if (counts) counts[c]++; if (R.leaf_last[e]) break; }
                child[c] = ~(int32_t)first;
                if (mask) *mask |= (uint8_t)(16u << c);
            }
        };
        if (out.format == kFlatQ4) {
            for (auto& n : out.nodes) {
                // inner children are consecutive nodes already (memory order above); flat4_encode_links refuses a node whose links the layout does not imply
                uint32_t counts[4];
                relink(n.child, 4, counts, &n.mask);
                if (!flat4_encode_links(n.child, counts, 0u, n.links) || out.nodes.size() >= (1u << 24)) out.compact_links = false;   // slab flags: set below
            }
            // The explicit-link form of the tree (no implied links, no slabs) is what a scene beyond 2^24 nodes or 2^26 entries gets; CTL_FLAT_FORCE_EXPLICIT=1 (knobs build only) builds it
            // for any scene so that the tests can walk that path
            if (const char* e = knob_env("CTL_FLAT_FORCE_EXPLICIT")) if (atoi(e) != 0) out.compact_links = false;
        } else for (auto& n : out.nodes_f4) relink(n.child, 4, nullptr, nullptr);
    }
    out.leaves.resize(entry_src.size());
    parallel_for(entry_src.size(), [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            const uint32_t src = entry_src[i];
            const wtri& t = tris[R.leaf_prims[src]];
            flat_leaf& L = out.leaves[i];
            std::memset(&L, 0, sizeof(L));
            const ctl_woop_tri& w = d.woop[t.woop];   // the mesh's own rows: the kernel applies the instance transform to the ray
            std::memcpy(L.a, w.a, 16); std::memcpy(L.b, w.b, 16); std::memcpy(L.c, w.c, 16);
            std::memcpy(L.inv, d.node_inv_transforms[t.node].m, 48); L.w33 = d.node_inv_transforms[t.node].m[15];
            L.index = (t.tri << 1) | (R.leaf_last[src] ? 1u : 0u); L.node = t.node;
        }
    });
    pt.lap("leaf entries");
    out.max_depth = wdepth;
    if (out.format != kFlatQ8) out.child_links.clear();
    out.root_slab = false; out.slab_nodes = 0;
    if (out.format == kFlatQ4) {
        out.child_links.resize(out.nodes.size() * 4);
        for (size_t i = 0; i < out.nodes.size(); i++) std::memcpy(&out.child_links[i * 4], out.nodes[i].child, 16);
    }
    if ((out.format == kFlatQ4 && out.compact_links) || out.format == kFlatQ8) {
        // oriented slabs of the nodes that have leaf children (flat_slab.h), for both wide formats: W = children per node, explicit links in out.child_links (W per node),
        // a leaf = the entries from its first one up to the next `last` flag.  World-space vertices in double from the object-space ones.
        const int W = out.format == kFlatQ8 ? 8 : 4;
        const size_t n_nodes = out.format == kFlatQ8 ? out.nodes_q8.size() : out.nodes.size();
        const uint32_t none = 0x76543210u;
        auto world_tri = [&](uint32_t entry, double w[3][3], double& slack) {
            const size_t g = R.leaf_prims[entry_src[entry]]; const wtri& t = tris[g];
            const ltri& l = mesh_local[d.nodes[t.node].mesh_index][t.local];
            const float* M = d.node_transforms[t.node].m;
            for (int j = 0; j < 3; j++) for (int r = 0; r < 3; r++) w[j][r] = (double)M[r * 4] * l.v[j][0] + (double)M[r * 4 + 1] * l.v[j][1] + (double)M[r * 4 + 2] * l.v[j][2] + (double)M[r * 4 + 3];
            slack = woop_slack(l.v, M);
        };
        std::vector<uint8_t> has_slab(n_nodes, 0);
        // An INNER child whose whole subtree holds at most `sub_cap` triangles (a bottom node: the patch of surface under it is as flat as its triangles) gets a real interval too:
        // a ray that crosses the patch's box but not the patch itself is turned away one level higher and never fetches the bottom node.  sub_tris: triangles under every node
        // (children follow their parent in memory, so one backwards sweep does it).
        static const int sub_cap = [] { const char* e = knob_env("CTL_FLAT_SLAB_SUBTREE"); const int v = e ? atoi(e) : kSlabSubtreeDefault; return v < 0 ? 0 : (v > kSlabSubtreeMax ? kSlabSubtreeMax : v); }();
        auto node_of = [&](int32_t k) { return out.format == kFlatQ8 ? (size_t)k : (size_t)k / 4; };   // explicit inner link -> node index
        std::vector<uint8_t> sub_tris(n_nodes, 0);
        for (size_t i = n_nodes; i-- > 0;) {
            uint32_t n = 0;
            for (int c = 0; c < W; c++) {
                const int32_t k = out.child_links[i * W + c];
                if ((uint32_t)k == none) continue;
                if (k >= 0) n += sub_tris[node_of(k)];
                else for (uint32_t e = (uint32_t)~k;; e++) { n++; if (out.leaves[e].index & 1u) break; }
            }
            sub_tris[i] = (uint8_t)std::min(n, 255u);
        }
        static const double useful_below = [] { const char* e = knob_env("CTL_FLAT_SLAB_USEFUL"); return e ? atof(e) : 0.6; }();   // builder knob (part of the cache key)
        struct slab_codes { uint32_t slab_n; float base; uint8_t lo[8], hi[8]; };
        // the slab of one node: origin / exponents / child-box codes as stored, exist / leafm = per-slot masks, ch = its W explicit links.  false: the node carries none.
        auto make_slab = [&](const float* origin, const uint8_t* ex, uint32_t exist, uint32_t leafm, const uint8_t ql[3][8], const uint8_t qh[3][8], const int32_t* ch, slab_codes& S) -> bool {
            if (useful_below <= 0.0) return false;
            // triangles of the leaf children and of the inner children with small subtrees; `tight` = the children that get an interval of their own
            constexpr int kT = 8 * kSlabSubtreeMax; static thread_local std::vector<slab_ctri> T; if (T.size() < (size_t)kT) T.resize(kT);
            int nt = 0; uint32_t tight = 0;
            const int kTw = W * kSlabSubtreeMax;   // the 4-wide tree's own cap (its arrays held 4 x the subtree limit)
            for (int c = 0; c < W; c++) {
                if (!((exist >> c) & 1)) continue;
                if ((leafm >> c) & 1) { tight |= 1u << c; for (uint32_t e = (uint32_t)~ch[c];; e++) { T[nt].c = c; world_tri(e, T[nt].w, T[nt].slack); nt++; if ((out.leaves[e].index & 1u) || nt == kTw) break; } continue; }
                if (sub_cap <= 0 || sub_tris[node_of(ch[c])] > sub_cap) continue;
                const int nt_before = nt; bool complete = true;
                int32_t stack[128]; int sp = 0; stack[sp++] = ch[c];
                while (sp && complete) {
                    const int32_t k = stack[--sp];
                    if (k >= 0) { for (int q = 0; q < W; q++) { const int32_t kk = out.child_links[node_of(k) * W + q]; if ((uint32_t)kk == none) continue; if (sp < (W == 4 ? 64 : 128)) stack[sp++] = kk; else complete = false; } }
                    else for (uint32_t e = (uint32_t)~k;; e++) { if (nt < kTw) { T[nt].c = c; world_tri(e, T[nt].w, T[nt].slack); nt++; } else complete = false; if (out.leaves[e].index & 1u) break; }
                }
                if (complete) tight |= 1u << c; else nt = nt_before;   // an interval must cover EVERY triangle under the child, or the child keeps the whole node
            }
            if (!tight || nt == 0) return false;
            double stepk[3], ext1 = 0, mag = 0;
            for (int k = 0; k < 3; k++) { stepk[k] = std::ldexp(1.0, (int)ex[k] - 127); ext1 += 255.0 * stepk[k]; mag = std::max(mag, std::fabs((double)origin[k]) + 255.0 * stepk[k]); }
            int best_n[3] = { 0, 0, 0 }; double best_cost = 1e300; double best_lo[8], best_hi[8];
            // candidate directions: the triangles' own normals (at most 24 of them, evenly picked) and — for patches — the area-weighted mean normal of every child and of all
            double mean_n[9][3] = {};
            for (int t = 0; t < nt; t++) {
                const double (*w)[3] = T[t].w;
                const double a[3] = { w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2] }, b[3] = { w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2] };
                const double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
                for (int k = 0; k < 3; k++) { mean_n[T[t].c][k] += n[k]; mean_n[W][k] += n[k]; }
            }
            const int cand_step = std::max(1, nt / 24), n_tri_cand = (nt + cand_step - 1) / cand_step;
            for (int ci = 0; ci < n_tri_cand + (nt > 4 ? W + 1 : 0); ci++) {
                double n[3];
                if (ci < n_tri_cand) {
                    const double (*w)[3] = T[ci * cand_step].w;
                    const double a[3] = { w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2] }, b[3] = { w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2] };
                    n[0] = a[1] * b[2] - a[2] * b[1]; n[1] = a[2] * b[0] - a[0] * b[2]; n[2] = a[0] * b[1] - a[1] * b[0];
                } else for (int k = 0; k < 3; k++) n[k] = mean_n[ci - n_tri_cand][k];
                const double m = std::max(std::max(std::fabs(n[0]), std::fabs(n[1])), std::fabs(n[2]));
                if (!(m > 0) || !std::isfinite(m)) continue;
                int nq[3]; for (int k = 0; k < 3; k++) nq[k] = (int)std::lround(n[k] / m * (double)kSlabNMax);
                bool dup = false; if (nq[0] == best_n[0] && nq[1] == best_n[1] && nq[2] == best_n[2]) dup = true;
                if (dup) continue;
                double lo[8], hi[8]; for (int c = 0; c < 8; c++) { lo[c] = 1e300; hi[c] = -1e300; }
                for (int t = 0; t < nt; t++) for (int j = 0; j < 3; j++) {
                    const double D = nq[0] * (T[t].w[j][0] - (double)origin[0]) + nq[1] * (T[t].w[j][1] - (double)origin[1]) + nq[2] * (T[t].w[j][2] - (double)origin[2]);
                    lo[T[t].c] = std::min(lo[T[t].c], D); hi[T[t].c] = std::max(hi[T[t].c], D);
                }
                double cost = 0; int nl = 0;
                for (int c = 0; c < W; c++) if ((tight >> c) & 1) {
                    double range = 0; for (int k = 0; k < 3; k++) range += std::fabs((double)nq[k]) * stepk[k] * (double)((int)qh[k][c] - (int)ql[k][c]);
                    cost += range > 0 ? std::min(1.0, (hi[c] - lo[c]) / range) : 1.0; nl++;
                }
                cost /= nl;
                if (cost < best_cost) { best_cost = cost; for (int k = 0; k < 3; k++) best_n[k] = nq[k]; for (int c = 0; c < W; c++) { best_lo[c] = lo[c]; best_hi[c] = hi[c]; } }
            }
            if (!(best_cost < useful_below)) return false;
            // static pad per child: round-off reach of the object-space test (2^-20 of the magnitudes involved) + the node-extent share of the kernel's evaluation error
            const double n1 = std::fabs((double)best_n[0]) + std::fabs((double)best_n[1]) + std::fabs((double)best_n[2]);
            double pad[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int t = 0; t < nt; t++) pad[T[t].c] = std::max(pad[T[t].c], n1 * 9.5367431640625e-7 * (T[t].slack + mag) + (double)kSlabRayPad * ext1);   // 2^-20 of the magnitudes: ~8 x the fp32 round-off of the object-space test
            // D range the codes span: the leaf children's padded intervals — and, when the node has inner children too, the whole node (its
            // quantisation grid's box), which their code 0 .. 255 must cover
            double nlo = 1e300, nhi = -1e300;
            for (int c = 0; c < W; c++) if ((tight >> c) & 1) { nlo = std::min(nlo, best_lo[c] - pad[c]); nhi = std::max(nhi, best_hi[c] + pad[c]); }
            if (exist != tight) {   // a child without an interval of its own spans the whole node
                double pmax = 0; for (int c = 0; c < W; c++) pmax = std::max(pmax, pad[c]);
                double blo = 0, bhi = 0; for (int k = 0; k < 3; k++) { const double x = best_n[k] * 255.0 * stepk[k]; if (x < 0) blo += x; else bhi += x; }
                nlo = std::min(nlo, blo - pmax); nhi = std::max(nhi, bhi + pmax);
            }
            const float base = round_down(nlo);
            // step: a float with 5 mantissa bits (the top 14 bits of its pattern share a word with the normal), rounded up; 254 steps span the range
            float stepf = round_up((nhi - (double)base) / 254.0);
            if (!(stepf > 0.0f) || !std::isfinite(stepf)) stepf = 1.17549435e-38f;
            { uint32_t bits; std::memcpy(&bits, &stepf, 4); bits = (bits + 0x3ffffu) & 0xfffc0000u; std::memcpy(&stepf, &bits, 4); }
            if (!std::isfinite(stepf) || stepf < 1.17549435e-38f) return false;
            const double step = (double)stepf;
            bool ok = true;
            for (int c = 0; c < W; c++) {
                long lo, hi;
                if (!((exist >> c) & 1)) { lo = 255; hi = 0; }
                else if (!((tight >> c) & 1)) { lo = 0; hi = 255; }
                else {
                    lo = (long)std::floor((best_lo[c] - pad[c] - (double)base) / step);
                    hi = (long)std::ceil((best_hi[c] + pad[c] - (double)base) / step);
                    // conservative under the fp32 evaluation base + step * code as well
                    while (lo > 0 && (double)(float)((double)base + step * (double)lo) > best_lo[c] - pad[c]) lo--;
                    while (hi < 255 && (double)(float)((double)base + step * (double)hi) < best_hi[c] + pad[c]) hi++;
                    if (lo < 0 || hi > 255 || (double)base + step * (double)lo > best_lo[c] - pad[c] || (double)base + step * (double)hi < best_hi[c] + pad[c]) ok = false;
                }
                S.lo[c] = (uint8_t)(lo & 255); S.hi[c] = (uint8_t)(hi & 255);
            }
            if (!ok || (double)base + 255.0 * step < nhi) return false;
            uint32_t sb; std::memcpy(&sb, &stepf, 4);
            S.slab_n = ((uint32_t)best_n[0] & 63u) | (((uint32_t)best_n[1] & 63u) << 6) | (((uint32_t)best_n[2] & 63u) << 12) | sb;
            S.base = base;
            return true;
        };
        if (out.format == kFlatQ4) {
            parallel_for(out.nodes.size(), [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; i++) {
                    flat4_node& f = out.nodes[i];
                    f.slab_n = 0; f.slab_base = 0.0f; f.slab_lo = 0; f.slab_hi = 0xffffffffu;
                    uint8_t ql[3][8] = {}, qh[3][8] = {};
                    const uint32_t qlw[3] = { f.qlo_x, f.qlo_y, f.qlo_z }, qhw[3] = { f.qhi_x, f.qhi_y, f.qhi_z };
                    for (int k = 0; k < 3; k++) for (int c = 0; c < 4; c++) { ql[k][c] = (uint8_t)(qlw[k] >> (8 * c)); qh[k][c] = (uint8_t)(qhw[k] >> (8 * c)); }
                    slab_codes S;
                    if (!make_slab(f.origin, f.e, f.mask & 15u, (uint32_t)(f.mask >> 4) & (f.mask & 15u), ql, qh, &out.child_links[i * 4], S)) continue;
                    f.slab_n = S.slab_n; f.slab_base = S.base; f.slab_lo = 0; f.slab_hi = 0;
                    for (int c = 0; c < 4; c++) { f.slab_lo |= (uint32_t)S.lo[c] << (8 * c); f.slab_hi |= (uint32_t)S.hi[c] << (8 * c); }
                    has_slab[i] = 1;
                }
            });
            for (size_t i = 0; i < out.nodes.size(); i++) {
                out.slab_nodes += has_slab[i];
                // slab flag of an inner child = bit 0 of its link: bit 0 of links[0] for slot 0, bit 0 of the slot's nibble otherwise (flatten.h)
                for (int c = 0; c < 4; c++) {
                    const int32_t k = out.child_links[i * 4 + c];
                    if (k < 0 || k == 0x76543210 || !has_slab[(size_t)k / 4]) continue;
                    if (c == 0) out.nodes[i].links[0] |= 1u; else if (c == 1) out.nodes[i].links[0] |= 1u << 26; else if (c == 2) out.nodes[i].links[1] |= 1u << 2; else out.nodes[i].links[0] |= 1u << 30;
                }
            }
            out.root_slab = has_slab[0] != 0;
        } else {
            parallel_for(out.nodes_q8.size(), [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; i++) {
                    flat8_node& f = out.nodes_q8[i];
                    uint8_t ql[3][8], qh[3][8];
                    const uint32_t* qlw[3] = { f.qlo_x, f.qlo_y, f.qlo_z }; const uint32_t* qhw[3] = { f.qhi_x, f.qhi_y, f.qhi_z };
                    for (int k = 0; k < 3; k++) for (int c = 0; c < 8; c++) { ql[k][c] = (uint8_t)(qlw[k][c >> 2] >> (8 * (c & 3))); qh[k][c] = (uint8_t)(qhw[k][c >> 2] >> (8 * (c & 3))); }
                    const uint32_t leafm = (f.base_b >> 24) & ~(uint32_t)f.imask, exist = leafm | f.imask;
                    slab_codes S;
                    if (!make_slab(f.origin, f.e, exist, leafm, ql, qh, &out.child_links[i * 8], S)) continue;
                    f.slab_n = S.slab_n; f.slab_base = S.base; f.slab_lo[0] = f.slab_lo[1] = f.slab_hi[0] = f.slab_hi[1] = 0;
                    for (int c = 0; c < 8; c++) { f.slab_lo[c >> 2] |= (uint32_t)S.lo[c] << (8 * (c & 3)); f.slab_hi[c >> 2] |= (uint32_t)S.hi[c] << (8 * (c & 3)); }
                    has_slab[i] = 1;
                }
            });
            // B of an inner slot: the child node has leaf slots or a slab — a step on it loads q5 too, and a lane that holds a parked leaf group waits before it (flat8.h)
            auto heavy = [&](size_t k) { const flat8_node& c = out.nodes_q8[k]; return has_slab[k] || ((c.base_b >> 24) & ~(uint32_t)c.imask) != 0u; };
            for (size_t i = 0; i < out.nodes_q8.size(); i++) {
                out.slab_nodes += has_slab[i];
                for (int c = 0; c < 8; c++) { const int32_t k = out.child_links[i * 8 + c]; if (k >= 0 && (uint32_t)k != none && heavy((size_t)k)) out.nodes_q8[i].base_b |= 1u << (24 + c); }
            }
            out.root_slab = heavy(0);
        }
        pt.lap("slabs");
    }
    if (out.format == kFlatQ4) {
        // Slots without a child.  Their box is inverted, which the slab test rejects — unless the ray origin is so far from a small node (~2^16 node extents) that
        // entry and exit plane round to the same distance on every axis; then the kernel follows the slot's link.  It must lead somewhere harmless: the link of a
        // slot that exists (a second visit of a sibling finds nothing new).  Implied links: an empty slot's nibble is 0, which is the node's first inner child; a node
        // without inner children gets the empty slot's LEAF bit set and nibble 15 = its first leaf entry (only the kernels and flat4_implied_links read a leaf bit
        // without its exists bit; host code asks for both).  Explicit links: the slot's word repeats the first child's.
        for (size_t i = 0; i < out.nodes.size(); i++) {
            flat4_node& f = out.nodes[i];
            const uint32_t exist = f.mask & 15u, leafm = (uint32_t)(f.mask >> 4) & exist;
            if (exist == 15u || exist == 0u) continue;
            if (out.compact_links) {
                if ((exist & ~leafm) != 0u) continue;   // nibble 0 -> first inner child
                for (int c = 1; c < 4; c++) if (!((exist >> c) & 1u)) {
                    f.mask |= (uint8_t)(16u << c);
                    if (c == 1) f.links[0] |= 15u << 26; else if (c == 2) f.links[1] |= 15u << 2; else { f.links[0] |= 3u << 30; f.links[1] |= 3u; }
                }
            } else for (int c = 1; c < 4; c++) if (!((exist >> c) & 1u)) f.child[c] = f.child[0];
        }
    }
    if (!key.empty()) {
        cache_writer wr("flat", key);
        if (wr.active()) { wr.value(out.format); wr.value(out.max_depth); { const int cl = out.compact_links ? 1 : 0; wr.value(cl); } wr.vector(out.nodes); wr.vector(out.nodes_q8); wr.vector(out.nodes_f4); wr.vector(out.nodes_f2); wr.vector(out.leaves); wr.vector(out.child_links); { const int rs = out.root_slab ? 1 : 0; wr.value(rs); const uint64_t sn = out.slab_nodes; wr.value(sn); const uint64_t sr = out.split_refs; wr.value(sr); } wr.commit(); pt.lap("cache write"); }
    }
    return true;
}

}  // namespace ctl
