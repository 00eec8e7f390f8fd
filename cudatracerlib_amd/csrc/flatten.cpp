// flatten.cpp — optional re-layout at scene upload: bake every node's transform into its triangles and build ONE
// world-space BVH over all instanced triangles ("flattening static instances into one BVH is allowed by the
// API and likely necessary").  The MI355X has 288 GB of HBM: 64 B per instanced triangle buys a traversal without the
// per-instance ray transform, without restarting at a mesh root per instance, and with two instead of four kinds of
// work per wave.  Results: same triangle / node / material as the two-level traversal; t,u,v agree to fp32 round-off
// (the Woop rows are recomputed in double precision from the world-space vertices) instead of bit-for-bit.
#include "flatten.h"
#include "bvh_builder.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <stdexcept>

namespace ctl {

namespace {

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
// Woop rows of a triangle (TriIntersectorData::setData, Engine/TriIntersectorData.cu:5-18) computed in double, rounded once
bool woop_rows(const double v[3][3], float a[4], float b[4], float c[4]) {
    double e0[3], e1[3], n[3];
    for (int k = 0; k < 3; k++) { e0[k] = v[0][k] - v[2][k]; e1[k] = v[1][k] - v[2][k]; }
    n[0] = e0[1] * e1[2] - e0[2] * e1[1]; n[1] = e0[2] * e1[0] - e0[0] * e1[2]; n[2] = e0[0] * e1[1] - e0[1] * e1[0];
    const double m[16] = { e0[0], e1[0], n[0], v[2][0], e0[1], e1[1], n[1], v[2][1], e0[2], e1[2], n[2], v[2][2], 0, 0, 0, 1 };
    double inv[16];
    if (!inv4(m, inv)) return false;
    a[0] = (float)inv[8]; a[1] = (float)inv[9]; a[2] = (float)inv[10]; a[3] = (float)-inv[11];
    for (int j = 0; j < 4; j++) { b[j] = (float)inv[j]; c[j] = (float)inv[4 + j]; }
    return true;
}
float round_down(double x) { float f = (float)x; return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f; }
float round_up(double x) { float f = (float)x; return ((double)f < x) ? std::nextafterf(f, INFINITY) : f; }

}  // namespace

bool flatten_scene(const ctl_scene_desc& d, flat_scene& out, size_t max_triangles, int width) {
    out.nodes.clear(); out.nodes8.clear(); out.leaves.clear(); out.width = width == 8 ? 8 : 4;
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
    struct wtri { double v[3][3]; uint32_t tri, node; };
    std::vector<wtri> tris; tris.reserve(total);
    std::vector<aabb> boxes; boxes.reserve(total);
    for (uint32_t k = 0; k < d.n_nodes; k++) {
        const ctl_node& N = d.nodes[k]; const ctl_kernel_mesh& km = d.meshes[N.mesh_index];
        const float* M = d.node_transforms[k].m;
        for (auto& e : mesh_tris[N.mesh_index]) {
            double lv[3][3];
            if (!woop_vertices(d.woop[e.second], lv)) continue;   // degenerate triangle: can never be hit
            wtri t; t.tri = km.tri_offset + e.first; t.node = k;
            aabb b; b.reset();
            for (int j = 0; j < 3; j++) {
                for (int r = 0; r < 3; r++) t.v[j][r] = (double)M[r * 4] * lv[j][0] + (double)M[r * 4 + 1] * lv[j][1] + (double)M[r * 4 + 2] * lv[j][2] + (double)M[r * 4 + 3];
                for (int r = 0; r < 3; r++) { const float lo = round_down(t.v[j][r]), hi = round_up(t.v[j][r]); if (lo < b.lo[r]) b.lo[r] = lo; if (hi > b.hi[r]) b.hi[r] = hi; }
            }
            tris.push_back(t); boxes.push_back(b);
        }
    }
    bvh_result R;
    build_bvh(boxes, 4, true, 60, R);
    // quantise a child box conservatively against the node's own box (one exponent per axis)
    auto quantise = [&](const aabb& nbox, const aabb* cbox, int n, float origin[3], uint8_t e_out[3], uint8_t qlo[3][8], uint8_t qhi[3][8]) {
        for (int k = 0; k < 3; k++) {
            origin[k] = nbox.lo[k];
            const double ext = (double)nbox.hi[k] - (double)nbox.lo[k];
            int e = 1;   // smallest normal exponent
            if (ext > 0) { int ex; std::frexp(ext / 255.0, &ex); e = ex + 127; /* 2^ex >= ext/255 */ if (e < 1) e = 1; if (e > 254) e = 254; }
            e_out[k] = (uint8_t)e;
            const double step = std::ldexp(1.0, e - 127);
            for (int c = 0; c < n; c++) {
                long lo = (long)std::floor(((double)cbox[c].lo[k] - (double)origin[k]) / step);
                long hi = (long)std::ceil(((double)cbox[c].hi[k] - (double)origin[k]) / step);
                // the device evaluates origin + step * q in fp32 (one rounding): keep the box conservative under that rounding too
                while (lo > 0 && (float)((double)origin[k] + step * (double)lo) > cbox[c].lo[k]) lo--;
                while (hi < 255 && (float)((double)origin[k] + step * (double)hi) < cbox[c].hi[k]) hi++;
                qlo[k][c] = (uint8_t)std::min(255L, std::max(0L, lo)); qhi[k][c] = (uint8_t)std::min(255L, std::max(0L, hi));
            }
        }
    };
    int wdepth = 0;
    if (out.width == 8) {
        std::vector<wide8_node> W;
        collapse_bvh8(R, W, wdepth);
        out.nodes8.resize(W.size());
        for (size_t i = 0; i < W.size(); i++) {
            const wide8_node& w = W[i]; flat8_node& f = out.nodes8[i];
            std::memset(&f, 0, sizeof(f));
            uint8_t qlo[3][8] = {}, qhi[3][8] = {};
            quantise(w.box, w.cbox, w.n, f.origin, f.e, qlo, qhi);
            uint32_t* ql[3] = { f.qlo_x, f.qlo_y, f.qlo_z }; uint32_t* qh[3] = { f.qhi_x, f.qhi_y, f.qhi_z };
            for (int k = 0; k < 3; k++) for (int c = 0; c < w.n; c++) { ql[k][c / 4] |= (uint32_t)qlo[k][c] << (8 * (c % 4)); qh[k][c / 4] |= (uint32_t)qhi[k][c] << (8 * (c % 4)); }
            for (int c = 0; c < 8; c++) {
                if (c < w.n) { f.mask |= (uint8_t)(1u << c); f.child[c] = w.child[c] >= 0 ? w.child[c] * 8 : w.child[c]; }
                else f.child[c] = 0x76543210;
            }
        }
    } else {
        // collapse to 4-wide nodes and quantise the child boxes conservatively
        std::vector<wide4_node> W;
        collapse_bvh4(R, W, wdepth);
        {   // memory order: the inner children of a node sit next to each other (<= 256 B = two 128-B L2 lines), subtrees stay
            // clustered.  Traversal is bound by the rate of random line fetches (tools/gather_probe.hip), and a ray that enters a node
            // usually enters one or two of its children next: siblings sharing a line turn some of those fetches into L2 hits.
            std::vector<int> new_id(W.size(), -1), order; order.reserve(W.size());
            std::vector<int> stack; new_id[0] = 0; order.push_back(0); stack.push_back(0);
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
        out.nodes.resize(W.size());
        for (size_t i = 0; i < W.size(); i++) {
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
                else f.child[c] = 0x76543210;
            }
        }
    }
    out.leaves.resize(R.leaf_prims.size());
    for (size_t i = 0; i < R.leaf_prims.size(); i++) {
        const wtri& t = tris[R.leaf_prims[i]];
        flat_leaf& L = out.leaves[i];
        std::memset(&L, 0, sizeof(L));
        if (!woop_rows(t.v, L.a, L.b, L.c)) { L.a[3] = 0; }   // all-zero rows: t = 0/0 = NaN, never accepted
        L.index = (t.tri << 1) | (R.leaf_last[i] ? 1u : 0u); L.node = t.node;
    }
    out.max_depth = wdepth;
    return true;
}

}  // namespace ctl
