// sbvh_builder.cpp — split BVH with spatial splits, restating the build the reference runs for every mesh
// (ConstructBVH, Engine/MeshLoader/BVHBuilderHelper.cpp:116-127 -> SplitBVHBuilder, Engine/SpatialStructures/BVH/SplitBVHBuilder.cpp:219-640;
// the algorithm is Stich et al. 2009 as implemented by Aila / Karras): full-sweep object splits on the three axes, spatial split
// candidates from 128 bins per axis with chopped references, reference unsplitting, SAH costs 1 / 1, leaves of at most 8 triangles,
// spatial splits only where the object split's children overlap by more than 1e-5 of the root area and above depth 48.
//
// Every comparison that decides the tree is made on the same float expressions in the same order as there (box area as
// 2 (x y + x z + y z), SAH sums left to right, ties by the same rules), and the references live on one stack that is consumed from
// the top, so the emitted arrays are the reference's arrays element for element — tests/test_sbvh.py holds its outputs as golden
// fixtures.  The sort only needs the reference's ORDER (centroid, then triangle id: a strict total order), not its quicksort.
#include "bvh_builder.h"
#include <algorithm>
#include <cfloat>
#include <cstring>

namespace ctl {
namespace {

constexpr int kMaxDepth = 64, kMaxSpatialDepth = 48, kSpatialBins = 128;
constexpr float kSplitAlpha = 1.0e-5f;
constexpr int kNoChild = 0x76543210;

struct box3 {
    float lo[3], hi[3];
    static box3 empty() { box3 b; for (int k = 0; k < 3; k++) { b.lo[k] = FLT_MAX; b.hi[k] = -FLT_MAX; } return b; }
    float area() const { const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2]; return 2.0f * (x * y + x * z + y * z); }   // AABB::Area (Math/AABB.h:19-23)
    box3 join(const box3& o) const { box3 r; for (int k = 0; k < 3; k++) { r.lo[k] = o.lo[k] < lo[k] ? o.lo[k] : lo[k]; r.hi[k] = o.hi[k] > hi[k] ? o.hi[k] : hi[k]; } return r; }
    box3 join(const float* p) const { box3 r; for (int k = 0; k < 3; k++) { r.lo[k] = p[k] < lo[k] ? p[k] : lo[k]; r.hi[k] = p[k] > hi[k] ? p[k] : hi[k]; } return r; }
    box3 meet(const box3& o) const { box3 r; for (int k = 0; k < 3; k++) { r.lo[k] = o.lo[k] > lo[k] ? o.lo[k] : lo[k]; r.hi[k] = o.hi[k] < hi[k] ? o.hi[k] : hi[k]; } return r; }
};
inline float min3(float a, float b, float c) { const float d = a < b ? a : b; return d < c ? d : c; }
inline float max3(float a, float b, float c) { const float d = a > b ? a : b; return d > c ? d : c; }

struct ref_t { int tri; box3 box; };
struct spec_t { int n = 0; box3 box = box3::empty(); };
struct tnode { box3 box; uint32_t a, b; bool leaf; };   // leaf: [a, b) into `order`; inner: children a (left), b (right)

struct sbvh {
    const float* P; const uint32_t* I; int max_leaf;
    std::vector<ref_t> refs; std::vector<tnode> nodes; std::vector<int> order; std::vector<box3> right_boxes;
    float min_overlap = 0; int max_depth = 0;

    const float* vert(int tri, int k) const { return P + 3 * (size_t)(I ? I[3 * (size_t)tri + k] : 3 * (uint32_t)tri + k); }

    uint32_t leaf(const spec_t& s) {   // createLeaf (:357-363): the references come off the top of the stack
        for (int i = 0; i < s.n; i++) { order.push_back(refs.back().tri); refs.pop_back(); }
        nodes.push_back(tnode{ s.box, (uint32_t)order.size() - (uint32_t)s.n, (uint32_t)order.size(), true });
        return (uint32_t)nodes.size() - 1;
    }
    void sort_refs(int first, int dim) {   // sortCompare (:257-266)
        std::sort(refs.begin() + first, refs.end(), [dim](const ref_t& x, const ref_t& y) {
            const float cx = x.box.lo[dim] + x.box.hi[dim], cy = y.box.lo[dim] + y.box.hi[dim];
            return cx < cy || (cx == cy && x.tri < y.tri); });
    }
    // clb::SplitNode (BVHBuilderHelper.cpp:78-112): clip the triangle at the plane, keep both parts inside the reference's box
    void split_ref(ref_t& l, ref_t& r, const ref_t& src, int dim, float pos) const {
        box3 lb = box3::empty(), rb = box3::empty();
        const float* v1 = vert(src.tri, 2);
        for (int i = 0; i < 3; i++) {
            const float* v0 = v1; v1 = vert(src.tri, i);
            const float a = v0[dim], b = v1[dim];
            if (a <= pos) lb = lb.join(v0);
            if (a >= pos) rb = rb.join(v0);
            if ((a < pos && b > pos) || (a > pos && b < pos)) {
                float t = (pos - a) / (b - a); t = t < 0.0f ? 0.0f : t; t = t > 1.0f ? 1.0f : t;   // clamp01 = min(max(t, 0), 1)
                const float x[3] = { v0[0] * (1.0f - t) + v1[0] * t, v0[1] * (1.0f - t) + v1[1] * t, v0[2] * (1.0f - t) + v1[2] * t };   // math::lerp<Vec3f, float>
                lb = lb.join(x); rb = rb.join(x);
            }
        }
        lb.hi[dim] = pos; rb.lo[dim] = pos;
        l.box = lb.meet(src.box); r.box = rb.meet(src.box);
        l.tri = r.tri = src.tri;
    }

    struct obj_split { float sah = FLT_MAX; int dim = 0, n_left = 0; box3 lbox = box3::empty(), rbox = box3::empty(); };
    obj_split find_object_split(const spec_t& s, float node_sah) {   // :367-411
        obj_split best; float best_tie = FLT_MAX;
        const int first = (int)refs.size() - s.n;
        for (int dim = 0; dim < 3; dim++) {
            sort_refs(first, dim);
            const ref_t* r = &refs[first];
            box3 rb = box3::empty();
            for (int i = s.n - 1; i > 0; i--) { rb = rb.join(r[i].box); right_boxes[i - 1] = rb; }
            box3 lb = box3::empty();
            for (int i = 1; i < s.n; i++) {
                lb = lb.join(r[i - 1].box);
                const float la = lb.area(), ra = right_boxes[i - 1].area();
                const float sah = node_sah + la * (float)i + ra * (float)(s.n - i);
                const float tie = (float)i * (float)i + (float)(s.n - i) * (float)(s.n - i);
                if (sah < best.sah || (sah == best.sah && tie < best_tie)) { best.sah = sah; best.dim = dim; best.n_left = i; best.lbox = lb; best.rbox = right_boxes[i - 1]; best_tie = tie; }
            }
        }
        return best;
    }
    struct spa_split { float sah = FLT_MAX; int dim = 0; float pos = 0.0f; };
    struct bin_t { box3 box; int enter, exit; };
    bin_t bins[3][kSpatialBins];
    spa_split find_spatial_split(const spec_t& s, float node_sah) {   // :427-513
        float origin[3], bin_size[3], inv_bin[3];
        for (int k = 0; k < 3; k++) { origin[k] = s.box.lo[k]; bin_size[k] = (s.box.hi[k] - origin[k]) * (1.0f / (float)kSpatialBins); inv_bin[k] = 1.0f / bin_size[k]; }
        for (int d = 0; d < 3; d++) for (int i = 0; i < kSpatialBins; i++) { bins[d][i].box = box3::empty(); bins[d][i].enter = bins[d][i].exit = 0; }
        for (size_t ri = refs.size() - (size_t)s.n; ri < refs.size(); ri++) {
            const ref_t& rf = refs[ri];
            int fb[3], lb[3];
            for (int k = 0; k < 3; k++) {
                fb[k] = std::min(std::max((int)((rf.box.lo[k] - origin[k]) * inv_bin[k]), 0), kSpatialBins - 1);
                lb[k] = std::min(std::max((int)((rf.box.hi[k] - origin[k]) * inv_bin[k]), fb[k]), kSpatialBins - 1);
            }
            for (int d = 0; d < 3; d++) {
                ref_t cur = rf;
                for (int i = fb[d]; i < lb[d]; i++) {
                    ref_t l, r; split_ref(l, r, cur, d, origin[d] + bin_size[d] * (float)(i + 1));
                    bins[d][i].box = bins[d][i].box.join(l.box);
                    cur = r;
                }
                bins[d][lb[d]].box = bins[d][lb[d]].box.join(cur.box);
                bins[d][fb[d]].enter++; bins[d][lb[d]].exit++;
            }
        }
        spa_split best;
        for (int d = 0; d < 3; d++) {
            box3 rb = box3::empty();
            for (int i = kSpatialBins - 1; i > 0; i--) { rb = rb.join(bins[d][i].box); right_boxes[i - 1] = rb; }
            box3 lb = box3::empty(); int ln = 0, rn = s.n;
            for (int i = 1; i < kSpatialBins; i++) {
                lb = lb.join(bins[d][i - 1].box); ln += bins[d][i - 1].enter; rn -= bins[d][i - 1].exit;
                const float sah = node_sah + lb.area() * (float)ln + right_boxes[i - 1].area() * (float)rn;
                if (sah < best.sah) { best.sah = sah; best.dim = d; best.pos = origin[d] + bin_size[d] * (float)i; }
            }
        }
        return best;
    }
    void do_spatial_split(spec_t& L, spec_t& R, const spec_t& s, const spa_split& sp) {   // :517-607
        const int left_start = (int)refs.size() - s.n; int left_end = left_start, right_start = (int)refs.size();
        L.box = R.box = box3::empty();
        for (int i = left_end; i < right_start; i++) {
            if (refs[i].box.hi[sp.dim] <= sp.pos) { L.box = L.box.join(refs[i].box); std::swap(refs[i], refs[left_end++]); }
            else if (refs[i].box.lo[sp.dim] >= sp.pos) { R.box = R.box.join(refs[i].box); std::swap(refs[i--], refs[--right_start]); }
        }
        while (left_end < right_start) {   // straddlers: keep whole on one side or duplicate, whichever is cheapest
            ref_t lr, rr; split_ref(lr, rr, refs[left_end], sp.dim, sp.pos);
            const box3 lub = L.box.join(refs[left_end].box), rub = R.box.join(refs[left_end].box), ldb = L.box.join(lr.box), rdb = R.box.join(rr.box);
            const float lac = (float)(left_end - left_start), rac = (float)((int)refs.size() - right_start), lbc = (float)(left_end - left_start + 1), rbc = (float)((int)refs.size() - right_start + 1);
            const float unsplit_left = lub.area() * lbc + R.box.area() * rac, unsplit_right = L.box.area() * lac + rub.area() * rbc, duplicate = ldb.area() * lbc + rdb.area() * rbc;
            const float m = min3(unsplit_left, unsplit_right, duplicate);
            if (m == unsplit_left) { L.box = lub; left_end++; }
            else if (m == unsplit_right) { R.box = rub; std::swap(refs[left_end], refs[--right_start]); }
            else { L.box = ldb; R.box = rdb; refs[left_end++] = lr; refs.push_back(rr); }
        }
        L.n = left_end - left_start; R.n = (int)refs.size() - right_start;
    }

    uint32_t build(spec_t s, int level) {   // buildNode (:278-353)
        if (level > max_depth) max_depth = level;
        {   // degenerate references go
            const int first = (int)refs.size() - s.n;
            for (int i = (int)refs.size() - 1; i >= first; i--) {
                const float sx = refs[i].box.hi[0] - refs[i].box.lo[0], sy = refs[i].box.hi[1] - refs[i].box.lo[1], sz = refs[i].box.hi[2] - refs[i].box.lo[2];
                if (min3(sx, sy, sz) < 0.0f || sx + sy + sz == max3(sx, sy, sz)) { refs[i] = refs.back(); refs.pop_back(); }
            }
            s.n = (int)refs.size() - first;
        }
        if (s.n <= 1 || level >= kMaxDepth) return leaf(s);
        const float area = s.box.area(), leaf_sah = area * (float)s.n, node_sah = area * 2.0f;
        const obj_split os = find_object_split(s, node_sah);
        spa_split ss;
        if (level < kMaxSpatialDepth) {
            const box3 overlap = os.lbox.meet(os.rbox);
            if (overlap.area() >= min_overlap) ss = find_spatial_split(s, node_sah);
        }
        const float best = min3(leaf_sah, os.sah, ss.sah);
        if (best == leaf_sah && s.n <= max_leaf) return leaf(s);
        spec_t L, R;
        if (best == ss.sah) do_spatial_split(L, R, s, ss);
        if (!L.n || !R.n) {   // performObjectSplit (:415-423)
            sort_refs((int)refs.size() - s.n, os.dim);
            L.n = os.n_left; L.box = os.lbox; R.n = s.n - os.n_left; R.box = os.rbox;
        }
        const uint32_t rn = build(R, level + 1);
        const uint32_t ln = build(L, level + 1);
        nodes.push_back(tnode{ s.box, ln, rn, false });
        return (uint32_t)nodes.size() - 1;
    }
};

void put_left(ctl_bvh_node& n, const box3& b) { n.a[0] = b.lo[0]; n.a[1] = b.hi[0]; n.a[2] = b.lo[1]; n.a[3] = b.hi[1]; n.c[0] = b.lo[2]; n.c[1] = b.hi[2]; }
void put_right(ctl_bvh_node& n, const box3& b) { n.b[0] = b.lo[0]; n.b[1] = b.hi[0]; n.b[2] = b.lo[1]; n.b[3] = b.hi[1]; n.c[2] = b.lo[2]; n.c[3] = b.hi[2]; }

// handleNode (SplitBVHBuilder.cpp:163-203): inner nodes in pre-order, leaves in the order they are reached
struct writer {
    const sbvh& B; bvh_result& out;
    int leaf_code(const tnode& t) {
        const uint32_t first = (uint32_t)out.leaf_prims.size();
        for (uint32_t j = t.a; j < t.b; j++) { out.leaf_prims.push_back((uint32_t)B.order[j]); out.leaf_last.push_back(j + 1 == t.b ? 1 : 0); }
        return ~(int)first;
    }
    int emit(uint32_t ti, uint32_t parent) {
        const tnode& t = B.nodes[ti];
        if (t.leaf) return t.a == t.b ? kNoChild : leaf_code(t);
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        const int a = emit(t.a, me * 4), b = emit(t.b, me * 4);
        ctl_bvh_node& n = out.nodes[me];
        std::memset(&n, 0, sizeof(n));
        n.child0 = a; n.child1 = b; n.parent = parent;
        put_left(n, B.nodes[t.a].box); put_right(n, B.nodes[t.b].box);
        return (int)(me * 4);
    }
};

} // namespace

void build_sbvh(const float* positions, const uint32_t* indices, uint32_t n_tri, int max_leaf, bvh_result& out) {
    out.nodes.clear(); out.leaf_prims.clear(); out.leaf_last.clear(); out.root = 0; out.max_depth = 0;
    sbvh B; B.P = positions; B.I = indices; B.max_leaf = max_leaf;
    spec_t root;
    B.refs.reserve((size_t)n_tri * 2);
    for (uint32_t j = 0; j < n_tri; j++) {   // clb::iterateObjects (BVHBuilderHelper.cpp:41-50)
        ref_t r; r.tri = (int)j; r.box = box3::empty();
        for (int k = 0; k < 3; k++) r.box = r.box.join(B.vert((int)j, k));
        root.box = root.box.join(r.box); root.n++;
        B.refs.push_back(r);
    }
    B.min_overlap = root.box.area() * kSplitAlpha;
    B.right_boxes.resize(std::max<size_t>(n_tri, kSpatialBins));
    const uint32_t r = n_tri ? B.build(root, 0) : 0;
    out.max_depth = B.max_depth;
    writer W{ B, out };
    if (!n_tri) { out.root = kNoChild; return; }
    const tnode& t = B.nodes[r];
    if (t.leaf) {   // one-leaf mesh: root node = (leaf, none), right box = [0, 0] (:176-189)
        ctl_bvh_node n; std::memset(&n, 0, sizeof(n));
        n.child0 = W.leaf_code(t); n.child1 = kNoChild; n.parent = 0xffffffffu;
        put_left(n, t.box);
        box3 z; for (int k = 0; k < 3; k++) z.lo[k] = z.hi[k] = 0.0f;
        put_right(n, z);
        out.nodes.push_back(n); out.root = 0;
    } else out.root = W.emit(r, 0xffffffffu);
}

} // namespace ctl
