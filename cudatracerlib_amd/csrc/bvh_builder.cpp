// bvh_builder.cpp — binned-SAH BVH2 build (see bvh_builder.h).
#include "bvh_builder.h"
#include <algorithm>
#include <atomic>
#include <future>
#include <thread>
#include <cmath>
#include <cstring>

namespace ctl {
namespace {

constexpr int kBins = 32;
constexpr int kEmptyChild = 0x76543210;

struct tmp_node { aabb box; int left, right; uint32_t begin, end; };   // leaf: left == -1

struct builder {
    const std::vector<aabb>& boxes;
    std::vector<float> cx, cy, cz;
    std::vector<uint32_t> idx, scratch;
    std::vector<tmp_node> pool;
    std::atomic<int> pool_used{ 0 };
    int max_leaf, depth_limit;
    float node_cost = 1.0f;   // SAH cost of visiting a node, in triangle tests
    std::atomic<int> max_depth{ 0 };

    explicit builder(const std::vector<aabb>& b, int ml, int dl) : boxes(b), max_leaf(ml), depth_limit(dl) {
        size_t n = b.size();
        cx.resize(n); cy.resize(n); cz.resize(n); idx.resize(n); scratch.resize(n >= (1u << 18) ? n : 0);
        for (size_t i = 0; i < n; i++) {
            cx[i] = 0.5f * (b[i].lo[0] + b[i].hi[0]); cy[i] = 0.5f * (b[i].lo[1] + b[i].hi[1]); cz[i] = 0.5f * (b[i].lo[2] + b[i].hi[2]);
            idx[i] = (uint32_t)i;
        }
        pool.resize(std::max<size_t>(2 * n, 2));
    }
    const float* cen(int axis) const { return axis == 0 ? cx.data() : (axis == 1 ? cy.data() : cz.data()); }

    // run f(chunk_begin, chunk_end, chunk) over [begin, end) on `threads` host threads
    template <typename F> static void chunks(uint32_t begin, uint32_t end, int threads, const F& f) {
        if (threads <= 1) { f(begin, end, 0); return; }
        std::vector<std::thread> th; const uint64_t n = end - begin;
        for (int t = 0; t < threads; t++) th.emplace_back([&, t]() { f(begin + (uint32_t)(n * t / threads), begin + (uint32_t)(n * (t + 1) / threads), t); });
        for (auto& x : th) x.join();
    }
    struct bins3 { aabb bb[3][kBins]; uint32_t cnt[3][kBins]; void reset() { for (int a = 0; a < 3; a++) for (int b = 0; b < kBins; b++) { bb[a][b].reset(); cnt[a][b] = 0; } } };

    // a subtree with par_budget b may use 2^b host threads: large nodes split their passes over them, then the two children share them
    int build(uint32_t begin, uint32_t end, int depth, int par_budget) {
        int me = pool_used.fetch_add(1);
        tmp_node& nd = pool[me];
        nd.begin = begin; nd.end = end; nd.left = nd.right = -1;
        const uint32_t n = end - begin;
        const int threads = (int)std::min<uint32_t>(1u << std::max(par_budget, 0), std::max<uint32_t>(1, n >> 17));
        aabb cb;
        {   // node box and centroid box
            std::vector<aabb> nb(threads), cbs(threads);
            chunks(begin, end, threads, [&](uint32_t b0, uint32_t b1, int t) {
                aabb x, c; x.reset(); c.reset();
                for (uint32_t i = b0; i < b1; i++) { const uint32_t p = idx[i]; x.grow(boxes[p]); const float cc[3] = { cx[p], cy[p], cz[p] }; c.grow(cc); }
                nb[t] = x; cbs[t] = c;
            });
            nd.box = nb[0]; cb = cbs[0];
            for (int t = 1; t < threads; t++) { nd.box.grow(nb[t]); cb.grow(cbs[t]); }
        }
        int d = max_depth.load();
        while (depth > d && !max_depth.compare_exchange_weak(d, depth)) {}
        if (n <= 1) return me;
        // how many levels are left vs. how many a balanced tree still needs
        int remaining = depth_limit - depth;
        uint32_t leaves_needed = (n + max_leaf - 1) / max_leaf;
        bool force_median = remaining <= 2 || leaves_needed > (1u << std::min(remaining - 2, 30));

        int best_axis = -1, best_bin = -1; float best_cost = 3.402823466e+38f;
        if (!force_median) {
            float lo[3], scale[3]; bool use[3];
            for (int a = 0; a < 3; a++) { lo[a] = cb.lo[a]; use[a] = cb.hi[a] > cb.lo[a]; scale[a] = use[a] ? kBins / (cb.hi[a] - cb.lo[a]) : 0.0f; }
            // one pass bins the primitives along all three axes
            std::vector<bins3> tb(threads);
            chunks(begin, end, threads, [&](uint32_t b0, uint32_t b1, int t) {
                bins3& B = tb[t]; B.reset();
                for (uint32_t i = b0; i < b1; i++) {
                    const uint32_t p = idx[i]; const aabb& bx = boxes[p]; const float c[3] = { cx[p], cy[p], cz[p] };
                    for (int a = 0; a < 3; a++) {
                        if (!use[a]) continue;
                        const int b = std::min(kBins - 1, std::max(0, (int)((c[a] - lo[a]) * scale[a])));
                        B.bb[a][b].grow(bx); B.cnt[a][b]++;
                    }
                }
            });
            for (int t = 1; t < threads; t++) for (int a = 0; a < 3; a++) for (int b = 0; b < kBins; b++) if (tb[t].cnt[a][b]) { tb[0].bb[a][b].grow(tb[t].bb[a][b]); tb[0].cnt[a][b] += tb[t].cnt[a][b]; }
            for (int axis = 0; axis < 3; axis++) {
                if (!use[axis]) continue;
                const aabb* bb = tb[0].bb[axis]; const uint32_t* cnt = tb[0].cnt[axis];
                float rarea[kBins]; uint32_t rcnt[kBins];
                aabb acc; acc.reset(); uint32_t ac = 0;
                for (int b = kBins - 1; b > 0; b--) { if (cnt[b]) acc.grow(bb[b]); ac += cnt[b]; rarea[b] = ac ? acc.area() : 0.0f; rcnt[b] = ac; }
                acc.reset(); ac = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    if (cnt[b]) acc.grow(bb[b]); ac += cnt[b];
                    if (ac == 0 || rcnt[b + 1] == 0) continue;
                    float cost = acc.area() * ac + rarea[b + 1] * rcnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }
        float area = nd.box.area();
        float leaf_cost = area * n, split_cost = area * node_cost + best_cost;   // SAH node cost 1, triangle cost 1 (SplitBVHBuilder.hpp Platform defaults)
        if (!force_median && (int)n <= max_leaf && (best_axis < 0 || leaf_cost <= split_cost)) return me;   // leaf
        uint32_t mid;
        if (best_axis >= 0 && !force_median) {
            const float* c = cen(best_axis);
            float lo = cb.lo[best_axis], scale = kBins / (cb.hi[best_axis] - lo);
            auto left = [&](uint32_t p) { int b = std::min(kBins - 1, std::max(0, (int)((c[p] - lo) * scale))); return b <= best_bin; };
            if (threads > 1) {   // count per chunk, then scatter both sides in order into the scratch array and copy back
                std::vector<uint32_t> nl(threads + 1, 0), first(threads + 1, begin);
                chunks(begin, end, threads, [&](uint32_t b0, uint32_t b1, int t) { uint32_t k = 0; for (uint32_t i = b0; i < b1; i++) k += left(idx[i]) ? 1u : 0u; nl[t + 1] = k; first[t] = b0; });
                first[threads] = end;
                for (int t = 0; t < threads; t++) nl[t + 1] += nl[t];
                mid = begin + nl[threads];
                chunks(begin, end, threads, [&](uint32_t b0, uint32_t b1, int t) {
                    uint32_t l = begin + nl[t], r = mid + (b0 - begin) - nl[t];
                    for (uint32_t i = b0; i < b1; i++) { const uint32_t p = idx[i]; if (left(p)) scratch[l++] = p; else scratch[r++] = p; }
                });
                chunks(begin, end, threads, [&](uint32_t b0, uint32_t b1, int) { std::memcpy(&idx[b0], &scratch[b0], (size_t)(b1 - b0) * 4); });
            } else {
                auto it = std::stable_partition(idx.begin() + begin, idx.begin() + end, left);   // stable like the threaded path: the tree does not depend on the thread count
                mid = (uint32_t)(it - idx.begin());
            }
        } else mid = begin;
        if (mid == begin || mid == end) {   // median split on the widest centroid axis (or by index when all centroids coincide)
            int axis = 0; float w = -1;
            for (int a = 0; a < 3; a++) { float e = cb.hi[a] - cb.lo[a]; if (e > w) { w = e; axis = a; } }
            mid = begin + n / 2;
            if (w > 0) { const float* c = cen(axis); std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end, [&](uint32_t a, uint32_t b) { return c[a] < c[b]; }); }
        }
        if (par_budget > 0 && n > 32768) {   // the children split this subtree's threads
            auto fut = std::async(std::launch::async, [&, mid, end, depth, par_budget]() { return build(mid, end, depth + 1, par_budget - 1); });
            int l = build(begin, mid, depth + 1, par_budget - 1);
            int r = fut.get();
            pool[me].left = l; pool[me].right = r;
        } else {
            int l = build(begin, mid, depth + 1, 0);
            int r = build(mid, end, depth + 1, 0);
            pool[me].left = l; pool[me].right = r;
        }
        return me;
    }
};

// Insertion-based re-optimisation of the finished binary tree (after Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding volume hierarchies", CGF 2013):
// a node x is taken out — its parent disappears, its sibling moves up — and put back as the sibling of the node y for which the summed surface area of the inner nodes
// drops most; leaves (up to max_leaf primitives each) move as units, so the leaf costs do not change and the inner areas are the whole SAH difference.
// The search walks up from x: with pivot A_k (the k-th ancestor of x's parent) the boxes above the pivot do not change — x stays below them — the path nodes below it
// shrink (gain R_k, accumulated on the way up), and the candidates y are the subtree of the pivot's other child, searched best-first with the bound
//   cost(insert below y) >= induced growth of y's ancestors inside that subtree + area(y u x) - area(y) + area(x).
// One pass visits the nodes in order of decreasing area and applies every move at once (the tree a later node sees is the current one).  Top-down binned SAH
// decides every split with what it sees at that level; this pass repairs the decisions that turned out badly further down — on the bench scenes the
// oracle-counted node visits per ray fall by the amounts in profiles/r05_reinsertion.log.
struct reinserter {
    std::vector<tmp_node>& N; int root; std::vector<int> par;
    reinserter(std::vector<tmp_node>& n, int r, int used) : N(n), root(r), par((size_t)used, -1) {
        for (int i = 0; i < used; i++) if (N[i].left >= 0) { par[N[i].left] = i; par[N[i].right] = i; }
    }
    int sibling(int x) const { const tmp_node& p = N[par[x]]; return p.left == x ? p.right : p.left; }
    static aabb join(const aabb& a, const aabb& b) { aabb u = a; u.grow(b); return u; }
    // The flattened tree stores a child's box on the 8-bit grid of its PARENT's box (flat4_node): under parent box P a child box b is what the rays meet grown by up to a
    // grid step extent(P) / 255 on every side.  A small subtree moved under a large node loses its tight box that way — the plain surface-area measure does not see it
    // (synthetic-sm-hard: -20 % inner area, +30 % leaf entries fetched).  infl = the area that growth adds to b.
    static float infl(const aabb& b, const aabb& P) {
        float e[3], g[3];
        for (int a = 0; a < 3; a++) { e[a] = b.hi[a] - b.lo[a]; g[a] = e[a] + (P.hi[a] - P.lo[a]) * (2.0f / 255.0f); }
        return 2.0f * (g[0] * g[1] + g[1] * g[2] + g[2] * g[0]) - 2.0f * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
    }
    void refit_up(int i) {
        for (; i >= 0; i = par[i]) {
            const aabb b = join(N[N[i].left].box, N[N[i].right].box);
            if (std::memcmp(&b, &N[i].box, sizeof(aabb)) == 0) break;
            N[i].box = b;
        }
    }
    // returns the gain (> 0) and the node to become x's sibling, or 0.  Read-only: a batch of searches runs on all host threads.
    float find(int x, int& best_y, std::vector<std::pair<int, float>>& stack) const {
        const aabb X = N[x].box; const float aX = X.area();
        float best = 0.0f; best_y = -1;
        const float old_x = infl(X, N[par[x]].box);
        int pivot = par[x], path_child = x;
        float R = N[pivot].box.area();           // R_0: x's parent disappears
        aabb nb = N[sibling(x)].box;             // N_0: what stands in the parent's place
        for (int k = 0;; k++) {
            const int s = N[pivot].left == path_child ? N[pivot].right : N[pivot].left;
            stack.clear(); stack.emplace_back(s, 0.0f);
            while (!stack.empty()) {
                const auto [y, c] = stack.back(); stack.pop_back();
                const aabb U = join(N[y].box, X);
                const float au = U.area();
                float g = R - (c + au);
                if (g > best) g -= infl(X, U) + infl(N[y].box, U) - old_x - infl(N[y].box, N[par[y]].box);   // what the move adds to the grid growth of x and y (weight 1; 4: no better)
                if (!(k == 0 && y == s) && g > best) { best = g; best_y = y; }    // (k == 0, y == s: where x is now)
                if (N[y].left >= 0) {
                    const float c2 = c + au - N[y].box.area();
                    if (R - (c2 + aX) > best) { stack.emplace_back(N[y].left, c2); stack.emplace_back(N[y].right, c2); }
                }
            }
            if (par[pivot] < 0) break;
            if (k >= 1) { nb = join(nb, N[s].box); R += N[pivot].box.area() - nb.area(); }   // the pivot becomes a path node: it shrinks to N_k
            path_child = pivot; pivot = par[pivot];
        }
        return best;
    }
    void move(int x, int y) {
        const int p = par[x], s0 = sibling(x), g = par[p];
        (N[g].left == p ? N[g].left : N[g].right) = s0; par[s0] = g;           // x's sibling takes the parent's place
        const int gy = par[y];
        (N[gy].left == y ? N[gy].left : N[gy].right) = p; par[p] = gy;          // the freed parent joins y and x where y was
        N[p].left = y; N[p].right = x; par[y] = p; par[x] = p;
        N[p].box = join(N[y].box, N[x].box);
        refit_up(g); refit_up(gy);
    }
    double inner_area() const { double a = 0; for (size_t i = 0; i < par.size(); i++) if (N[i].left >= 0) a += N[i].box.area(); return a; }
    int depth() const {   // nodes on the longest root-to-leaf path, counted as builder::max_depth does (root = 0)
        int best = 0; std::vector<std::pair<int, int>> st; st.emplace_back(root, 0);
        while (!st.empty()) { const auto [i, d] = st.back(); st.pop_back(); best = std::max(best, d); if (N[i].left >= 0) { st.emplace_back(N[i].left, d + 1); st.emplace_back(N[i].right, d + 1); } }
        return best;
    }
    bool below(int y, int x) const { for (; y >= 0; y = par[y]) if (y == x) return true; return false; }   // y inside the subtree of x
    // `fraction` of the nodes per pass, largest first.  Batches: the searches of a batch run in parallel on the tree as the previous batch left it, then its moves are applied
    // in order; a move whose nodes an earlier move of the batch touched is left for the next pass (its gain was computed on a tree that no longer exists).  The result does
    // not depend on the number of threads.  Measured on synthetic-SM (8.6 M leaf entries; profiles/r05_reinsertion.log): the summed inner area falls by 9 % with 2 passes over
    // all nodes (75 s on 8 host cores), by 12.5 % with 16 passes over the largest 3 % (25 s) — the large nodes are where the top-down build's early decisions sit; Bittner's
    // combined inefficiency measure instead of the plain area picks nodes that do no better.
    void run(int passes, float fraction) {
        constexpr size_t kBatch = 1024;   // 65536: a pass is 25 % faster and moves a third less (more of a batch's moves meet a touched node)
        const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::pair<float, int>> order;
        std::vector<std::pair<float, int>> found(kBatch);
        std::vector<uint8_t> touched(par.size(), 0); std::vector<int> marks;
        for (int pass = 0; pass < passes; pass++) {
            order.clear();
            for (int i = 0; i < (int)par.size(); i++) if (par[i] >= 0 && par[par[i]] >= 0) order.emplace_back(-N[i].box.area(), i);
            const size_t take = std::min(order.size(), (size_t)((double)order.size() * fraction) + 1);
            std::partial_sort(order.begin(), order.begin() + take, order.end());
            size_t moved = 0;
            for (size_t b0 = 0; b0 < take; b0 += kBatch) {
                const size_t nb = std::min(kBatch, take - b0);
                std::vector<std::thread> th;
                for (int t = 0; t < threads; t++) th.emplace_back([&, t]() {
                    std::vector<std::pair<int, float>> stack;
                    for (size_t q = (size_t)t; q < nb; q += (size_t)threads) {
                        const int x = order[b0 + q].second; int y = -1; float g = 0.0f;
                        if (par[x] >= 0 && par[par[x]] >= 0) g = find(x, y, stack);
                        found[q] = { g, y };
                    }
                });
                for (auto& x : th) x.join();
                marks.clear();
                for (size_t q = 0; q < nb; q++) {
                    const int x = order[b0 + q].second, y = found[q].second;
                    if (!(found[q].first > 0.0f) || y < 0) continue;
                    const int p = par[x]; if (p < 0 || par[p] < 0 || par[y] < 0) continue;
                    const int s0 = sibling(x), g = par[p], gy = par[y];
                    if (touched[x] || touched[y] || touched[p] || touched[s0] || touched[g] || touched[gy] || y == p || below(y, x)) continue;
                    move(x, y); moved++;
                    for (int m : { x, y, p, s0, g, gy }) { touched[m] = 1; marks.push_back(m); }
                }
                for (int m : marks) touched[m] = 0;
            }
            if (moved == 0) break;
        }
    }
};

void set_left(ctl_bvh_node& n, const aabb& b) { n.a[0] = b.lo[0]; n.a[1] = b.hi[0]; n.a[2] = b.lo[1]; n.a[3] = b.hi[1]; n.c[0] = b.lo[2]; n.c[1] = b.hi[2]; }
void set_right(ctl_bvh_node& n, const aabb& b) { n.b[0] = b.lo[0]; n.b[1] = b.hi[0]; n.b[2] = b.lo[1]; n.b[3] = b.hi[1]; n.c[2] = b.lo[2]; n.c[3] = b.hi[2]; }

struct emitter {
    const builder& B; bvh_result& out;
    int emit_leaf(const tmp_node& t) {
        uint32_t first = (uint32_t)out.leaf_prims.size();
        for (uint32_t i = t.begin; i < t.end; i++) { out.leaf_prims.push_back(B.idx[i]); out.leaf_last.push_back(i + 1 == t.end ? 1 : 0); }
        return ~(int)first;
    }
    // returns the child code of the subtree; `parent` is the parent's float4 index (SplitBVHBuilder.cpp:191-201)
    int emit(int ti, uint32_t parent) {
        const tmp_node& t = B.pool[ti];
        if (t.left < 0) return emit_leaf(t);
        uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        int a = emit(t.left, me * 4);
        int b = emit(t.right, me * 4);
        ctl_bvh_node& n = out.nodes[me];
        std::memset(&n, 0, sizeof(n));
        n.child0 = a; n.child1 = b; n.parent = parent;
        set_left(n, B.pool[t.left].box); set_right(n, B.pool[t.right].box);
        return (int)(me * 4);
    }
};

} // namespace

void build_bvh(const std::vector<aabb>& prim_boxes, int max_leaf, bool wrap_single_leaf, int max_depth_limit, bvh_result& out, float node_cost, int reinsertion_passes, float reinsertion_fraction) {
    out.nodes.clear(); out.leaf_prims.clear(); out.leaf_last.clear(); out.root = 0; out.max_depth = 0;
    if (prim_boxes.empty()) { out.root = kEmptyChild; return; }
    builder B(prim_boxes, max_leaf, max_depth_limit);
    B.node_cost = node_cost;
    int budget = 0; while ((1u << (budget + 1)) <= std::max(1u, std::thread::hardware_concurrency()) && budget < 6) budget++;   // <= 64 threads
    int root = B.build(0, (uint32_t)prim_boxes.size(), 0, std::max(budget, 3));
    out.max_depth = B.max_depth.load();
    if (reinsertion_passes > 0 && B.pool[root].left >= 0) {
        const int used = B.pool_used.load();
        std::vector<tmp_node> before(B.pool.begin(), B.pool.begin() + used);    // kept: a tree that came out deeper than the traversal stack allows is not used
        reinserter Rn(B.pool, root, used);
        const double a0 = Rn.inner_area();
        Rn.run(reinsertion_passes, reinsertion_fraction);
        const int d = Rn.depth();
        if (d > max_depth_limit) std::copy(before.begin(), before.end(), B.pool.begin());
        else out.max_depth = d;
        if (std::getenv("CTL_VERBOSE")) std::fprintf(stderr, "build_bvh: reinsertion %d pass(es): inner area %.6g -> %.6g, depth %d -> %d%s\n", reinsertion_passes, a0, Rn.inner_area(), B.max_depth.load(), d, d > max_depth_limit ? " (discarded: too deep)" : "");
    }
    emitter E{ B, out };
    out.leaf_prims.reserve(prim_boxes.size()); out.leaf_last.reserve(prim_boxes.size());
    const tmp_node& r = B.pool[root];
    if (r.left < 0) {
        int leaf = E.emit_leaf(r);
        if (wrap_single_leaf) {   // SplitBVHBuilder.cpp:176-189: root node = (leaf, none), right box = [0,0]
            ctl_bvh_node n; std::memset(&n, 0, sizeof(n));
            n.child0 = leaf; n.child1 = kEmptyChild; n.parent = 0xffffffffu;
            set_left(n, r.box);
            aabb z; for (int i = 0; i < 3; i++) z.lo[i] = z.hi[i] = 0.0f;
            set_right(n, z);
            out.nodes.push_back(n); out.root = 0;
        } else out.root = leaf;
    } else {
        out.root = E.emit(root, 0xffffffffu);   // == 0
    }
}

} // namespace ctl

namespace ctl {
namespace {
aabb node_child_box(const ctl_bvh_node& n, int which) {
    aabb b;
    if (which == 0) { b.lo[0] = n.a[0]; b.hi[0] = n.a[1]; b.lo[1] = n.a[2]; b.hi[1] = n.a[3]; b.lo[2] = n.c[0]; b.hi[2] = n.c[1]; }
    else { b.lo[0] = n.b[0]; b.hi[0] = n.b[1]; b.lo[1] = n.b[2]; b.hi[1] = n.b[3]; b.lo[2] = n.c[2]; b.hi[2] = n.c[3]; }
    return b;
}
struct collapser {
    const bvh_result& R; std::vector<wide4_node>& out; int max_depth = 0;
    int emit(int code2, const aabb& box, int depth) {   // code2: BVH2 inner-node code (float4 units)
        if (depth > max_depth) max_depth = depth;
        struct item { int code; aabb box; };
        item it[4]; int n = 0;
        const ctl_bvh_node& root = R.nodes[code2 / 4];
        if (root.child0 != 0x76543210) it[n++] = { root.child0, node_child_box(root, 0) };
        if (root.child1 != 0x76543210) it[n++] = { root.child1, node_child_box(root, 1) };
        while (n < 4) {
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < n; i++) if (it[i].code >= 0) { const float a = it[i].box.area(); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            const ctl_bvh_node& c = R.nodes[it[best].code / 4];
            const bool has0 = c.child0 != 0x76543210, has1 = c.child1 != 0x76543210;
            if (has0 && has1) { it[n++] = { c.child1, node_child_box(c, 1) }; it[best] = { c.child0, node_child_box(c, 0) }; }
            else if (has0) it[best] = { c.child0, node_child_box(c, 0) };
            else if (has1) it[best] = { c.child1, node_child_box(c, 1) };
            else break;
        }
        const int me = (int)out.size();
        out.emplace_back();
        out[me].box = box; out[me].n = n;
        for (int i = 0; i < 4; i++) { out[me].child[i] = 0x76543210; out[me].cbox[i].reset(); }
        for (int i = 0; i < n; i++) {
            const aabb cb = it[i].box;
            const int c = it[i].code < 0 ? it[i].code : emit(it[i].code, cb, depth + 1);
            out[me].child[i] = c; out[me].cbox[i] = cb;
        }
        return me;
    }
};

// SAH-optimal collapse (the dynamic programme of Ylitie, Karras & Laine, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs",
// HPG 2017, section 4.1, for 4 slots): for every BVH2 subtree v and k = 1..4 slots of a parent, F(v, k) = the cheapest way to hang v's primitives into at most k
// slots — one slot holding v as a wide node (area * node_cost + the best split of its two children over 4 slots), one slot holding ALL of v's primitives as one
// leaf (area * count * 1, when count <= max_leaf), or v's two children sharing the k slots.  The greedy collapse leaves 42 % of the nodes of the bench
// scene with two children (a binary bottom node over two one-triangle leaves becomes a wide node of its own); this one fills the slots and sizes the leaves.
struct dp_collapser {
    bvh_result& R; std::vector<wide4_node>& out; const float node_cost; const int max_leaf;
    int max_depth = 0;
    struct rec { float F[4]; uint32_t first, count; float area; uint8_t one_is_leaf; uint8_t split[4]; };   // split[k]: slots given to child 0 when v spreads over k + 1 slots (0 = v stays in one slot)
    std::vector<rec> T;   // per inner BVH2 node

    static bool inner(int code) { return code >= 0 && code != 0x76543210; }
    uint32_t leaf_count(int code) const { uint32_t n = 1; for (uint32_t e = (uint32_t)~code; !R.leaf_last[e]; e++) n++; return n; }
    // F(child, k) for k = 1..4 (index k - 1)
    void child_cost(int code, const aabb& box, float F[4]) const {
        if (code == 0x76543210) { F[0] = F[1] = F[2] = F[3] = 0.0f; return; }
        if (inner(code)) { for (int k = 0; k < 4; k++) F[k] = T[code / 4].F[k]; return; }
        const float c = box.area() * (float)leaf_count(code);
        F[0] = F[1] = F[2] = F[3] = c;
    }
    void solve() {
        T.resize(R.nodes.size());
        for (int v = (int)R.nodes.size() - 1; v >= 0; v--) {   // children are emitted after their parent (bvh_builder's emitter): reverse index order is a post-order
            const ctl_bvh_node& n = R.nodes[v];
            rec& t = T[v];
            aabb b0 = node_child_box(n, 0), b1 = node_child_box(n, 1), box; box.reset();
            const bool has0 = n.child0 != 0x76543210, has1 = n.child1 != 0x76543210;
            if (has0) box.grow(b0); if (has1) box.grow(b1);
            t.area = box.area();
            auto first_of = [&](int code) { return inner(code) ? T[code / 4].first : (uint32_t)~code; };
            auto count_of = [&](int code) { return inner(code) ? T[code / 4].count : leaf_count(code); };
            t.first = has0 ? first_of(n.child0) : first_of(n.child1);
            t.count = (has0 ? count_of(n.child0) : 0u) + (has1 ? count_of(n.child1) : 0u);
            float F0[4], F1[4];
            child_cost(n.child0, b0, F0); child_cost(n.child1, b1, F1);
            // D[k]: the two children share k + 1 slots (k = 1..3); a missing child takes none
            float D[4] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f }; uint8_t Di[4] = { 0, 0, 0, 0 };
            if (has0 && has1) {
                for (int k = 1; k < 4; k++) for (int i = 1; i <= k; i++) { const float c = F0[i - 1] + F1[k - i]; if (c < D[k]) { D[k] = c; Di[k] = (uint8_t)i; } }
            } else {
                const float* Fc = has0 ? F0 : F1;
                for (int k = 1; k < 4; k++) { D[k] = Fc[k]; Di[k] = has0 ? (uint8_t)(k + 1) : 0; }
            }
            const float as_node = t.area * node_cost + D[3];
            const float as_leaf = (int)t.count <= max_leaf ? t.area * (float)t.count : 3.402823466e+38f;
            t.one_is_leaf = as_leaf <= as_node;
            t.F[0] = t.one_is_leaf ? as_leaf : as_node; t.split[0] = 0;
            for (int k = 1; k < 4; k++) {
                t.F[k] = t.F[k - 1]; t.split[k] = t.split[k - 1] ? t.split[k - 1] : 0;
                uint8_t how = t.split[k - 1] ? (uint8_t)(0x80 | (k - 1)) : 0;   // 0x80 | j: use the arrangement found for j + 1 slots
                if (D[k] < t.F[k]) { t.F[k] = D[k]; how = (uint8_t)(0x40 | Di[k]); }   // 0x40 | i: spread over exactly k + 1 slots, i of them to child 0
                t.split[k] = how;
            }
        }
    }
    struct item { int code; aabb box; bool merged_leaf; };
    // the slots subtree `code` occupies when it may use up to k + 1 of them
    void collect(int code, const aabb& box, int k, item* it, int& n) {
        if (!inner(code)) { it[n++] = { code, box, false }; return; }
        const rec& t = T[code / 4];
        uint8_t how = t.split[k]; int kk = k;
        while (how & 0x80) { kk = how & 3; how = t.split[kk]; }
        if (!(how & 0x40)) { it[n++] = { code, box, t.one_is_leaf != 0 }; return; }   // one slot
        const int i = how & 7;
        const ctl_bvh_node& nd = R.nodes[code / 4];
        const bool has0 = nd.child0 != 0x76543210, has1 = nd.child1 != 0x76543210;
        if (has0 && has1) { collect(nd.child0, node_child_box(nd, 0), i - 1, it, n); collect(nd.child1, node_child_box(nd, 1), kk - i, it, n); }
        else if (has0) collect(nd.child0, node_child_box(nd, 0), kk, it, n);
        else collect(nd.child1, node_child_box(nd, 1), kk, it, n);
    }
    int emit(int code2, const aabb& box, int depth) {
        if (depth > max_depth) max_depth = depth;
        item it[4]; int n = 0;
        const ctl_bvh_node& nd = R.nodes[code2 / 4];
        const bool has0 = nd.child0 != 0x76543210, has1 = nd.child1 != 0x76543210;
        if (has0 && has1) {
            // the best split of the node's own 4 slots (D[3] of solve())
            float F0[4], F1[4]; child_cost(nd.child0, node_child_box(nd, 0), F0); child_cost(nd.child1, node_child_box(nd, 1), F1);
            int bi = 1; float bc = 3.402823466e+38f;
            for (int i = 1; i <= 3; i++) { const float c = F0[i - 1] + F1[3 - i]; if (c < bc) { bc = c; bi = i; } }
            collect(nd.child0, node_child_box(nd, 0), bi - 1, it, n); collect(nd.child1, node_child_box(nd, 1), 3 - bi, it, n);
        } else if (has0) collect(nd.child0, node_child_box(nd, 0), 3, it, n);
        else if (has1) collect(nd.child1, node_child_box(nd, 1), 3, it, n);
        const int me = (int)out.size();
        out.emplace_back();
        out[me].box = box; out[me].n = n;
        for (int i = 0; i < 4; i++) { out[me].child[i] = 0x76543210; out[me].cbox[i].reset(); }
        for (int i = 0; i < n; i++) {
            int c;
            if (!inner(it[i].code)) c = it[i].code;
            else if (it[i].merged_leaf) {   // the whole subtree as one leaf: its entries are consecutive in leaf_prims (depth-first emission); only the last one closes it
                const rec& t = T[it[i].code / 4];
                for (uint32_t e = t.first; e + 1 < t.first + t.count; e++) R.leaf_last[e] = 0;
                R.leaf_last[t.first + t.count - 1] = 1;
                c = ~(int)t.first;
            } else c = emit(it[i].code, it[i].box, depth + 1);
            out[me].child[i] = c; out[me].cbox[i] = it[i].box;
        }
        return me;
    }
};
} // namespace

namespace {
struct collapser8 {
    const bvh_result& R; std::vector<wide8_node>& out; int max_depth = 0;
    struct item { int code; aabb box; };
    // children of BVH2 node `code2` opened greedily into at most eight items
    int gather(int code2, item* it) const {
        int n = 0;
        const ctl_bvh_node& root = R.nodes[code2 / 4];
        if (root.child0 != 0x76543210) it[n++] = { root.child0, node_child_box(root, 0) };
        if (root.child1 != 0x76543210) it[n++] = { root.child1, node_child_box(root, 1) };
        while (n < 8) {
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < n; i++) if (it[i].code >= 0) { const float a = it[i].box.area(); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            const ctl_bvh_node& c = R.nodes[it[best].code / 4];
            const bool has0 = c.child0 != 0x76543210, has1 = c.child1 != 0x76543210;
            if (has0 && has1) { it[n++] = { c.child1, node_child_box(c, 1) }; it[best] = { c.child0, node_child_box(c, 0) }; }
            else if (has0) it[best] = { c.child0, node_child_box(c, 0) };
            else if (has1) it[best] = { c.child1, node_child_box(c, 1) };
            else break;
        }
        return n;
    }
    // iterative (explicit stack): the wide tree of a degenerate scene can be as deep as the BVH2
    void run(const aabb& root_box) {
        struct todo { int code2, me, depth; };
        std::vector<todo> stack;
        out.emplace_back(); out[0].box = root_box;
        stack.push_back({ 0, 0, 0 });
        while (!stack.empty()) {
            const todo t = stack.back(); stack.pop_back();
            if (t.depth > max_depth) max_depth = t.depth;
            item it[8]; const int n = gather(t.code2, it);
            // octant-ordered slots: greedy matching on cost(child, slot) = sign(slot) . (centroid(child) - centroid(node))
            float cen[3]; for (int k = 0; k < 3; k++) cen[k] = 0.5f * (out[t.me].box.lo[k] + out[t.me].box.hi[k]);
            float cost[8][8];
            for (int c = 0; c < n; c++) for (int s = 0; s < 8; s++) {
                float v = 0; for (int k = 0; k < 3; k++) { const float d = 0.5f * (it[c].box.lo[k] + it[c].box.hi[k]) - cen[k]; v += ((s >> k) & 1) ? d : -d; }
                cost[c][s] = v;
            }
            int slot_of[8]; bool cu[8] = {}, su[8] = {};
            for (int r = 0; r < n; r++) {
                int bc = -1, bs = -1; float bv = -3.402823466e+38f;
                for (int c = 0; c < n; c++) if (!cu[c]) for (int s = 0; s < 8; s++) if (!su[s] && cost[c][s] > bv) { bv = cost[c][s]; bc = c; bs = s; }
                cu[bc] = true; su[bs] = true; slot_of[bc] = bs;
            }
            wide8_node w; w.box = out[t.me].box;
            for (int s = 0; s < 8; s++) { w.child[s] = 0x76543210; w.cbox[s].reset(); }
            for (int c = 0; c < n; c++) {
                const int s = slot_of[c];
                w.cbox[s] = it[c].box;
                if (it[c].code < 0) w.child[s] = it[c].code;
                else { const int k = (int)out.size(); out.emplace_back(); out[k].box = it[c].box; w.child[s] = k; stack.push_back({ it[c].code, k, t.depth + 1 }); }
            }
            out[t.me] = w;
        }
    }
};
}  // namespace

void collapse_bvh8(const bvh_result& R, std::vector<wide8_node>& out, int& max_depth) {
    out.clear(); max_depth = 0;
    if (R.nodes.empty()) return;
    aabb box; box.reset();
    box.grow(node_child_box(R.nodes[0], 0)); if (R.nodes[0].child1 != 0x76543210) box.grow(node_child_box(R.nodes[0], 1));
    collapser8 C{ R, out };
    C.run(box);
    max_depth = C.max_depth;
}

void collapse_bvh4(bvh_result& R, std::vector<wide4_node>& out, int& max_depth, int mode, float node_cost, int max_leaf) {
    out.clear(); max_depth = 0;
    if (R.nodes.empty()) return;
    aabb box; box.reset();
    box.grow(node_child_box(R.nodes[0], 0)); if (R.nodes[0].child1 != 0x76543210) box.grow(node_child_box(R.nodes[0], 1));
    if (mode == 0) {
        collapser C{ R, out };
        C.emit(0, box, 0);
        max_depth = C.max_depth;
    } else {
        dp_collapser C{ R, out, node_cost, max_leaf };
        C.solve();
        C.emit(0, box, 0);
        max_depth = C.max_depth;
    }
}
} // namespace ctl
