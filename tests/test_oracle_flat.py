"""The oracle's traversal of the product's FLATTENED BVH (oracle/ocore.h traceRayFlat) against its restatement of the reference's
two-level traversal: same rays, bit-identical (t, u, v, triangle, node), every node format.  CPU only — this pins the checker that
supplies bench.py's N_inner / N_tri ("the CPU restatement in counting mode with the same BVH")."""
import numpy as np
import pytest

import cudatracerlib_amd as ctl
from cudatracerlib_amd import api, scenes


def rays_for(desc, n, seed):
    rs = np.random.RandomState(seed)
    lo, hi = np.array(desc.box_min[:]), np.array(desc.box_max[:])
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = rs.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), size=(n, 3))
    d = rs.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 4:7] = d; r[:, 3] = desc.ray_trace_eps; r[:, 7] = np.float32(3.402823466e+38)
    r[:6, 4:7] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)   # the 2^-80 guard
    return r


@pytest.mark.parametrize("fmt", [api.FLAT_Q8, api.FLAT_Q4, api.FLAT_F4, api.FLAT_F2])
def test_flat_traversal_equals_two_level(orc, fmt):
    for sc, n in ((scenes.synthetic_sm(32, 32, n_instances=60, subdiv=2), 6000), (scenes.cornell_box(32, 32, glass_sphere=True), 4000)):
        d = sc.desc
        rays = rays_for(d, n, 5)
        fb = api.FlatBvh(d, fmt)
        assert fb.desc.format == fmt and fb.desc.n_leaves > 0 and fb.desc.node_bytes == (128 if fmt in (api.FLAT_F4, api.FLAT_Q8) else 64)
        want, c2 = orc.intersect(d, rays, count=True)
        got, cf = orc.intersect(d, rays, count=True, flat=fb.desc)
        ties = (got["tri_idx"] != want["tri_idx"]) & (got["dist"] == want["dist"])
        same = ~ties
        for k in ("tri_idx", "node_idx"):
            assert np.array_equal(got[k][same], want[k][same]), k
        for k in ("dist", "u", "v"):
            assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
        assert ties.sum() <= n // 1000 and (want["tri_idx"] >= 0).mean() > 0.2
        assert cf["n_inst"] == 0 and cf["n_tri"] > 0 and cf["n_inner"] > 0
        occ_f = orc.intersect(d, rays, any_hit=True, flat=fb.desc)["tri_idx"] >= 0
        occ_2 = orc.intersect(d, rays, any_hit=True)["tri_idx"] >= 0
        assert np.array_equal(occ_f, occ_2)


@pytest.mark.parametrize("fmt", [api.FLAT_Q4, api.FLAT_Q8])
def test_early_split_clipping_keeps_every_hit(orc, fmt):
    """a scene with beams — thin triangles that cross the room diagonally, hundreds of times longer than the sphere triangles around them: the flattened BVH enters each as SEVERAL
    references (csrc/flatten.cpp: the same leaf entry under the boxes of the parts of the triangle), every (triangle, instance) pair is still there, and a traversal of it reports the
    hits of the reference's two-level traversal bit for bit — aimed rays along the beams and at their parts' seams included"""
    sc = scenes.beams_over_spheres()
    d = sc.desc
    fb = api.FlatBvh(d, fmt)
    L = fb.leaves()
    pairs = np.stack([L[:, 12] >> 1, L[:, 13]], axis=1)
    uniq, cnt = np.unique(pairs, axis=0, return_counts=True)
    n_tris = 2 + 40 * 320 + 48
    assert len(uniq) == n_tris and fb.desc.n_leaves > n_tris + 100 and cnt.max() >= 8          # the beams have many references ...
    beam_node = d.n_nodes - 1
    assert set(uniq[cnt > 1][:, 1].tolist()) <= {beam_node, 0}                                   # ... the spheres' triangles one each
    rays = rays_for(d, 6000, 11)
    # rays aimed at points ON the beams (uniformly along them: the seams of the parts are hit at random offsets) from random origins
    rs = np.random.RandomState(3)
    P = d.view("tri_data", np.uint32, d.n_tri_data, 8)   # (only used for the count; the targets come from the entries' Woop rows, as in tests/test_gpu_intersect.py)
    E = L[L[:, 13] == beam_node]
    R = E[:, :12].view(np.float32).astype(np.float64).reshape(-1, 3, 4)
    M = np.zeros((len(E), 4, 4)); M[:, 0] = R[:, 1]; M[:, 1] = R[:, 2]; M[:, 2] = R[:, 0]; M[:, 2, 3] *= -1; M[:, 3, 3] = 1
    Mi = np.linalg.inv(M)
    v2 = Mi[:, :3, 3]; v0 = v2 + Mi[:, :3, 0]; v1 = v2 + Mi[:, :3, 1]
    k = rs.randint(0, len(E), size=4000); a, b = rs.uniform(size=(2, 4000, 1)); flip = (a + b) > 1; a = np.where(flip, 1 - a, a); b = np.where(flip, 1 - b, b)
    target = v2[k] + a * (v0[k] - v2[k]) + b * (v1[k] - v2[k])           # (the beam node's transform is the identity)
    o = rs.uniform(-11, 11, size=(4000, 3)); o[:, 1] = rs.uniform(0.2, 8, size=4000)
    dirs = target - o; dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    aimed = np.zeros((4000, 8), np.float32); aimed[:, :3] = o; aimed[:, 4:7] = dirs; aimed[:, 3] = d.ray_trace_eps; aimed[:, 7] = np.float32(3.4e38)
    rays = np.concatenate([rays, aimed])
    want = orc.intersect(d, rays); got = orc.intersect(d, rays, flat=fb.desc)
    ties = (got["tri_idx"] != want["tri_idx"]) & (got["dist"] == want["dist"])
    same = ~ties
    assert ties.sum() <= 10 and (want["node_idx"] == beam_node).sum() > 1500
    for k2 in ("tri_idx", "node_idx"):
        assert np.array_equal(got[k2][same], want[k2][same]), k2
    for k2 in ("dist", "u", "v"):
        assert np.array_equal(got[k2][same].view(np.uint32), want[k2][same].view(np.uint32)), k2
    assert np.array_equal(orc.intersect(d, rays, any_hit=True, flat=fb.desc)["tri_idx"] >= 0, orc.intersect(d, rays, any_hit=True)["tri_idx"] >= 0)


def test_leaf_entries_carry_the_meshes_own_woop_rows_and_the_inverse_transform(orc):
    """a flattened leaf entry (128 B) = the object-space Woop rows of its triangle, bit for bit, (globalTri << 1 | last, node), and a copy of
    rows 0..2 and element (3,3) of that node's inverse transform"""
    sc = scenes.synthetic_sm(32, 32, n_instances=12, subdiv=1)
    d = sc.desc
    fb = api.FlatBvh(d, api.FLAT_Q4)
    L = fb.leaves()
    assert L.shape[1] == 32
    woop = d.view("woop", np.uint32, d.n_woop, 12)
    rows = {w.tobytes() for w in woop}
    assert all(L[i, :12].tobytes() in rows for i in range(len(L)))
    assert (L[:, 13] < d.n_nodes).all() and ((L[:, 12] >> 1) < d.n_tri_data).all()
    assert L[-1, 12] & 1
    inv = d.view("node_inv_transforms", np.uint32, d.n_nodes, 16)
    assert np.array_equal(L[:, 16:28], inv[L[:, 13], :12]) and np.array_equal(L[:, 28], inv[L[:, 13], 15])


def check_implied_links(fb):
    """flat4_node: the links a traversal step derives from the first 48 B (inner children = consecutive nodes, leaf children = consecutive
    entries, per-slot entry counts) are the explicit child[] words, for every node of the tree"""
    N = fb.nodes(); L = fb.leaves()
    assert N.shape[1] == 16
    meta = N[:, 3]; exist = (meta >> 24) & 15; leafm = (meta >> 28) & exist; innerm = exist & ~leafm & 15
    # a slot without a child: its link repeats a sibling's (flatten.cpp) — the first inner child, or, in a node of leaves only, the first leaf entry (leaf bit set without the exists bit)
    imp0 = api.FlatBvh.implied_links(N); ch0 = fb.child_links()
    for c in range(1, 4):
        gone = ((exist >> c) & 1) == 0
        all_leaves = gone & (innerm == 0)
        assert (((meta[all_leaves] >> (28 + c)) & 1) == 1).all() and np.array_equal(imp0[all_leaves, c], ch0[all_leaves, 0])
        some_inner = gone & (innerm != 0)
        first_inner = np.array([ch0[i, np.nonzero((innerm[i] >> np.arange(4)) & 1)[0][0]] for i in np.nonzero(some_inner)[0]], np.int32).reshape(-1)
        assert (((meta[some_inner] >> (28 + c)) & 1) == 0).all() and np.array_equal(imp0[some_inner, c] & ~3, first_inner)
    assert fb.desc.compact == 1
    imp = api.FlatBvh.implied_links(N)
    child = fb.child_links()      # the explicit links (host side); a compact tree's last 16 B per node hold the oriented slab instead (flat_slab.h)
    has_slab = np.zeros(len(N), bool)
    n_leaf_before = np.zeros(len(N), np.int64); n_inner_before = np.zeros(len(N), np.int64)
    inner_base = np.full(len(N), -1, np.int64); leaf_base = np.full(len(N), -1, np.int64)
    for c in range(4):
        is_leaf = ((leafm >> c) & 1) == 1; is_inner = ((innerm >> c) & 1) == 1
        assert np.array_equal(imp[is_leaf, c], child[is_leaf, c])
        assert np.array_equal(imp[is_inner, c] & ~3, child[is_inner, c]) and ((imp[is_inner, c] & 2) == 0).all()
        assert (child[~is_leaf & ~is_inner, c] == 0x76543210).all()
        # the layout behind the links: inner children are consecutive nodes, the entries of the leaf children consecutive entries, both in slot order
        first_i = is_inner & (inner_base < 0); inner_base[first_i] = child[first_i, c] // 4
        assert np.array_equal(child[is_inner, c] // 4, (inner_base + n_inner_before)[is_inner])
        first_l = is_leaf & (leaf_base < 0); leaf_base[first_l] = ~child[first_l, c]
        assert np.array_equal(~child[is_leaf, c], (leaf_base + n_leaf_before)[is_leaf])
        first = (~child[is_leaf, c]).astype(np.int64); k = np.ones(len(first), np.int64)
        for j in range(3):
            more = (L[np.minimum(first + k - 1, len(L) - 1), 12] & 1) == 0; k = k + (more & (k == j + 1))
        assert (L[first + k - 1, 12] & 1).all() and k.max(initial=1) <= 4
        cnt_c = np.zeros(len(N), np.int64); cnt_c[is_leaf] = k
        # slab flag of an inner child: the bit the traversal hands down in the link
        has_slab[(child[is_inner, c] // 4)[(imp[is_inner, c] & 1) == 1]] = True
        n_leaf_before += cnt_c; n_inner_before += is_inner
    has_slab[0] = bool(fb.desc.root_slab)
    assert has_slab.sum() == fb.desc.n_slab_nodes and (leafm[has_slab] != 0).mean() > 0.5      # nodes with leaf children, and their parents (flatten.cpp: an inner child with a small subtree gets an interval too)


def test_implied_child_links_of_the_quantised_nodes(orc):
    for sc in (scenes.synthetic_sm(32, 32, n_instances=40, subdiv=2), scenes.cornell_box(32, 32, glass_sphere=True)):
        check_implied_links(api.FlatBvh(sc.desc, api.FLAT_Q4))


def check_q8_layout(fb):
    """flat8_node (csrc/flat8.h): the links a traversal step derives — inner child = first inner child + rank of the slot among the inner slots, leaf entry = first entry of
    the node + rank among the leaf slots — are the explicit links; every leaf slot is ONE entry; empty slots carry inverted boxes; B of an inner slot = the child has leaf
    slots or a slab; slots are octant-ordered (the child on the + side of an axis sits in a slot with that axis' bit set more often than not)"""
    N = fb.nodes(); L = fb.leaves(); ch = fb.child_links()
    assert N.shape[1] == 32 and ch.shape == (len(N), 8) and fb.desc.compact == 1
    imask = N[:, 3] >> 24; B = N[:, 4] >> 24; base = N[:, 4] & 0xffffff; leaf_base = N[:, 5]
    lmask = B & ~imask; heavy = B & imask
    assert (L[:, 12] & 1).all()                                     # every entry closes its leaf
    n_in = np.zeros(len(N), np.int64); n_lf = np.zeros(len(N), np.int64)
    has_leaf = lmask != 0; has_slab = N[:, 6] != 0
    for s in range(8):
        inner = ((imask >> s) & 1) == 1; leaf = ((lmask >> s) & 1) == 1; empty = ~inner & ~leaf
        assert np.array_equal(ch[inner, s], (base + n_in)[inner].astype(np.int32))
        assert np.array_equal(~ch[leaf, s], (leaf_base + n_lf)[leaf].astype(np.int32))
        assert (ch[empty, s] == 0x76543210).all()
        byte = lambda w0: (N[:, w0 + (s >> 2)] >> (8 * (s & 3))) & 255
        for lo, hi in ((8, 14), (10, 16), (12, 18)):
            assert (byte(lo)[empty] == 255).all() and (byte(hi)[empty] == 0).all()
            assert (byte(lo)[~empty] <= byte(hi)[~empty]).all()
        kids = ch[inner, s]
        assert np.array_equal(((heavy >> s) & 1)[inner] == 1, (has_leaf | has_slab)[kids])
        n_in += inner; n_lf += leaf
    assert n_lf.sum() == len(L) and bool(fb.desc.root_slab) == bool(has_leaf[0] or has_slab[0]) and has_slab.sum() == fb.desc.n_slab_nodes
    # children start right behind their parents' blocks: node 0 is the root, every other node is someone's inner child exactly once
    refs = np.sort(ch[(ch >= 0) & (ch != 0x76543210)]); assert np.array_equal(refs, np.arange(1, len(N)))
    # octant order: centre of the child's box (codes) against the centre of the node's children, per axis
    agree = tot = 0
    for axis, (lo, hi) in enumerate(((8, 14), (10, 16), (12, 18))):
        cen = np.zeros((len(N), 8)); ex = np.zeros((len(N), 8), bool)
        for s in range(8):
            cen[:, s] = ((N[:, lo + (s >> 2)] >> (8 * (s & 3))) & 255).astype(float) + ((N[:, hi + (s >> 2)] >> (8 * (s & 3))) & 255)
            ex[:, s] = ((imask | lmask) >> s) & 1 == 1
        mean = (cen * ex).sum(1) / np.maximum(ex.sum(1), 1)
        for s in range(8):
            side = cen[:, s] - mean; sel = ex[:, s] & (np.abs(side) > 16)
            agree += ((side[sel] > 0) == bool((s >> axis) & 1)).sum(); tot += sel.sum()
    assert agree > 0.8 * tot, (agree, tot)


def test_layout_of_the_8_wide_nodes(orc):
    for sc in (scenes.synthetic_sm(32, 32, n_instances=40, subdiv=2), scenes.cornell_box(32, 32, glass_sphere=True)):
        fb = api.FlatBvh(sc.desc, api.FLAT_Q8)
        assert fb.desc.format == api.FLAT_Q8
        check_q8_layout(fb)


def test_render_counts_in_flat_mode(orc):
    """orc.render(flat=..., counts=...) renders the same image as the two-level oracle and reports per-ray node / triangle visits"""
    sc = scenes.cornell_box(24, 24, glass_sphere=True)
    d = sc.desc
    tables = orc.sequence_tables(1)
    want, rays = orc.render(d, 24, 24, n_passes=1, tables=tables, max_path_length=4)
    fb = api.FlatBvh(d, api.FLAT_F4)
    counts = {}
    got, rays_f = orc.render(d, 24, 24, n_passes=1, tables=tables, max_path_length=4, flat=fb.desc, counts=counts)
    assert rays == rays_f and np.array_equal(got[..., 6], want[..., 6])
    assert np.allclose(got[..., :3], want[..., :3], rtol=1e-6, atol=1e-7)
    assert counts["path_rays"] + counts["occ_rays"] == rays and counts["path_inner"] > counts["path_rays"] and counts["path_inst"] == 0


def test_sah_optimal_collapse_builds_a_valid_smaller_tree():
    """CTL_FLAT_COLLAPSE=1 (the dynamic-programming collapse of bvh_builder.h; a measurement knob: only the -DCTL_MEASUREMENT_KNOBS build of the library
    reads it, once per process, hence the child process with CTL_AMD_LIB): the tree
    has fewer 4-wide nodes than the greedy one, leaves of up to four entries, implied links that equal the explicit ones, and the oracle's traversal
    of it reports the two-level (t, u, v, triangle, node) bit for bit."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from cudatracerlib_amd import api, scenes
import oracle
sys.path.insert(0, %r)
from test_oracle_flat import rays_for, check_implied_links
orc = oracle.Oracle()
sc = scenes.synthetic_sm(32, 32, n_instances=60, subdiv=2)
d = sc.desc
fb = api.FlatBvh(d, api.FLAT_Q4)
N, L = fb.nodes(), fb.leaves()
last = L[:, 12] & 1
sizes = np.diff(np.concatenate([[-1], np.nonzero(last)[0]]))
assert sizes.max() <= 4 and sizes.min() >= 1
check_implied_links(fb)
rays = rays_for(d, 6000, 5)
want = orc.intersect(d, rays)
got = orc.intersect(d, rays, flat=fb.desc)
same = ~((got["tri_idx"] != want["tri_idx"]) & (got["dist"] == want["dist"]))
assert same.mean() > 0.999
for k in ("tri_idx", "node_idx"):
    assert np.array_equal(got[k][same], want[k][same]), k
for k in ("dist", "u", "v"):
    assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
print("NODES", fb.desc.n_nodes, "MULTI", int((sizes > 1).sum()))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("0", "1"):
        env = dict(os.environ, CTL_FLAT_COLLAPSE=mode, CTL_AMD_LIB=os.path.join(root, "cudatracerlib_amd", "libctl_knobs.so"))
        r = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests"))], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("NODES")][-1].split()
        out[mode] = (int(line[1]), int(line[3]))
    assert out["1"][0] < 0.9 * out["0"][0], out     # fewer wide nodes
    assert out["1"][1] > out["0"][1], out           # some leaves were merged


def test_product_library_reads_no_measurement_knob():
    """csrc/knobs.h: knob_env() is getenv only in the -DCTL_MEASUREMENT_KNOBS build.  The product library builds the same tree whatever the
    environment says (here: the collapse and the leaf-size knobs), and the knobs build does react."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, hashlib
sys.path.insert(0, %r)
from cudatracerlib_amd import api, scenes
api.set_cache_dir("")
sc = scenes.synthetic_sm(32, 32, n_instances=30, subdiv=2)
fb = api.FlatBvh(sc.desc, api.FLAT_Q4)
print("TREE", fb.desc.n_nodes, hashlib.sha1(fb.nodes().tobytes()).hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(lib, **knobs):
        env = dict(os.environ, **knobs)
        env.pop("CTL_AMD_LIB", None)
        if lib:
            env["CTL_AMD_LIB"] = os.path.join(root, "cudatracerlib_amd", lib)
        r = subprocess.run([sys.executable, "-c", code % root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("TREE")][-1]

    plain = run(None)
    assert run(None, CTL_FLAT_COLLAPSE="1", CTL_FLAT_MAX_LEAF="4", CTL_FLAT_SLAB_USEFUL="2.0", CTL_FLAT_BFS_TOP="0") == plain
    assert run("libctl_knobs.so") == plain
    assert run("libctl_knobs.so", CTL_FLAT_COLLAPSE="1", CTL_FLAT_MAX_LEAF="4") != plain


def test_missing_children_have_inverted_boxes():
    """flat4_node: a slot without a child carries lo = 255 / hi = 0 on every axis (the kernel has no per-slot "child exists" test), a slot with a child lo <= hi"""
    for sc in (scenes.synthetic_sm(32, 32, n_instances=40, subdiv=2), scenes.cornell_box(32, 32, glass_sphere=True)):
        fb = api.FlatBvh(sc.desc, api.FLAT_Q4)      # (the arrays belong to it)
        N = fb.nodes()
        exist = (N[:, 3] >> 24) & 15
        assert ((exist != 15).sum() > 0)
        for c in range(4):
            has = ((exist >> c) & 1) == 1
            for lo_w, hi_w in ((4, 5), (6, 7), (8, 9)):
                lo = (N[:, lo_w] >> (8 * c)) & 255; hi = (N[:, hi_w] >> (8 * c)) & 255
                assert (lo[has] <= hi[has]).all()
                assert (lo[~has] == 255).all() and (hi[~has] == 0).all()


EXPLICIT_LINKS_CODE = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from cudatracerlib_amd import api, scenes
import oracle
sys.path.insert(0, %r)
from test_oracle_flat import rays_for
orc = oracle.Oracle()
sc = scenes.synthetic_sm(32, 32, n_instances=60, subdiv=2)
d = sc.desc
fb = api.FlatBvh(d, api.FLAT_Q4)
assert fb.desc.compact == 0 and fb.desc.n_slab_nodes == 0 and fb.desc.root_slab == 0
N = fb.nodes()
ch = fb.child_links(); stored = N[:, 12:16].view(np.int32); gone = ch == 0x76543210
assert np.array_equal(stored[~gone], ch[~gone])                               # the last 16 B of every node are its explicit links ...
assert np.array_equal(stored[gone], np.broadcast_to(ch[:, :1], ch.shape)[gone])   # ... and a slot without a child repeats the first child's (flatten.cpp)
rays = rays_for(d, 6000, 5)
want = orc.intersect(d, rays)
got = orc.intersect(d, rays, flat=fb.desc)
same = ~((got["tri_idx"] != want["tri_idx"]) & (got["dist"] == want["dist"]))
assert same.mean() > 0.999
for k in ("tri_idx", "node_idx"):
    assert np.array_equal(got[k][same], want[k][same]), k
for k in ("dist", "u", "v"):
    assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
if %r:
    gpu = api
    assert gpu.device_count() >= 1
    scene = gpu.Scene(d, flatten=True)
    for any_hit in (False, True):
        g = gpu.intersect(scene, rays, any_hit=any_hit); w = orc.intersect(d, rays, any_hit=any_hit, flat=fb.desc)
        if any_hit:
            assert np.array_equal(g["tri_idx"] >= 0, w["tri_idx"] >= 0)
        else:
            for k in ("tri_idx", "node_idx"):
                assert np.array_equal(g[k], w[k]), k
            for k in ("dist", "u", "v"):
                assert np.array_equal(g[k].view(np.uint32), w[k].view(np.uint32)), k
print("EXPLICIT OK")
'''


def run_explicit_links_child(on_gpu):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CTL_FLAT_FORCE_EXPLICIT="1", CTL_AMD_LIB=os.path.join(root, "cudatracerlib_amd", "libctl_knobs.so"))
    r = subprocess.run([sys.executable, "-c", EXPLICIT_LINKS_CODE % (root, os.path.join(root, "tests"), on_gpu)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "EXPLICIT OK" in r.stdout, r.stderr[-2000:]


def test_tree_with_explicit_links():
    """The form a scene beyond 2^24 nodes or 2^26 - 15 leaf entries gets (flatten.h: no implied links, the last 16 B of a node are child[4], no slabs), built for a
    small scene through CTL_FLAT_FORCE_EXPLICIT=1 (knobs build, child process): the oracle's traversal of it reports the two-level (t, u, v, triangle, node) bit for bit."""
    run_explicit_links_child(False)
