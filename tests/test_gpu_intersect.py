"""HIP intersect kernel vs the oracle's restatement of the reference traversal.
Bar: bit-exact (t, u, v, triangle, node) — the kernel evaluates the accepting arithmetic in the reference's order."""
import numpy as np
import pytest
from cudatracerlib_amd import scenes

pytestmark = pytest.mark.gpu


def camera_and_random_rays(desc, n, seed, any_tmax=False):
    rs = np.random.RandomState(seed)
    lo, hi = np.array(desc.box_min[:]), np.array(desc.box_max[:])
    o = rs.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), size=(n, 3))
    d = rs.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = o; rays[:, 3] = desc.ray_trace_eps; rays[:, 4:7] = d
    rays[:, 7] = rs.uniform(0.05, 1.0, size=n) * np.linalg.norm(hi - lo) if any_tmax else np.float32(3.402823466e+38)
    # degenerate directions: axis-aligned, zero components (the 2^-80 guard, TraceHelper.cu:417-420)
    rays[:6, 4:7] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    return rays


def check(gpu, orc, desc, rays, any_hit):
    scene = gpu.Scene(desc)
    got = gpu.intersect(scene, rays, any_hit=any_hit)
    want = orc.intersect(desc, rays, any_hit=any_hit)
    if any_hit:
        # any-hit: which triangle is found first depends on traversal order; occlusion itself must agree
        assert np.array_equal(got["tri_idx"] >= 0, want["tri_idx"] >= 0)
        return
    for k in ("tri_idx", "node_idx"):
        bad = np.nonzero(got[k] != want[k])[0]
        # equal-t ties between different triangles may resolve differently; everything else must match
        assert all(got["dist"][i] == want["dist"][i] for i in bad), (k, bad[:10])
        assert len(bad) <= len(rays) // 1000
    same = got["tri_idx"] == want["tri_idx"]
    for k in ("dist", "u", "v"):
        assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
    assert (want["tri_idx"] >= 0).mean() > 0.2   # the test actually hits geometry


@pytest.mark.parametrize("any_hit", [False, True])
def test_cornell_glass(gpu, orc, any_hit):
    sc = scenes.cornell_box(64, 64, glass_sphere=True)
    check(gpu, orc, sc.desc, camera_and_random_rays(sc.desc, 20000, 1, any_tmax=any_hit), any_hit)


@pytest.mark.parametrize("any_hit", [False, True])
def test_instanced_scene(gpu, orc, any_hit):
    sc = scenes.synthetic_sm(64, 64, n_instances=300, subdiv=2)
    check(gpu, orc, sc.desc, camera_and_random_rays(sc.desc, 30000, 2, any_tmax=any_hit), any_hit)


def test_empty_and_ragged(gpu, orc):
    sc = scenes.cornell_box(32, 32)
    scene = gpu.Scene(sc.desc)
    assert len(gpu.intersect(scene, np.zeros((0, 8), np.float32))) == 0
    for n in (1, 63, 64, 65, 1000):   # not a multiple of the wave size
        rays = camera_and_random_rays(sc.desc, max(n, 6), 3)[:n]
        got, want = gpu.intersect(scene, rays), orc.intersect(sc.desc, rays)
        assert np.array_equal(got["tri_idx"], want["tri_idx"]) and np.array_equal(got["dist"].view(np.uint32), want["dist"].view(np.uint32))


def test_traversal_counts_match_oracle(gpu, orc):
    """N_inner / N_tri / N_inst of the roofline formula — same BVH, same rays, same visit counts as the
    restated wavefront traversal up to traversal-order effects (< 15 %)."""
    sc = scenes.synthetic_sm(64, 64, n_instances=200, subdiv=2)
    rays = camera_and_random_rays(sc.desc, 20000, 4)
    scene = gpu.Scene(sc.desc)
    g = gpu.intersect_count(scene, rays)
    _, o = orc.intersect(sc.desc, rays, count=True)
    for k in ("n_inner", "n_tri", "n_inst"):
        assert abs(g[k] - o[k]) <= 0.15 * o[k], (k, g[k], o[k])


def check_flat(gpu, orc, desc, rays, any_hit, fmt):
    """the flattened layout against the oracle's restatement of the reference's TWO-LEVEL traversal: bit for bit, equal-t ties excepted"""
    scene = gpu.Scene(desc, flatten=True, flat_format=fmt)
    got = gpu.intersect(scene, rays, any_hit=any_hit)
    want = orc.intersect(desc, rays, any_hit=any_hit)
    if any_hit:   # which triangle is found first depends on the visiting order; occlusion itself must agree
        assert np.array_equal(got["tri_idx"] >= 0, want["tri_idx"] >= 0)
        return got
    for k in ("tri_idx", "node_idx"):
        bad = np.nonzero(got[k] != want[k])[0]
        assert all(got["dist"][i] == want["dist"][i] for i in bad), (k, bad[:10])      # only rays that hit two triangles at the same t
        assert len(bad) <= len(rays) // 1000
    same = got["tri_idx"] == want["tri_idx"]
    for k in ("dist", "u", "v"):
        assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
    assert (want["tri_idx"] >= 0).mean() > 0.2
    return got


@pytest.mark.parametrize("fmt", ["q8", "q4"])   # f4 / f2 are measurement builds (-DCTL_FLAT_EXPERIMENTS)
@pytest.mark.parametrize("any_hit", [False, True])
def test_flattened_world_space_bvh(gpu, orc, any_hit, fmt):
    """CTL_SCENE_FLATTEN: one world-space BVH over all instanced triangles, every node format.  The tree only culls — each leaf entry is
    evaluated with the reference's instance-transform + Woop arithmetic — so (t, u, v, triangle, node) equal the two-level traversal."""
    sc = scenes.synthetic_sm(64, 64, n_instances=300, subdiv=2)
    check_flat(gpu, orc, sc.desc, camera_and_random_rays(sc.desc, 30000, 7, any_tmax=any_hit), any_hit, fmt)


@pytest.mark.parametrize("fmt", ["q8", "q4"])
@pytest.mark.parametrize("any_hit", [False, True])
def test_flattened_bvh_with_split_references(gpu, orc, any_hit, fmt):
    """a scene whose long diagonal beams the flattened BVH enters as several references each (early split clipping, csrc/flatten.cpp; tests/test_oracle_flat.py pins the structure):
    the kernels report the two-level traversal's hits bit for bit — a triangle met twice is accepted once"""
    from cudatracerlib_amd import api
    sc = scenes.beams_over_spheres()
    fb = api.FlatBvh(sc.desc, api.FLAT_FORMATS[fmt])
    assert fb.desc.n_leaves > 2 + 40 * 320 + 48 + 100
    got = check_flat(gpu, orc, sc.desc, camera_and_random_rays(sc.desc, 30000, 7, any_tmax=any_hit), any_hit, fmt)
    if not any_hit:
        assert (got["node_idx"] == sc.desc.n_nodes - 1).sum() > 300      # rays that end on a beam


def test_flattened_rays_through_vertices_and_edges(gpu, orc):
    """rays aimed exactly at mesh vertices, edge midpoints and centroids.  At a vertex several triangles are hit at t values one unit in the last
    place apart, and WHICH of them a traversal reports depends on the last bit of its box tests — between the reference's own two-level traversal
    and a flat one over the same triangles as much as here (the oracle's two traversals differ on 2 % of these rays).  So: the GPU must equal the
    oracle's traversal of the SAME flattened arrays up to such near-ties, and every reported hit must be the reference's evaluation of that triangle."""
    from cudatracerlib_amd import api
    sc = scenes.synthetic_sm(64, 64, n_instances=40, subdiv=2)
    d = sc.desc
    fb = api.FlatBvh(d, api.FLAT_Q4)
    L = fb.leaves()[::3]
    # object-space vertices from the entries' Woop rows: inverse of [b; c; a] (TriIntersectorData.cu:20-32), then through the node's transform
    R = L[:, :12].view(np.float32).astype(np.float64).reshape(-1, 3, 4)
    M = np.zeros((len(L), 4, 4)); M[:, 0] = R[:, 1]; M[:, 1] = R[:, 2]; M[:, 2] = R[:, 0]; M[:, 2, 3] *= -1; M[:, 3, 3] = 1
    keep = np.abs(np.linalg.det(M)) > 1e-12
    Mi = np.linalg.inv(M[keep])
    xf = d.view("node_transforms", np.float32, d.n_nodes, 16).astype(np.float64).reshape(-1, 4, 4)[L[keep, 13]]
    def world(p):
        return np.einsum("nij,nj->ni", xf[:, :3, :3], p) + xf[:, :3, 3]
    v2 = Mi[:, :3, 3]; v0 = world(v2 + Mi[:, :3, 0]); v1 = world(v2 + Mi[:, :3, 1]); v2 = world(v2)
    targets = np.concatenate([v0, v1, 0.5 * (v0 + v1), 0.5 * (v1 + v2), (v0 + v1 + v2) / 3.0])
    rs = np.random.RandomState(9)
    o = np.array(d.box_max[:]) * rs.uniform(-0.9, 0.9, size=(len(targets), 3))
    dirs = targets - o; dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.zeros((len(targets), 8), np.float32); rays[:, :3] = o; rays[:, 4:7] = dirs; rays[:, 3] = d.ray_trace_eps; rays[:, 7] = np.float32(3.4e38)
    scene = gpu.Scene(d, flatten=True)
    got = gpu.intersect(scene, rays)
    want = orc.intersect(d, rays, flat=fb.desc)
    assert np.array_equal(got["tri_idx"] >= 0, want["tri_idx"] >= 0)
    same = (got["tri_idx"] == want["tri_idx"]) & (got["node_idx"] == want["node_idx"])
    for k in ("dist", "u", "v"):
        assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
    other = ~same
    assert other.mean() < 0.03
    ulp = np.abs(got["dist"][other].view(np.int32).astype(np.int64) - want["dist"][other].view(np.int32).astype(np.int64))
    assert (ulp <= 2).all()                                                         # near-ties only
    # every reported hit is the reference's evaluation of that triangle: re-trace each ray with (tmin, tmax) closed in around the reported t
    hitm = got["tri_idx"] >= 0
    probe = rays[hitm].copy()
    probe[:, 7] = np.nextafter(np.nextafter(got["dist"][hitm], np.float32(np.inf)), np.float32(np.inf))
    probe[:, 3] = np.nextafter(np.nextafter(got["dist"][hitm], np.float32(0)), np.float32(0))
    again = orc.intersect(d, probe)
    agree = again["tri_idx"] == got["tri_idx"][hitm]
    assert agree.mean() > 0.9                                                       # (where several triangles share the interval the probe may return another one)
    for k in ("dist", "u", "v"):
        assert np.array_equal(again[k][agree].view(np.uint32), got[k][hitm][agree].view(np.uint32)), k
    occ = gpu.intersect(scene, rays, any_hit=True)["tri_idx"] >= 0
    assert np.array_equal(occ, orc.intersect(d, rays, any_hit=True, flat=fb.desc)["tri_idx"] >= 0)    # the oracle's any-hit traversal of the SAME arrays: equal
    # the reference's own two-level traversal: equal but for rays ALONG an edge that two triangles share (the diagonal of the floor quad — its midpoint is a target once per reference of
    # the floor triangles): the flat traversal reports the far floor triangle there, the two-level one passes between the two (2 of 20595 rays, with or without split clipping)
    assert (occ != (orc.intersect(d, rays, any_hit=True)["tri_idx"] >= 0)).mean() < 5e-4


def test_flattened_cornell_and_ragged(gpu, orc):
    sc = scenes.cornell_box(64, 64, glass_sphere=True)
    check_flat(gpu, orc, sc.desc, camera_and_random_rays(sc.desc, 20000, 11), False, "q4")
    scene = gpu.Scene(sc.desc, flatten=True)
    assert len(gpu.intersect(scene, np.zeros((0, 8), np.float32))) == 0
    for n in (1, 63, 64, 65, 1000):
        rays = camera_and_random_rays(sc.desc, max(n, 6), 3)[:n]
        got, want = gpu.intersect(scene, rays), orc.intersect(sc.desc, rays)
        assert np.array_equal(got["tri_idx"], want["tri_idx"]) and np.array_equal(got["dist"].view(np.uint32), want["dist"].view(np.uint32))


@pytest.mark.parametrize("fmt", ["q8", "q4"])
def test_flattened_counts_against_the_oracle_on_the_same_arrays(gpu, orc, fmt):
    """N_inner / N_tri from the CPU restatement in counting mode with the SAME BVH.  The oracle walks the product's own
    flattened arrays depth-first; the kernel postpones leaves and descends speculatively, so it may visit somewhat more nodes, never fewer
    leaf entries than the closest hit needs."""
    from cudatracerlib_amd import api
    sc = scenes.synthetic_sm(64, 64, n_instances=200, subdiv=2)
    rays = camera_and_random_rays(sc.desc, 20000, 4)
    fb = api.FlatBvh(sc.desc, api.FLAT_FORMATS[fmt])
    want_hits, o = orc.intersect(sc.desc, rays, count=True, flat=fb.desc)
    two_level = orc.intersect(sc.desc, rays)
    assert np.array_equal(want_hits["dist"].view(np.uint32), two_level["dist"].view(np.uint32))    # the oracle's two traversals agree
    scene = gpu.Scene(sc.desc, flatten=True, flat_format=fmt)
    g = gpu.intersect_count(scene, rays)
    assert g["n_inst"] == 0 and o["n_inst"] == 0
    assert 0.98 * o["n_inner"] <= g["n_inner"] <= 1.35 * o["n_inner"], (g, o)
    assert 0.98 * o["n_tri"] <= g["n_tri"] <= 1.35 * o["n_tri"], (g, o)


def test_flattened_tree_with_explicit_links(gpu, orc):
    """the explicit-link form of the flattened tree (what a scene too large for the implied links gets; CTL_FLAT_FORCE_EXPLICIT=1 in the knobs build, child process):
    the kernels read child[4] from the node's last 16 B — bit-exact against the oracle's traversal of the same arrays"""
    from test_oracle_flat import run_explicit_links_child
    run_explicit_links_child(True)


def test_far_ray_origins_against_tiny_nodes(gpu, orc):
    """millimetre-sized instances hit from 3 .. 3e4 units away.  From far enough the entry and exit planes of a small node round to the same distance and the 8-bit boxes
    stop culling — including the inverted box of a slot WITHOUT a child, whose link the kernel then follows (flatten.cpp gives such a slot a sibling's link, so that
    the walk ends).  Box culling in fp32 is only reliable while the ray origin is within ~1e5 triangle sizes — the reference's own two-level traversal and a flat one
    then disagree on which triangles they look at (the oracle's two traversals: 0 rays of 8000 at 10 units, 9 at 100, 86 at 1000 on this scene) — so: from 3 units the
    GPU equals the oracle's traversal of the same arrays; from everywhere the traversal ends, agrees on hit / miss for > 97 % of the rays, and every hit it reports is the
    reference's evaluation of that triangle."""
    from cudatracerlib_amd import api
    V, F = scenes.icosphere(2)
    Pb, Ib, Nb = scenes.unit_box()
    meshes = [dict(V=V.astype(np.float32), F=F, N=None, material=("diffuse", (0.7, 0.7, 0.7))), dict(V=Pb, F=Ib, N=Nb, material=("diffuse", (0.2, 0.5, 0.7)))]
    rs = np.random.RandomState(21)
    nodes = []
    for k in range(12):
        xf = np.eye(4); xf[:3, :3] = scenes._rotation(rs) * (1e-3 * (1 + k % 3)); xf[:3, 3] = rs.uniform(-0.01, 0.01, size=3)
        nodes.append((k % 2, xf.astype(np.float32)))
    P, I, Nq = scenes._quad([[-5, 30, -5], [5, 30, -5], [5, 30, 5], [-5, 30, 5]], [0, -1, 0])
    lights = [(len(nodes), (10.0, 10.0, 10.0))]
    nodes.append((2, None)); meshes.append(dict(V=P, F=I, N=Nq, material=("diffuse", (0.5, 0.5, 0.5))))
    sc = scenes.build_scene(dict(meshes=meshes, nodes=nodes, lights=lights, camera=scenes._camera((0, 0.05, -0.2), (0, 0, 0), 40.0, 32, 32)))
    d = sc.desc
    fb = api.FlatBvh(d, api.FLAT_Q4)
    N = fb.nodes(); exist = (N[:, 3] >> 24) & 15
    assert (exist != 15).mean() > 0.2                                               # the tree has slots without a child
    targets = rs.uniform(-0.012, 0.012, size=(24000, 3))
    dirs = rs.normal(size=targets.shape); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dist = np.repeat([3.0, 3e2, 3e3, 3e4], len(targets) // 4)[:, None]
    rays = np.zeros((len(targets), 8), np.float32)
    rays[:, :3] = targets - dist * dirs; rays[:, 4:7] = dirs; rays[:, 3] = d.ray_trace_eps; rays[:, 7] = np.float32(3.4e38)
    scene = gpu.Scene(d, flatten=True)
    got = gpu.intersect(scene, rays)
    want = orc.intersect(d, rays, flat=fb.desc)
    near = slice(0, len(targets) // 4)
    assert (want["tri_idx"][near] >= 0).mean() > 0.2
    for k in ("tri_idx", "node_idx"):
        assert np.array_equal(got[k][near], want[k][near]), k
    for k in ("dist", "u", "v"):
        assert np.array_equal(got[k][near].view(np.uint32), want[k][near].view(np.uint32)), k
    assert ((got["tri_idx"] >= 0) == (want["tri_idx"] >= 0)).mean() > 0.97
    # every reported hit is the reference's evaluation of that triangle: the two-level oracle over an interval closed in around the reported t
    hitm = got["tri_idx"] >= 0
    probe = rays[hitm].copy()
    probe[:, 7] = np.nextafter(np.nextafter(got["dist"][hitm], np.float32(np.inf)), np.float32(np.inf))
    probe[:, 3] = np.nextafter(np.nextafter(got["dist"][hitm], np.float32(0)), np.float32(0))
    again = orc.intersect(d, probe)
    agree = (again["tri_idx"] == got["tri_idx"][hitm]) & (again["node_idx"] == got["node_idx"][hitm])
    assert agree.mean() > 0.8                                                       # (from far away many triangles share one representable t, and the probe's own box tests lose some)
    for k in ("dist", "u", "v"):
        assert np.array_equal(again[k][agree].view(np.uint32), got[k][hitm][agree].view(np.uint32)), k
    occ = gpu.intersect(scene, rays, any_hit=True)["tri_idx"] >= 0
    assert (occ == (orc.intersect(d, rays, any_hit=True, flat=fb.desc)["tri_idx"] >= 0)).mean() > 0.97


@pytest.mark.parametrize("flatten", [True, False])
def test_ray_claims_across_the_static_shares(gpu, flatten):
    """The waves' ray claims (traverse.h ray_claims, round 6): a wave's first claim is static, only the rays behind the static shares go through the launch's cursor.  Queue
    lengths on every side of that boundary — fewer rays than waves, exactly the static shares (one resident wave x 64 rays each), one ray more, several dynamic claims, a last
    partial wave — give, ray for ray, what the same rays give in launches of 4096 (where every ray lies in a static share and no atomic is issued at all): every ray is traced
    exactly once and lands in its own slot, closest hit and occlusion.  (The 4096-ray launches themselves are held to the oracle by the other tests of this file.)"""
    sc = scenes.synthetic_sm(64, 64, n_instances=40, subdiv=2)
    d = sc.desc
    scene = gpu.Scene(d, flatten=flatten)
    waves = 256 * 6 * 4          # resident traversal waves on an MI355X (kernels.hip traversal_blocks): 256 CUs x six workgroups (either layout) x four waves
    rays = camera_and_random_rays(d, waves * 64 * 3 + 77, 11)
    occ = camera_and_random_rays(d, waves * 64 * 3 + 77, 12, any_tmax=True)
    ref = {k: np.concatenate([gpu.intersect(scene, rays[i:i + 4096])[k] for i in range(0, len(rays), 4096)]) for k in ("dist", "u", "v", "tri_idx", "node_idx")}
    ref_occ = np.concatenate([gpu.intersect(scene, occ[i:i + 4096], any_hit=True)["tri_idx"] >= 0 for i in range(0, len(occ), 4096)])
    assert (ref["tri_idx"] >= 0).mean() > 0.3 and 0.1 < ref_occ.mean() < 0.95
    for n in (1, 63, 64, 65, 4097, waves * 64 - 1, waves * 64, waves * 64 + 1, waves * 64 + 64 * 37 + 5, len(rays)):
        got = gpu.intersect(scene, rays[:n])
        for k in ("tri_idx", "node_idx"):
            assert np.array_equal(got[k], ref[k][:n]), (n, k)
        for k in ("dist", "u", "v"):
            assert np.array_equal(got[k].view(np.uint32), ref[k][:n].view(np.uint32)), (n, k)
        assert np.array_equal(gpu.intersect(scene, occ[:n], any_hit=True)["tri_idx"] >= 0, ref_occ[:n]), (n, "occlusion")
