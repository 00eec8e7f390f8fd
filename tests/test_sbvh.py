"""Mesh BVH construction (row a19): sbvh_builder.cpp restates the reference's SplitBVHBuilder (object splits by full sweep, spatial
splits from 128 chopped bins, reference unsplitting; Engine/SpatialStructures/BVH/SplitBVHBuilder.cpp:219-640 via ConstructBVH,
Engine/MeshLoader/BVHBuilderHelper.cpp:116-127).  The bar is bit equality of the emitted arrays — BVHNodeData, Woop rows, index
words — with what the reference's own build returns: against committed golden fixtures (tests/golden/sbvh.npz, made by
tests/golden/generate.py from oracle/_ref) and, where oracle/_ref is present, against the reference live on more meshes."""
import ctypes as C
import os
import numpy as np
import pytest
import cudatracerlib_amd as ctl
from cudatracerlib_amd import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sbvh.npz")


def sbvh_cases():
    """seeded inputs shared with tests/golden/generate.py"""
    rs = np.random.RandomState(7)
    cases = {}
    cases["icosphere2"] = scenes.icosphere(2)
    n = 400                                                        # soup of long thin triangles: the case spatial splits exist for
    c = rs.uniform(-5, 5, (n, 1, 3)); d = rs.normal(size=(n, 1, 3)) * rs.uniform(0.1, 6, (n, 1, 1)); e = rs.normal(size=(n, 3, 3)) * 0.15
    T = (c + d * np.array([[-1], [0], [1]])[None] + e).astype(np.float32)
    cases["thin_soup"] = (T.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3))
    g = np.linspace(0, 1, 6, dtype=np.float32); X, Y = np.meshgrid(g, g)     # planar grid: zero extent on one axis, many equal centroids
    V = np.stack([X.ravel(), np.zeros(36, np.float32), Y.ravel()], 1)
    F = np.array([[i * 6 + j, i * 6 + j + 1, (i + 1) * 6 + j + 1] for i in range(5) for j in range(5)] +
                 [[i * 6 + j, (i + 1) * 6 + j + 1, (i + 1) * 6 + j] for i in range(5) for j in range(5)], np.uint32)
    cases["plane"] = (V, F)
    cases["one_tri"] = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32))
    cases["two_tris"] = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5], [6, 5, 5], [5, 6, 6]], np.float32), np.array([[0, 1, 2], [3, 4, 5]], np.uint32))
    big = rs.normal(size=(60, 3, 3)).astype(np.float32) * np.float32(4.0)      # large overlapping triangles: duplicates at several levels
    cases["overlap"] = (big.reshape(-1, 3), np.arange(180, dtype=np.uint32).reshape(-1, 3))
    return cases


def build(V, F, mode="sbvh"):
    sc = ctl.DynamicScene(); sc.set_bvh_mode(mode)
    sc.CreateNode(sc.add_mesh(V, F))
    sc.setCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 16, 16)
    d = sc.UpdateScene()
    return sc, (d.view("bvh_nodes", np.uint32, d.n_bvh_nodes, 16).copy(), d.view("woop", np.uint32, d.n_woop, 12).copy(), d.view("woop_index", np.uint32, d.n_woop, 1).copy().ravel())


def test_arrays_equal_the_reference_build_golden():
    g = np.load(GOLDEN)
    cases = sbvh_cases()
    splits = 0
    for name, (V, F) in cases.items():
        assert np.array_equal(g[name + "_V"], np.ascontiguousarray(V, np.float32)) and np.array_equal(g[name + "_F"], F), name   # the fixture's inputs are these inputs
        _, (nodes, woop, index) = build(V, F)
        assert np.array_equal(nodes, g[name + "_nodes"]), name
        assert np.array_equal(woop, g[name + "_woop"]), name
        assert np.array_equal(index, g[name + "_index"]), name
        splits += len(index) - len(F)
    assert splits > 50                                             # the fixtures do exercise spatial splits (duplicated references)


def _ref_build(r, V, F):
    V = np.ascontiguousarray(V, np.float32); F = np.ascontiguousarray(F, np.uint32)
    nn, nt = C.c_uint32(), C.c_uint32()
    r.ref_construct_bvh(V.ctypes.data, F.ctypes.data, len(V), F.size, C.byref(nn), C.byref(nt))
    nodes = np.zeros((nn.value, 16), np.uint32); tris = np.zeros((nt.value, 12), np.uint32); idx = np.zeros(nt.value, np.uint32)
    r.ref_construct_bvh_fetch(nodes.ctypes.data, tris.ctypes.data, idx.ctypes.data)
    return nodes[:max(1, nn.value - 2)], tris[:nt.value - 2], idx[:nt.value - 2]


def test_arrays_equal_the_reference_build_live(ref):
    rs = np.random.RandomState(11)
    meshes = [scenes.icosphere(4), scenes.unit_box()[:2]]
    V, F = scenes.icosphere(3)
    meshes.append((V * np.float32([8, 0.2, 1]), F))              # squashed sphere: slivers
    soup = (rs.normal(size=(1500, 3, 3)) * rs.uniform(0.05, 3, (1500, 1, 1)) + rs.uniform(-10, 10, (1500, 1, 3))).astype(np.float32)
    meshes.append((soup.reshape(-1, 3), np.arange(4500, dtype=np.uint32).reshape(-1, 3)))
    for V, F in meshes:
        want = _ref_build(ref, V, F)
        _, got = build(V, F)
        for a, b in zip(want, got):
            assert np.array_equal(a, b)


def test_binned_and_sbvh_trees_return_the_same_hits(orc):
    """the closest hit does not depend on the tree: same triangle and distance from both builders (duplicated references included)"""
    V, F = sbvh_cases()["thin_soup"]
    rs = np.random.RandomState(3)
    o = rs.uniform(-8, 8, (600, 3)).astype(np.float32); t = rs.uniform(-4, 4, (600, 3)).astype(np.float32)
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, np.full((600, 1), 1e-4, np.float32), d.astype(np.float32), np.full((600, 1), 1e30, np.float32)], axis=1)
    res = {}
    for mode in ("sbvh", "binned"):
        sc, arrays = build(V, F, mode)
        res[mode] = (orc.intersect(sc.desc, rays), len(arrays[2]))
    a, b = res["sbvh"][0], res["binned"][0]
    assert res["sbvh"][1] > len(F) and res["binned"][1] == len(F)
    assert np.array_equal(a["tri_idx"], b["tri_idx"]) and np.array_equal(a["dist"], b["dist"]) and (a["tri_idx"] >= 0).sum() > 100


def test_auto_mode_and_limits():
    V, F = scenes.icosphere(2)
    _, auto = build(V, F, "auto")
    _, sbvh = build(V, F, "sbvh")
    assert all(np.array_equal(x, y) for x, y in zip(auto, sbvh))  # small meshes get the reference's tree by default
    with pytest.raises(KeyError):
        build(V, F, "fastest")
