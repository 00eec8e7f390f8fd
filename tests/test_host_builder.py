"""Host side of the product (scene builder, BVH, sampler, C-ABI surface) — no GPU needed.
The oracle is the checker: builder output must carry exactly the reference's encodings and bit patterns."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cudatracerlib_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def brute_force(P, rays):
    """closest hit of rays against a triangle soup P (n,3,3), float64 Moller-Trumbore"""
    o, d = rays[:, :3].astype(np.float64), rays[:, 4:7].astype(np.float64)
    best = np.full(len(rays), np.inf)
    P = P.astype(np.float64)
    for tri in P:
        e1, e2 = tri[1] - tri[0], tri[2] - tri[0]
        pv = np.cross(d, e2); det = pv @ e1
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - tri[0]; u = (tv * pv).sum(1) * inv
            qv = np.cross(tv, e1); v = (d * qv).sum(1) * inv; t = (qv @ e2) * inv
        ok = (np.abs(det) > 1e-12) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > rays[:, 3]) & (t < rays[:, 7])
        best = np.where(ok & (t < best), t, best)
    return best


def test_library_exports_every_declared_symbol(ctl):
    hdr = open(os.path.join(ROOT, "include", "ctl_amd.h")).read()
    names = sorted(set(re.findall(r"\b(ctl_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 40
    for n in names:
        assert hasattr(ctl.lib, n), "libctl_amd.so does not export " + n


def test_no_cpu_fallback_without_device(ctl):
    if ctl.device_count() > 0:
        pytest.skip("a device is present")
    h = C.c_void_p()
    assert ctl.lib.ctl_tracer_create(b"WavefrontPathTracer", C.byref(h)) == -2   # CTL_ERR_NO_DEVICE
    assert b"no HIP device" in ctl.lib.ctl_last_error()
    assert ctl.lib.ctl_image_create(8, 8, C.byref(h)) == -2
    sc = scenes.cornell_box(16, 16)
    with pytest.raises(ctl.CtlError):
        ctl.Scene(sc.desc)
    assert ctl.lib.ctl_tracer_create(b"PhotonTracer", C.byref(h)) == -5      # CTL_ERR_UNSUPPORTED: not on the hot path


def test_woop_and_triangle_data_match_oracle(ctl, orc):
    rs = np.random.RandomState(3)
    V = (rs.normal(size=(300, 3)) * 4).astype(np.float32)
    F = rs.randint(0, 300, size=(200, 3)).astype(np.uint32)
    F = F[(F[:, 0] != F[:, 1]) & (F[:, 1] != F[:, 2]) & (F[:, 0] != F[:, 2])]
    N = rs.normal(size=(300, 3)).astype(np.float32); N /= np.linalg.norm(N, axis=1, keepdims=True)
    UV = rs.uniform(-1, 3, size=(300, 2)).astype(np.float32)
    sc = ctl.DynamicScene()
    m = sc.add_mesh(V, F, normals=N, uvs=UV, tri_material=(np.arange(len(F)) % 3).astype(np.uint8), materials=[ctl.diffuse(), ctl.diffuse(), ctl.diffuse()])
    sc.CreateNode(m); sc.setCamera((0, 0, -30), (0, 0, 0), (0, 1, 0), 45, 32, 32)
    d = sc.UpdateScene()
    woop = d.view("woop", np.float32, d.n_woop, 12); widx = d.view("woop_index", np.uint32, d.n_woop, 1)[:, 0]
    tri = d.view("tri_data", np.uint32, d.n_tri_data, 8)
    assert d.n_tri_data == len(F)
    want = np.zeros(12, np.float32)
    seen = np.zeros(len(F), bool)
    for k in range(d.n_woop):
        t = widx[k] >> 1
        seen[t] = True
        v = [np.ascontiguousarray(V[F[t, j]]) for j in range(3)]
        orc.lib.orc_woop_set_data(v[0].ctypes.data_as(C.c_void_p), v[1].ctypes.data_as(C.c_void_p), v[2].ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
        assert np.array_equal(bits(woop[k]), bits(want)), k
    assert seen.all()
    out = np.zeros(8, np.uint32)
    for t in range(len(F)):
        P = np.ascontiguousarray(V[F[t]].reshape(9)); Nn = np.ascontiguousarray(N[F[t]].reshape(9)); T = np.ascontiguousarray(UV[F[t]].reshape(6))
        orc.lib.orc_triangle_data_pack(P.ctypes.data, Nn.ctypes.data, T.ctypes.data, t % 3, 0, out.ctypes.data)
        assert np.array_equal(tri[t], out), t


def check_bvh_encoding(nodes, n_leaf_entries, last_flags, root=0):
    """reference encoding (SplitBVHBuilder.cpp:163-203): children = node*4 | ~leaf | 0x76543210; boxes contain children"""
    seen_leaf = np.zeros(n_leaf_entries, bool)
    stack = [root]
    count = 0
    while stack:
        a = stack.pop()
        assert a % 4 == 0 and 0 <= a // 4 < len(nodes)
        n = nodes[a // 4]
        count += 1
        for c in (n[12].view(np.int32), n[13].view(np.int32)):
            c = int(c)
            if c == 0x76543210:
                continue
            if c < 0:
                k = ~c
                while True:
                    assert not seen_leaf[k]; seen_leaf[k] = True
                    if last_flags[k]:
                        break
                    k += 1
            else:
                stack.append(c)
    assert seen_leaf.all()
    return count


def test_mesh_and_scene_bvh_are_valid_and_hits_match_brute_force(ctl, orc):
    sc = scenes.synthetic_sm(32, 32, n_instances=40, subdiv=1)
    d = sc.desc
    nodes = d.view("bvh_nodes", np.float32, d.n_bvh_nodes, 16); widx = d.view("woop_index", np.uint32, d.n_woop, 1)[:, 0]
    meshes = d.view("meshes", np.uint32, d.n_meshes, 5)
    offs_n = np.append(meshes[:, 1] // 4, d.n_bvh_nodes); offs_w = np.append(meshes[:, 3], d.n_woop)
    for m in range(d.n_meshes):
        sub = nodes[offs_n[m]:offs_n[m + 1]]
        check_bvh_encoding(sub, offs_w[m + 1] - offs_w[m], (widx[offs_w[m]:offs_w[m + 1]] & 1).astype(bool))
    # scene BVH: one leaf per node, ~nodeIdx (BVHRebuilder.cpp:380-383)
    top = d.view("scene_bvh_nodes", np.float32, d.n_scene_bvh_nodes, 16)
    check_bvh_encoding(top, d.n_nodes, np.ones(d.n_nodes, bool), d.scene_start_node)
    # hits through the two-level BVH == brute force over all instanced triangles
    woop = d.view("woop", np.float32, d.n_woop, 12)
    xf = d.view("node_transforms", np.float32, d.n_nodes, 16).reshape(-1, 4, 4); nd = d.view("nodes", np.uint32, d.n_nodes, 6)
    tris = []
    back = np.zeros((3, 3), np.float32)
    for k in range(d.n_nodes):
        mi = nd[k, 0]
        for w in range(offs_w[mi], offs_w[mi + 1]):
            orc.lib.orc_woop_get_data(np.ascontiguousarray(woop[w]).ctypes.data_as(C.c_void_p), back[0].ctypes.data_as(C.c_void_p), back[1].ctypes.data_as(C.c_void_p), back[2].ctypes.data_as(C.c_void_p))
            tris.append((xf[k][:3, :3].astype(np.float64) @ back.T.astype(np.float64)).T + xf[k][:3, 3])
    tris = np.array(tris)
    rs = np.random.RandomState(5)
    n = 600
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rs.uniform(-60, 60, size=(n, 3)); dd = rs.normal(size=(n, 3)); rays[:, 4:7] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    rays[:, 3] = d.ray_trace_eps; rays[:, 7] = 3.0e38
    got = orc.intersect(d, rays)
    want = brute_force(tris, rays)
    hit = got["tri_idx"] >= 0
    assert np.array_equal(hit, np.isfinite(want)) or (hit != np.isfinite(want)).sum() <= 2   # grazing edges
    both = hit & np.isfinite(want)
    assert np.allclose(got["dist"][both], want[both], rtol=2e-3, atol=2e-3)


def test_area_light_shape_set_and_light_cdf(ctl, orc):
    sc = scenes.cornell_box(32, 32)
    d = sc.desc
    assert d.num_lights == 1 and d.n_lights_buf == 1
    L = d.lights[0]
    assert L.type == 2 and L.count == 2 and abs(L.sum_area - 130.0 * 105.0) < 1.0
    anim = np.ctypeslib.as_array((C.c_ubyte * d.n_anim_bytes).from_address(d.anim))
    cdf = np.frombuffer(anim[L.area_dist_index:L.area_dist_index + 12].tobytes(), np.float32)
    assert cdf[0] == 0.0 and cdf[2] == 1.0 and 0.49 < cdf[1] < 0.51
    st = np.frombuffer(anim[L.triangles_index:L.triangles_index + 128].tobytes(), np.float32).reshape(2, 16)
    assert np.allclose(st[:, 9:12], [[0, -1, 0]] * 2, atol=2e-2)   # ShapeSet::triData::n = shading normal, faces down
    assert d.light_cdf[0] == 1.0
    mats = d.materials
    node = d.view("nodes", np.uint32, d.n_nodes, 6)[0]
    assert mats[node[1] + 3].node_light_index == 0 and node[3] == 0 and node[5] == 1
    assert abs(d.ray_trace_eps - 1e-4 * np.linalg.norm(np.array(d.box_max[:]) - np.array(d.box_min[:]))) < 1e-6


def test_sequence_generator_matches_oracle(ctl, orc):
    g = ctl.SequenceGenerator()
    want = orc.sequence_tables(2)
    for k in range(2):
        t1, t2 = g.compute()
        assert np.array_equal(bits(t1), bits(want[k][0])) and np.array_equal(bits(t2), bits(want[k][1]))


def test_parallel_table_generation_is_the_same_stream(ctl):
    """compute_many = skip-ahead by one pass's 368 640 draws per table set: identical bits to sequential generation, and the
    generator continues from the right state afterwards"""
    a, b = ctl.SequenceGenerator(), ctl.SequenceGenerator()
    seq = [a.compute() for _ in range(5)]
    m1, m2 = b.compute_many(3, threads=3)
    for k in range(3):
        assert np.array_equal(m1[k].view(np.uint32), seq[k][0].view(np.uint32)) and np.array_equal(m2[k].view(np.uint32), seq[k][1].view(np.uint32))
    n1, n2 = b.compute_many(2, threads=8)
    for k in range(2):
        assert np.array_equal(n1[k].view(np.uint32), seq[3 + k][0].view(np.uint32)) and np.array_equal(n2[k].view(np.uint32), seq[3 + k][1].view(np.uint32))


def test_builder_errors(ctl):
    sc = ctl.DynamicScene()
    with pytest.raises(ctl.CtlError):
        sc.UpdateScene()                                  # no nodes
    V = np.eye(3, dtype=np.float32)
    m = sc.add_mesh(V, np.array([[0, 1, 2]], np.uint32))
    with pytest.raises(ctl.CtlError):
        sc.add_mesh(V, np.array([[0, 1, 7]], np.uint32))  # index out of range
    with pytest.raises(ctl.CtlError):
        sc.CreateNode(5)
    n = sc.CreateNode(m)
    with pytest.raises(ctl.CtlError):
        sc.CreateLight(n, 3, (1, 1, 1))                   # "Could not find material name in mesh!"
    with pytest.raises(ctl.CtlError):
        sc.UpdateScene()                                  # no camera
    with pytest.raises(ctl.CtlError):
        sc.CreateNode(m, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0.5, 1]], np.float32))   # projective


def test_rejected_node_leaves_the_builder_intact(ctl, orc):
    """a CreateNode that fails on its transform must not leave a node behind (nodes / transforms stay in step)"""
    sc = ctl.DynamicScene()
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    m = sc.add_mesh(V, np.array([[0, 1, 2]], np.uint32))
    n0 = sc.CreateNode(m)
    for bad in (np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0.5, 1]], np.float32),     # projective
                np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1]], np.float32)):       # singular
        with pytest.raises(ctl.CtlError):
            sc.CreateNode(m, bad)
    T = np.eye(4, dtype=np.float32); T[0, 3] = 2.0
    n1 = sc.CreateNode(m, T)
    assert (n0, n1) == (0, 1)
    sc.setCamera((0.3, 0.3, 5), (0.3, 0.3, 0), (0, 1, 0), 40.0, 8, 8)
    d = sc.UpdateScene()
    assert d.n_nodes == 2
    xf = d.view("node_transforms", np.float32, 2, 16)
    assert xf[0, 3] == 0.0 and xf[1, 3] == 2.0          # node 1 carries ITS transform, not the rejected ones'
    rays = np.array([[2.25, 0.25, 5, 0, 0, 0, -1, 1e30], [0.25, 0.25, 5, 0, 0, 0, -1, 1e30]], np.float32)
    hits = orc.intersect(d, rays)
    assert list(hits["node_idx"]) == [1, 0]
