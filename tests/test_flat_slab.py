"""The oriented slabs of the flattened BVH's bottom nodes (cudatracerlib_amd/csrc/flat_slab.h, built in flatten.cpp) — checked without a GPU:

 * every leaf child's decoded interval  [base + step * lo, base + step * hi]  along its node's direction n contains all vertices of its triangles, with
   room to spare (the static pad the builder adds for the round-off of the object-space test), and inner children span the whole code range;
 * the whole chain: the oracle's traversal of the product's arrays evaluates the slabs with the kernel's fp32 expressions (oracle/ocore.h traceRayFlat)
   and still reports the two-level traversal's (t, u, v, triangle, node) bit for bit — also on a scene built to break a badly padded culling structure
   (meshes far from their own origin, instances scaled by 1e-3 .. 1e3, ray origins far outside the scene) — while fetching clearly fewer leaf entries."""
import numpy as np
import pytest

from cudatracerlib_amd import api, scenes
from test_oracle_flat import rays_for

N_MAX = 31
RAY_PAD = np.float32(31.0 * 1.9073486328125e-6)


def woop_vertices(L):
    """object-space vertices of the leaf entries' Woop rows (TriIntersectorData::getData, Engine/TriIntersectorData.cu:20-32), float64"""
    w = L[:, :12].view(np.float32).astype(np.float64)
    a, b, c = w[:, 0:4], w[:, 4:8], w[:, 8:12]
    m = np.zeros((len(L), 4, 4))
    m[:, 0] = b; m[:, 1] = c; m[:, 2] = a; m[:, 2, 3] = -a[:, 3]; m[:, 3, 3] = 1.0
    inv = np.linalg.inv(m)
    v2 = inv[:, :3, 3]
    return np.stack([v2 + inv[:, :3, 0], v2 + inv[:, :3, 1], v2], 1)


def world_vertices(desc, L):
    xf = desc.view("node_transforms", np.float32, desc.n_nodes, 16).astype(np.float64).reshape(-1, 4, 4)[L[:, 13]]
    v = woop_vertices(L)
    return np.einsum("nij,nvj->nvi", xf[:, :3, :3], v) + xf[:, None, :3, 3]


def decode_slabs(N):
    nw = N[:, 12]
    s6 = lambda v: ((v & 63).astype(np.int64) ^ 32) - 32
    n = np.stack([s6(nw), s6(nw >> 6), s6(nw >> 12)], 1)
    step = (nw & np.uint32(0xfffc0000)).view(np.float32)
    base = N[:, 13].view(np.float32)
    lo = np.stack([(N[:, 14] >> (8 * c)) & 255 for c in range(4)], 1)
    hi = np.stack([(N[:, 15] >> (8 * c)) & 255 for c in range(4)], 1)
    return n, step, base, lo, hi


def subtree_entries(links, L, link):
    """leaf entries under a child link (explicit links: node index * 4, ~first entry)"""
    out = []; stack = [int(link)]
    while stack:
        k = stack.pop()
        if k >= 0:
            stack.extend(int(x) for x in links[k // 4] if x != 0x76543210)
        else:
            e = ~k
            while True:
                out.append(e)
                if L[e, 12] & 1:
                    break
                e += 1
    return np.array(out, np.int64)


def slab_nodes(fb):
    """indices of the nodes that carry a slab: the children whose parent link says so (+ the root)"""
    N = fb.nodes(); links = fb.child_links(); imp = api.FlatBvh.implied_links(N)
    exist = (N[:, 3] >> 24) & 15; leafm = (N[:, 3] >> 28) & exist
    has = np.zeros(len(N), bool)
    for c in range(4):
        sel = (((exist & ~leafm) >> c) & 1 == 1) & ((imp[:, c] & 1) == 1)      # inner children whose link carries the flag
        has[links[sel, c] // 4] = True
    has[0] = bool(fb.desc.root_slab)
    return has


def ill_conditioned_scene():
    """an icosphere whose vertices sit 5000 units from the mesh's own origin, instanced with scales 1e-3 .. 30 and moved back to the room; plus a box scaled 400x"""
    V, F = scenes.icosphere(2)
    far = (V + np.array([5000.0, -3000.0, 4000.0])).astype(np.float32)
    Pb, Ib, Nb = scenes.unit_box()
    meshes = [dict(V=far, F=F, N=None, material=("diffuse", (0.7, 0.7, 0.7))), dict(V=Pb, F=Ib, N=Nb, material=("diffuse", (0.2, 0.5, 0.7)))]
    rs = np.random.RandomState(3)
    nodes = []
    for s in (1e-3, 0.02, 1.0, 7.0, 30.0):
        xf = np.eye(4); R = scenes._rotation(rs); xf[:3, :3] = R * s
        xf[:3, 3] = rs.uniform(-20, 20, size=3) - xf[:3, :3] @ np.array([5000.0, -3000.0, 4000.0])
        nodes.append((0, xf.astype(np.float32)))
    xf = np.eye(4); xf[:3, :3] = scenes._rotation(rs) * 400.0; nodes.append((1, xf.astype(np.float32)))
    xf = np.eye(4); xf[:3, :3] = scenes._rotation(rs) * 1e-2; xf[:3, 3] = (3, 4, 5); nodes.append((1, xf.astype(np.float32)))
    P, I, Nq = scenes._quad([[-5, 30, -5], [5, 30, -5], [5, 30, 5], [-5, 30, 5]], [0, -1, 0])
    lights = [(len(nodes), (10.0, 10.0, 10.0))]
    nodes.append((2, None)); meshes.append(dict(V=P, F=I, N=Nq, material=("diffuse", (0.5, 0.5, 0.5))))
    return scenes.build_scene(dict(meshes=meshes, nodes=nodes, lights=lights, camera=scenes._camera((0, 5, -68.0), (0, 0, 0), 60.0, 32, 32)))


SCENES = {
    "sm": lambda: scenes.synthetic_sm(32, 32, n_instances=60, subdiv=2),
    "cornell": lambda: scenes.cornell_box(32, 32, glass_sphere=True),
    "ill": ill_conditioned_scene,
}


@pytest.mark.parametrize("name", list(SCENES))
def test_every_leaf_child_lies_inside_its_decoded_slab(name):
    sc = SCENES[name](); d = sc.desc
    fb = api.FlatBvh(d, api.FLAT_Q4)
    N, L, links = fb.nodes(), fb.leaves(), fb.child_links()
    has = slab_nodes(fb)
    assert has.sum() == fb.desc.n_slab_nodes and has.sum() > 0.5 * ((N[:, 3] >> 28) != 0).sum()        # most nodes with leaf children get one
    n, step, base, lo, hi = decode_slabs(N)
    assert (np.abs(n[has]).max(1) == N_MAX).all() and (step[has] > 0).all()
    W = world_vertices(d, L)
    origin = N[:, :3].view(np.float32).astype(np.float64)
    meta = N[:, 3]; exist = (meta >> 24) & 15; leafm = (meta >> 28) & exist      # (an empty slot may carry a leaf bit: flatten.cpp)
    cnt = np.zeros((len(N), 4), np.int64)                                                               # entries of every leaf child: up to the entry flagged last
    for c in range(4):
        idx = np.nonzero(((leafm >> c) & 1) == 1)[0]; e = ~links[idx, c]; k = np.ones(len(idx), np.int64)
        for j in range(3):
            more = (L[np.minimum(e + k - 1, len(L) - 1), 12] & 1) == 0; k = k + (more & (k == j + 1))
        cnt[idx, c] = k
    checked = 0; tight_inner = 0
    for c in range(4):
        is_leaf = has & (((leafm >> c) & 1) == 1)
        is_inner = has & (((exist >> c) & 1) == 1) & ~is_leaf
        # an inner child: the whole node, or — when at most kSlabSubtree triangles hang under it — an interval that holds every one of them (checked below like a leaf child's)
        whole = (lo[:, c] == 0) & (hi[:, c] == 255)
        for i in np.nonzero(is_inner & ~whole)[0]:
            ents = subtree_entries(links, L, links[i, c]); assert 0 < len(ents) <= 64
            D = np.einsum("i,nvi->nv", n[i].astype(np.float64), W[ents] - origin[i])
            d0 = float(np.float32(base[i] + step[i] * np.float32(lo[i, c]))); d1 = float(np.float32(base[i] + step[i] * np.float32(hi[i, c])))
            ext = np.ldexp(255.0, ((int(meta[i]) >> (8 * np.arange(3))) & 255).astype(np.int64) - 127)
            room = np.abs(n[i]).sum() * 2.0 ** -20 * (np.abs(origin[i]) + ext).max()
            assert D.min() - d0 >= room and d1 - D.max() >= room
            tight_inner += 1
        gone = has & (((exist >> c) & 1) == 0)
        assert (lo[gone, c] == 255).all() and (hi[gone, c] == 0).all()
        idx = np.nonzero(is_leaf)[0]
        first = ~links[idx, c]
        for j in range(4):
            sel = cnt[idx, c] > j
            k = idx[sel]; e = first[sel] + j
            D = np.einsum("ni,nvi->nv", n[k].astype(np.float64), W[e] - origin[k, None, :])                # D(x) = n . (x - origin), exact enough in float64
            # the interval as the kernel's fp32 arithmetic sees it: base + step * code
            d0 = (base[k] + step[k] * lo[k, c].astype(np.float32)).astype(np.float64); d1 = (base[k] + step[k] * hi[k, c].astype(np.float32)).astype(np.float64)
            # room the builder must leave: 2^-20 of |n|_1 x the node's world magnitude (the round-off reach of the object-space test, flatten.cpp)
            ext = np.ldexp(255.0, ((meta[k, None] >> (8 * np.arange(3))) & 255).astype(np.int64) - 127)
            mag = (np.abs(origin[k]) + ext).max(1)
            room = np.abs(n[k]).sum(1) * 2.0 ** -20 * mag
            assert (D.min(1) - d0 >= room).all() and (d1 - D.max(1) >= room).all()
            checked += len(k)
    assert checked == L[np.isin(np.arange(len(L)), [])].shape[0] + checked and checked > 0.9 * len(L) * (has & (leafm != 0)).sum() / max(1, ((leafm != 0)).sum())
    assert tight_inner > 0.1 * ((leafm != 0) & has).sum()                                               # bottom nodes do get intervals in their parents


@pytest.mark.parametrize("name", list(SCENES))
def test_slab_traversal_reports_the_two_level_hits_bit_for_bit(orc, name):
    sc = SCENES[name](); d = sc.desc
    rays = rays_for(d, 20000, 11)
    rs = np.random.RandomState(4)
    far = rays[:4000].copy()                                                       # origins far outside the scene, aimed back at it: the ray-dependent pad
    lo, hi = np.array(d.box_min[:]), np.array(d.box_max[:]); ext = float((hi - lo).max())
    tgt = rs.uniform(lo, hi, size=(4000, 3))
    far[:, :3] = (tgt + rs.normal(size=(4000, 3)) * ext * rs.choice([3.0, 100.0, 1e4], size=(4000, 1))).astype(np.float32)
    dirs = tgt - far[:, :3]; far[:, 4:7] = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([rays, far])
    fb = api.FlatBvh(d, api.FLAT_Q4)
    want, _ = orc.intersect(d, rays, count=True)
    got, cf = orc.intersect(d, rays, count=True, flat=fb.desc)
    ties = (got["tri_idx"] != want["tri_idx"]) & (got["dist"] == want["dist"])
    same = ~ties
    assert ties.sum() <= len(rays) // 500 and (want["tri_idx"] >= 0).mean() > 0.15
    for k in ("tri_idx", "node_idx"):
        assert np.array_equal(got[k][same], want[k][same]), k
    for k in ("dist", "u", "v"):
        assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k
    assert np.array_equal(orc.intersect(d, rays, any_hit=True, flat=fb.desc)["tri_idx"] >= 0, orc.intersect(d, rays, any_hit=True)["tri_idx"] >= 0)


def test_slabs_cut_the_leaf_entry_fetches(orc):
    """the point of the slabs: the rays of a render (bounce and shadow rays START on a surface, inside the boxes of the neighbouring triangles) fetch clearly fewer
    leaf entries than with the boxes alone, visit a few per cent fewer nodes and render the same image — counted by zeroing the slab flags in a copy of the arrays
    (the builder's own switch is a measurement knob)"""
    sc = scenes.synthetic_sm(48, 48, n_instances=400, subdiv=3); d = sc.desc      # the denser the scene the more there is to cull: 2000 instances of the bench scene -25 %, 400 here -12 %
    fb = api.FlatBvh(d, api.FLAT_Q4)
    tables = orc.sequence_tables(1)
    c1, c0 = {}, {}
    img1, rays1 = orc.render(d, 48, 48, n_passes=1, tables=tables, max_path_length=6, flat=fb.desc, counts=c1)
    N = api.FlatBvh.clear_slab_flags(fb.nodes().copy())                          # no inner link carries the flag any more
    plain = api.FlatBvhDesc.from_buffer_copy(fb.desc)
    plain.nodes = N.ctypes.data; plain.root_slab = 0
    img0, rays0 = orc.render(d, 48, 48, n_passes=1, tables=tables, max_path_length=6, flat=plain, counts=c0)
    assert rays1 == rays0 and np.array_equal(img1, img0)
    assert 0.90 * c0["path_inner"] < c1["path_inner"] < 0.995 * c0["path_inner"], (c1, c0)      # fewer node visits too: the interval of an inner child with a small subtree keeps rays out of bottom nodes
    assert c1["path_tri"] < 0.92 * c0["path_tri"] and c1["occ_tri"] < 0.94 * c0["occ_tri"], (c1, c0)
