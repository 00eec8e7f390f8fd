"""Compiled-geometry cache (cudatracerlib_amd/csrc/scene_cache.h) — the role of the reference's .xmsh files
(Engine/Mesh.cpp:46-98 reads what Mesh::CompileMesh :199-290 wrote): a warm run must hand back exactly the arrays a cold run
computed, a changed input must miss, a damaged file must be ignored, and nothing is written without a directory."""
import os
import numpy as np
import pytest
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api, scenes


def _arrays(d):
    return (d.view("tri_data", np.uint32, d.n_tri_data, 8).copy(), d.view("woop", np.uint32, d.n_woop, 12).copy(),
            d.view("woop_index", np.uint32, d.n_woop, 1).copy(), d.view("bvh_nodes", np.uint32, d.n_bvh_nodes, 16).copy(),
            d.view("meshes", np.uint32, d.n_meshes, 5).copy(), np.array(d.box_min[:] + d.box_max[:], np.float32))


@pytest.fixture()
def cache_dir(tmp_path):
    d = str(tmp_path / "cache")
    api.set_cache_dir(d)
    yield d
    api.set_cache_dir(None)


def test_no_directory_no_files(tmp_path):
    api.set_cache_dir(None)
    sc = scenes.cornell_box(32, 32, glass_sphere=True)
    assert api.flatten_probe(sc.desc)["leaves"] > 0
    assert not os.path.exists(str(tmp_path / "cache"))


def test_warm_run_returns_the_cold_run(cache_dir):
    api.set_cache_dir(None)
    plain = scenes.cornell_box(32, 32, glass_sphere=True)
    want = _arrays(plain.desc); want_flat = api.flatten_probe(plain.desc)
    api.set_cache_dir(cache_dir)
    cold = scenes.cornell_box(32, 32, glass_sphere=True)
    cold_flat = api.flatten_probe(cold.desc)
    files = sorted(os.listdir(cache_dir))
    assert sum(f.startswith("mesh_") for f in files) == 2 and sum(f.startswith("flat_") for f in files) == 1
    assert not any(".tmp" in f for f in files)
    stamp = {f: os.path.getmtime(os.path.join(cache_dir, f)) for f in files}
    warm = scenes.cornell_box(32, 32, glass_sphere=True)
    warm_flat = api.flatten_probe(warm.desc)
    assert sorted(os.listdir(cache_dir)) == files and all(os.path.getmtime(os.path.join(cache_dir, f)) == stamp[f] for f in files)   # nothing rewritten
    for a, b, c in zip(want, _arrays(cold.desc), _arrays(warm.desc)):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert want_flat == cold_flat == warm_flat
    assert api.flatten_probe(warm.desc, api.FLAT_F4) != warm_flat  # the node format is part of the key
    assert len(os.listdir(cache_dir)) == len(files) + 1


def test_changed_input_misses_and_damaged_file_is_ignored(cache_dir):
    V, F = scenes.icosphere(2)
    def build(scale):
        sc = ctl.DynamicScene()
        sc.CreateNode(sc.add_mesh(V * scale, F, normals=V))
        sc.setCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 16, 16)
        return sc, sc.UpdateScene()
    sc1, d1 = build(1.0)
    assert len(os.listdir(cache_dir)) == 1
    sc2, d2 = build(1.5)
    files = sorted(os.listdir(cache_dir))
    assert len(files) == 2                                         # other vertex data -> other entry
    ref = _arrays(d1)
    for f in files:                                                # truncate both entries
        p = os.path.join(cache_dir, f)
        data = open(p, "rb").read()
        open(p, "wb").write(data[:len(data) // 2])
    sc3, d3 = build(1.0)
    for a, b in zip(ref, _arrays(d3)):
        assert np.array_equal(a, b)
    for f in files:                                                # one flipped payload byte: sizes are right, the checksum is not
        p = os.path.join(cache_dir, f)
        data = bytearray(open(p, "rb").read())
        data[len(data) // 2] ^= 0x40
        open(p, "wb").write(bytes(data))
    sc5, d5 = build(1.0)
    for a, b in zip(ref, _arrays(d5)):
        assert np.array_equal(a, b)
    assert not [f for f in os.listdir(cache_dir) if ".tmp" in f]   # writers leave no temporaries behind
    open(os.path.join(cache_dir, files[0]), "wb").write(b"not a cache file")
    sc4, d4 = build(1.0)
    for a, b in zip(ref, _arrays(d4)):
        assert np.array_equal(a, b)


def test_flattened_bvh_is_well_formed(cache_dir):
    """every instanced triangle appears among the leaf entries — once, or, where early split clipping (flatten.cpp) entered a large triangle as several references, once per reference
    (the axis-aligned walls of the box are not clipped: halving them removes no empty space; tests/test_oracle_flat.py has a scene with beams)"""
    sc = scenes.cornell_box(32, 32, glass_sphere=True)
    d = sc.desc
    r = api.flatten_probe(d)
    nodes = d.view("nodes", np.uint32, d.n_nodes, 6)
    meshes = d.view("meshes", np.uint32, d.n_meshes, 5)
    tri_off = np.sort(meshes[:, 0]); counts = np.diff(np.append(tri_off, d.n_tri_data))
    per_mesh = dict(zip(tri_off.tolist(), counts.tolist()))
    want = sum(per_mesh[int(meshes[int(n[0]), 0])] for n in nodes)
    assert r["leaves"] >= want and r["nodes"] >= 1 and 1 <= r["depth"] <= 31
    fb = api.FlatBvh(d)
    L = fb.leaves()
    uniq = np.unique(np.stack([L[:, 12] >> 1, L[:, 13]], axis=1), axis=0)     # words 12, 13 of an entry: {triangle << 1 | last, node}
    assert len(uniq) == want                                     # every (triangle, instance) pair is there
