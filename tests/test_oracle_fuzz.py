"""Seeded random scenes (cudatracerlib_amd.scenes.fuzz_scene: every BSDF model, mirrored and sheared instances, every emitter kind) through the checker, on the CPU:
the host code builds each of them, the oracle renders finite non-negative radiance, the same seed is the same frame, and the flattened BVH of each scene reports the hits of
the reference's two-level traversal bit for bit (mirrored / sheared instance transforms go through the same inverse-transform rows either way).
The GPU side of the same scenes is tests/test_gpu_fuzz.py."""
import numpy as np
import pytest
from cudatracerlib_amd import api, scenes

SEEDS = list(range(12))


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_scene_renders_and_flattens(orc, seed):
    sc = scenes.fuzz_scene(seed)
    d = sc.desc
    tables = orc.sequence_tables(2)
    img, rays = orc.render(d, 48, 32, n_passes=2, tables=tables, max_path_length=6, rr_start=4)
    assert np.isfinite(img).all() and (img[..., :3] >= 0).all() and rays >= 48 * 32 * 2
    # every sample lands on the film, except the ones Image::AddSample drops: a NaN radiance (a one-sided rough coating seen from BEHIND evaluates outside its domain, as in the
    # reference — BSDF_Complex.cu:159-223 has no side check before the microfacet sample) stays a NaN through clampNegative and the sample is not counted (Image.cu:25-28)
    assert (img[..., 6] <= 2.0).all() and img[..., 6].sum() >= 0.97 * 2 * 48 * 32
    if seed < 3:
        sc2 = scenes.fuzz_scene(seed)          # (the descriptor points into the scene object: keep it alive)
        again, rays2 = orc.render(sc2.desc, 48, 32, n_passes=2, tables=tables, max_path_length=6, rr_start=4)
        assert rays2 == rays and np.array_equal(again, img)
    # flattened BVH == two-level traversal on random rays through the scene box
    rs = np.random.RandomState(seed)
    lo, hi = np.array(d.box_min[:]), np.array(d.box_max[:])
    n = 3000
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = rs.uniform(lo, hi, size=(n, 3)); dd = rs.normal(size=(n, 3)); r[:, 4:7] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    r[:, 3] = d.ray_trace_eps; r[:, 7] = np.float32(3.402823466e+38)
    fb = api.FlatBvh(d, api.FLAT_Q4)
    want = orc.intersect(d, r); got = orc.intersect(d, r, flat=fb.desc)
    ties = (got["tri_idx"] != want["tri_idx"]) & (got["dist"] == want["dist"])
    same = ~ties
    assert ties.sum() <= 3
    for k in ("tri_idx", "node_idx"):
        assert np.array_equal(got[k][same], want[k][same]), k
    for k in ("dist", "u", "v"):
        assert np.array_equal(got[k][same].view(np.uint32), want[k][same].view(np.uint32)), k


@pytest.mark.parametrize("env", [True, False])
def test_nan_samples_are_dropped_like_the_reference_drops_them(orc, env):
    """scenes.coating_from_behind: a primary hit on the panel evaluates a rough coating outside its domain: NaN from its sample when the specular lobe is chosen, as in the reference; the throughput is NaN,
    the continuation ray leaves the scene, and `cl += misWeight * cf * EvalEnvironment(r)` (PathTracer.cu:99-111) — executed with or without a map — makes the radiance NaN:
    Image::AddSample's clampNegative keeps the NaN (max(0, NaN) = (0 > NaN) ? 0 : NaN, Math/MathFunc.h:96) and the sample is not counted (Engine/Image.cu:25-28).
    Both lines were restated wrongly until round 5's parity fuzz (NaN clamped to 0 and counted; no term without a map)."""
    sc = scenes.coating_from_behind(32, 24, env=env)
    img, rays = orc.render(sc.desc, 32, 24, n_passes=3, tables=orc.sequence_tables(3), max_path_length=6)
    assert np.isfinite(img).all()
    centre = img[8:16, 12:20, 6]
    assert (centre <= 3).all() and 3 * centre.size - centre.sum() >= 5      # the panel: the samples that chose the specular lobe (NaN) are dropped, the nested lobe's are counted
    assert img[-3:, :, 6].mean() > 2.9 and img[-3:, :, :3].mean() > 0        # the floor below it: (nearly) every sample counted — a path that bounces into the panel's back is dropped too


@pytest.mark.parametrize("seed", [0, 3, 146])
def test_zero_stop_side_image(orc, seed):
    """orc.render(zero_stop=...): the samples the reference drops as NaN AFTER the path's throughput had become exactly zero, each as the radiance collected up to that vertex —
    what the product's kernels (which end such a path at once) count; tests/test_gpu_fuzz.py holds their frames to `frame + zero_stop`.  Here: the side image is empty for a
    scene whose materials are evaluated inside their domain, holds a few samples for the scenes with one-sided coatings seen from behind, never more than the frame dropped,
    and leaves the frame itself untouched."""
    W, H, P = 48, 32, 3
    sc = scenes.fuzz_scene(seed, W, H)
    tables = orc.sequence_tables(P)
    plain, rays = orc.render(sc.desc, W, H, n_passes=P, tables=tables, max_path_length=6, rr_start=4)
    for kw in ({}, dict(wavefront_rules=True), dict(partials=True)):
        zs = np.zeros((H, W, 7), np.float32)
        img, _ = orc.render(sc.desc, W, H, n_passes=P, tables=tables, max_path_length=6, rr_start=4, zero_stop=zs, **kw)
        if not kw: assert np.array_equal(img, plain)
        assert np.isfinite(zs).all() and (zs[..., :3] >= 0).all()
        dropped = P - img[..., 6]
        assert (zs[..., 6] <= dropped).all()
        assert (zs[..., 6].sum() > 0) == (seed != 0), (seed, kw, float(zs[..., 6].sum()))
