"""Parity fuzz: seeded RANDOM scenes (cudatracerlib_amd.scenes.fuzz_scene — materials drawn from all fourteen BSDF models with random parameters and textures, nested models
under coatings and blends, normal / height maps, mirrored and sheared instances, one to three emitters of the five kinds) rendered by the HIP wavefront path tracer and by
the oracle's PathTrace<DIRECT> on the same sampler tables, at the bar of tests/test_gpu_render.py (99.95 % of pixels within 2e-3 (1 + ref), image means to 1e-3, weights
equal) — two-level and flattened BVH.  The hand-made scenes of the other tests each exercise what they were written for; these exercise combinations nobody chose."""
import numpy as np
import pytest
from cudatracerlib_amd import scenes

pytestmark = pytest.mark.gpu

W, H, PASSES, DEPTH, RR = 48, 32, 3, 6, 4


def _render(gpu, scene, tables):
    tr = gpu.WavefrontPathTracer()
    p = tr.getParameters(); p.setValue("MaxPathLength", DEPTH); p.setValue("RRStartDepth", RR)
    tr.Resize(W, H); tr.InitializeScene(scene)
    img = gpu.Image(W, H)
    for k in range(PASSES):
        tr.setSamplerTables(*tables[k])
        tr.DoPass(img, new_trace=(k == 0))
    return img.getPixelData()


@pytest.mark.parametrize("seed", list(range(12)))
def test_fuzz_scene_gpu_equals_oracle(gpu, orc, seed):
    sc = scenes.fuzz_scene(seed, W, H)
    d = sc.desc
    tables = orc.sequence_tables(PASSES)
    want, _ = orc.render(d, W, H, n_passes=PASSES, tables=tables, max_path_length=DEPTH, rr_start=RR)
    for flatten in (False, True):
        got = _render(gpu, gpu.Scene(d, flatten=flatten), tables)
        assert np.array_equal(got[..., 6], want[..., 6]), ("weightSum", seed, flatten)
        g, w = got[..., :3], want[..., :3]
        assert np.isfinite(g).all()
        ok = (np.abs(g - w) <= 2e-3 * (1 + np.abs(w))).all(axis=2)
        # 1536 pixels: 99.95 % allows none; the bar is "at most one pixel off" so that a single tie between two equidistant surfaces (BVHs differ) does not fail a seed
        assert (~ok).sum() <= 1, (seed, flatten, int((~ok).sum()), np.argwhere(~ok)[:4].tolist(), g[~ok][:2].tolist(), w[~ok][:2].tolist())
        assert abs(g.mean() - w.mean()) <= 1e-3 * max(w.mean(), 1e-6), (seed, flatten, float(g.mean()), float(w.mean()))
        # bit-equal pixels: the bar of tests/test_gpu_render.py where the device runs the checker's arithmetic; rough plastic / rough coating look their transmittance up in
        # the per-material 1-D reduction of the table (equal up to fp32 rounding), so scenes that hold one are held to the tolerance only
        if not any(d.materials[i].bsdf_type in (9, 14) for i in range(d.n_materials)):
            assert (g == w).all(axis=2).mean() >= 0.9, (seed, flatten, float((g == w).all(axis=2).mean()))
