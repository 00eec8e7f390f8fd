"""Block samplers behind the wavefront plugin (csrc/block_sampler.{h,hip}, parameter BlockSamplerType) against the numpy restatement
(oracle/block_sampler.py) and the oracle's renderer: before every pass the device-side sampler must ask for the samples per block that the
restatement derives from the same frames, and the pass must add to the frame what PathTrace<DIRECT> renders with those counts."""
import numpy as np
import pytest
from cudatracerlib_amd import scenes
from oracle import block_sampler as B

pytestmark = pytest.mark.gpu
W, H = 256, 192                                                    # 4 x 3 blocks of 64 x 64


def _close(got, want):
    assert np.array_equal(got[..., 6], want[..., 6])
    g, w = got[..., :3], want[..., :3]
    assert (np.abs(g - w) <= 2e-3 * (1 + np.abs(w))).all(axis=2).mean() >= 0.995
    assert abs(g.mean() - w.mean()) <= 1e-3 * max(w.mean(), 1e-6)


@pytest.mark.parametrize("kind", [B.VARIANCE, B.DIFFERENCE])
def test_adaptive_samplers_follow_the_restatement(gpu, orc, kind):
    sc = scenes.cornell_box(W, H, glass_sphere=True)
    d = sc.desc
    scene = gpu.Scene(d)
    tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 4); p.setValue("BlockSamplerType", kind)
    tr.Resize(W, H); tr.InitializeScene(scene)
    img = gpu.Image(W, H)
    ref = B.BlockSampler(kind, W, H)
    n_passes = 13
    tables = orc.sequence_tables(n_passes)
    prev = np.zeros((H, W, 7), np.float32)
    seen = set()
    for k in range(n_passes):
        want_counts = ref.counts()
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
        got_counts = tr.getBlockCounts(W, H)
        if not np.array_equal(got_counts, want_counts):            # only a tie at the boundary of the weighted quarter may differ
            diff = np.flatnonzero(got_counts.ravel() != want_counts.ravel())
            keys = ref.keys[diff]
            assert len(diff) == 2 and abs(keys[0] - keys[1]) <= 1e-4 * max(abs(keys).max(), 1e-6), (k, got_counts, want_counts, ref.keys)
        seen.update(np.unique(got_counts).tolist())
        frame = img.getPixelData()
        if k in (0, 10, 12):                                       # a uniform pass and two mixed passes against the oracle's renderer
            want, _ = orc.render(d, W, H, n_passes=1, tables=[tables[k]], max_path_length=4, block_counts=got_counts)
            _close(frame - prev, want)
        ref.add_pass(frame, 1.0 / (k + 1), got_counts)
        prev = frame
    assert seen == {0, 1, 2}                                       # skipped, sampled once and sampled twice all occurred
    assert tr.getNumPassesDone() == n_passes


def test_user_weights_select_and_limits(gpu, orc):
    sc = scenes.cornell_box(W, H)
    scene = gpu.Scene(sc.desc)
    tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 3); p.setValue("BlockSamplerType", B.SELECT)
    tr.Resize(W, H); tr.InitializeScene(scene)
    tr.setBlockWeight(1, 1, 1.0); tr.setBlockWeight(3, 2, 2.0)
    img = gpu.Image(W, H)
    tr.DoPasses(img, 3, new_trace=True)
    f = img.getPixelData()
    inside = np.zeros((H, W), bool); inside[64:128, 64:128] = True; inside[128:192, 192:256] = True
    assert np.all(f[..., 6][inside] >= 2) and np.all(f[..., 6][~inside] <= 1) and f[..., 6][~inside].sum() < 0.02 * inside.sum()   # a few samples jitter across a block edge
    assert np.array_equal(tr.getBlockCounts(W, H), np.array([[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.uint8))
    # uniform sampler with a deselected block
    tr2 = gpu.WavefrontPathTracer(); tr2.getParameters().setValue("MaxPathLength", 3)
    tr2.Resize(W, H); tr2.InitializeScene(scene)
    tr2.setBlockWeight(0, 0, 0.0)
    img2 = gpu.Image(W, H)
    tr2.DoPasses(img2, 3, new_trace=True)
    w2 = img2.getPixelData()[..., 6]
    # block (0, 0): sampled in the first pass only (the weights are noticed by the first AddPass); (y + jitter) rounding up moves a few samples to the next row
    assert (w2[8:56, 8:56] == 1).mean() > 0.995 and (w2[72:, 72:] == 3).mean() > 0.995 and w2[72:, 72:].mean() == pytest.approx(3, abs=0.01)
    # the plain uniform sampler stays on the batched fast path and reports ones
    tr3 = gpu.WavefrontPathTracer(); tr3.Resize(W, H)
    assert np.all(tr3.getBlockCounts(W, H) == 1)
    with pytest.raises(gpu.CtlError):
        tr3.setBlockWeight(9, 0, 1.0)
    # megakernel plugin: block samplers are the wavefront plugin's
    flat = gpu.Scene(sc.desc, flatten=True)
    pt = gpu.PathTracer(); pt.getParameters().setValue("BlockSamplerType", B.SELECT)
    pt.Resize(W, H); pt.InitializeScene(flat); pt.setBlockWeight(0, 0, 1.0)
    with pytest.raises(gpu.CtlError) as e:
        pt.DoPass(gpu.Image(W, H), new_trace=True)
    assert e.value.code == -5
