"""Block samplers of the oracle (oracle/block_sampler.py <- Kernel/BlockSampler/, Kernel/PixelVarianceBuffer.h): the scheme's closed forms."""
import numpy as np
from oracle import block_sampler as B


def _frame(h, w, seed, k):
    rs = np.random.RandomState(seed)
    f = np.zeros((h, w, 7), np.float32)
    f[..., 6] = k
    f[..., :3] = rs.rand(h, w, 3).astype(np.float32) * k
    return f


def test_uniform_and_select_and_user_weights():
    s = B.BlockSampler(B.UNIFORM, 200, 130)                        # 4 x 3 blocks
    assert (s.bx, s.by) == (4, 3) and np.all(s.counts() == 1)
    s.set_weight(1, 2, 0.0); s.set_weight(3, 0, 5.0)
    assert np.all(s.counts() == 1)                                 # m_nonZero is only noticed by the first AddPass
    s.add_pass(_frame(130, 200, 0, 1), 1.0, s.counts())
    c = s.counts()
    assert c[2, 1] == 0 and c.sum() == 11 and s.indices[0] == 3    # deselected block skipped, heaviest block first
    sel = B.BlockSampler(B.SELECT, 200, 130)
    assert sel.counts().sum() == 0
    sel.set_weight(2, 1, 1.0)
    c = sel.counts(); assert c.sum() == 1 and c[1, 2] == 1


def test_variance_sampler_switches_to_mixed_sampling_after_ten_passes():
    h, w = 128, 256                                                # 4 x 2 blocks
    s = B.BlockSampler(B.VARIANCE, w, h)
    acc = np.zeros((h, w, 7), np.float32)
    rs = np.random.RandomState(1)
    noise = np.ones((h, w), np.float32) * 0.01; noise[:64, 64:128] = 2.0      # block (1, 0) is the noisy one
    for k in range(12):
        c = s.counts()
        assert np.all(c == 1) if k < 10 else True
        if k >= 10:
            assert c[0, 1] >= 1                                    # the noisy block is among the weighted quarter (8 // 4 = 2 blocks)
            det = np.zeros(8, int); det[(k % 2)::2] = 1            # every second block in turn, passCounter = passes done so far
            assert np.all(c.ravel() - det >= 0) and (c.ravel() - det).sum() == 2 and c.max() <= 2
        per_px = np.repeat(np.repeat(c, 64, 0), 64, 1)
        add = (0.5 + noise * rs.randn(h, w)).astype(np.float32)
        acc[..., :3] += (add * per_px)[..., None]; acc[..., 6] += per_px
        s.add_pass(acc, 1.0 / (k + 1), c)
    assert s.indices[0] == 1 and s.keys[1] == s.keys.max()
    # moments: E and Var of the per-pass estimator luminance (VarAccumulator)
    n = s.n_var[0, 0]; assert n == 12 - (c[0, 0] == 0)
    e = s.sum_x[10, 70] / s.n_var[10, 70]
    assert abs(e - 0.5 * (0.212671 + 0.715160 + 0.072169)) < 1.5


def test_difference_sampler_needs_two_passes_and_ranks_by_half_buffer_error():
    h, w = 64, 192
    s = B.BlockSampler(B.DIFFERENCE, w, h)
    acc = np.zeros((h, w, 7), np.float32)
    rs = np.random.RandomState(2)
    for k in range(11):
        c = s.counts()
        amp = np.full((h, w), 0.01, np.float32); amp[:, 128:] = 1.0            # block 2 fluctuates
        acc[..., :3] += np.abs(1.0 + amp * rs.randn(h, w)).astype(np.float32)[..., None]; acc[..., 6] += 1
        s.add_pass(acc, 1.0 / (k + 1), c)
        if k == 0:
            assert s.passes_done == 1 and np.all(s.keys == 0)      # first pass: early return (DifferenceBlockSampler.cu:34-35)
    assert s.indices[0] == 2
    c = s.counts()                                                 # passes_done = 11 >= 10: mixed; 3 // 4 = 0 weighted blocks, every second block from 11 % 2
    assert list(c.ravel()) == [0, 1, 0]
