"""ORACLE (test infrastructure, never imported by the product): numpy restatement of the reference's block samplers and pixel-variance
buffer — Kernel/PixelVarianceBuffer.{h,cu}, Kernel/BlockSampler/{IBlockSampler,UniformBlockSampler,VarianceBlockSampler,
DifferenceBlockSampler,SelectBlockSampler}.h/.cu, driven as Tracer<true>::DoPass does (Kernel/Tracer.h:209-248): `counts()` before a
pass, `add_pass(frame, splat_scale, counts)` after it.  Blocks are 64 x 64 pixels (BLOCK_SAMPLER_BlockSize with BLOCK_FACTOR 2),
flattened row-major.  Parity unpinned (these files contain kernels); the tests check the scheme's closed forms besides the GPU comparison.
"""
import numpy as np

F = np.float32
BLOCK = 64
UNIFORM, VARIANCE, DIFFERENCE, SELECT = 0, 1, 2, 3


class BlockSampler:
    def __init__(self, kind, width, height, fraction_deterministic=2, fraction_weighted=4):
        self.kind, self.w, self.h = kind, width, height
        self.bx, self.by = (width + BLOCK - 1) // BLOCK, (height + BLOCK - 1) // BLOCK
        self.n = self.bx * self.by
        self.fd, self.fw = fraction_deterministic, fraction_weighted
        self.user = np.full(self.n, 0.0 if kind == SELECT else 1.0, F)
        self.indices = list(range(self.n))
        self.non_zero = False
        self.start_new_rendering()

    def start_new_rendering(self):
        self.passes_done = 0
        z3 = lambda: np.zeros((self.h, self.w, 3), F)
        self.prev_I, self.half = z3(), z3()
        self.iterations = np.zeros((self.h, self.w), np.int64); self.weight = np.zeros((self.h, self.w), F)
        self.sum_x = np.zeros((self.h, self.w), F); self.sum_x2 = np.zeros((self.h, self.w), F); self.n_var = np.zeros((self.h, self.w), np.int64)
        self.keys = np.zeros(self.n, F)

    def set_weight(self, block_x, block_y, w):
        self.user[block_y * self.bx + block_x] = w

    # ---- IterateBlocks -> BlockSamplerBuffer::Update: samples per block of the next pass
    def counts(self):
        c = np.zeros(self.n, np.uint8)
        if self.kind == UNIFORM:
            if self.non_zero:
                for b in self.indices:
                    if self.user[b] <= 0:
                        break
                    c[b] += 1
            else:
                c[:] = 1
        elif self.kind in (VARIANCE, DIFFERENCE):
            if self.passes_done < 10:
                c[:] = 1
            else:                                                   # MixedBlockIterate (IBlockSampler.h:131-153)
                for i in range(self.n // self.fw):
                    c[self.indices[i]] += 1
                for i in range(self.passes_done % self.fd, self.n, self.fd):
                    c[i] += 1
        else:
            c[self.user != 0] = 1
        return c.reshape(self.by, self.bx)

    def _per_pixel(self, per_block):
        return np.repeat(np.repeat(np.asarray(per_block).reshape(self.by, self.bx), BLOCK, axis=0), BLOCK, axis=1)[:self.h, :self.w]

    def _block_sum(self, a, mask=None):
        a = np.where(mask, a, 0) if mask is not None else a
        pad = np.zeros((self.by * BLOCK, self.bx * BLOCK), np.float64); pad[:self.h, :self.w] = a
        return pad.reshape(self.by, BLOCK, self.bx, BLOCK).sum(axis=(1, 3)).ravel()

    # ---- PixelVarianceBuffer::AddPass + <sampler>::AddPass
    def add_pass(self, frame, splat_scale, counts):
        c = self._per_pixel(counts).astype(F)
        on = c > 0
        new = (frame[..., 0:3] + frame[..., 3:6] * F(splat_scale)).astype(F)
        with np.errstate(divide="ignore", invalid="ignore"):
            est = ((new - self.prev_I) / c[..., None]).astype(F)
        self.prev_I = np.where(on[..., None], new, self.prev_I)
        self.weight = np.where(on, frame[..., 6], self.weight)
        odd = on & (self.iterations % 2 == 1)
        self.half = np.where(odd[..., None], (self.half + est).astype(F), self.half)
        self.iterations = self.iterations + on
        lum = (est[..., 0] * F(0.212671) + est[..., 1] * F(0.715160) + est[..., 2] * F(0.072169)).astype(F)
        self.sum_x = np.where(on, (self.sum_x + lum).astype(F), self.sum_x)
        self.sum_x2 = np.where(on, (self.sum_x2 + lum * lum).astype(F), self.sum_x2)
        self.n_var = self.n_var + on
        sq_user = (self.user * self.user).astype(F)
        if self.kind == VARIANCE:
            self.passes_done += 1
            with np.errstate(divide="ignore", invalid="ignore"):
                N = self.n_var.astype(F); inv = F(1) / N
                var = ((self.sum_x2 - self.sum_x * self.sum_x * inv) * inv).astype(F); e = (self.sum_x / N).astype(F)
            ok = (var >= 0) & ~np.isnan(var)
            in_frame = np.ones((self.h, self.w), bool)
            var_i, n_v = self._block_sum(var, ok), self._block_sum(ok.astype(F))
            e_i, e_i2, n_e = self._block_sum(e), self._block_sum(e * e), self._block_sum(in_frame.astype(F))
            with np.errstate(divide="ignore", invalid="ignore"):
                w1 = np.where(n_v == 0, 0.0, np.sqrt(var_i / n_v)); w2 = np.where(n_e == 0, 0.0, np.sqrt(e_i2 / n_e - (e_i / n_e) ** 2))
                a = (w1 - w1.min()) / (w1.max() - w1.min()) if w1.max() > w1.min() else np.zeros(self.n)
                b = (w2 - w2.min()) / (w2.max() - w2.min()) if w2.max() > w2.min() else np.zeros(self.n)
            wgt = np.nan_to_num(0.85 * a + 0.15 * b, nan=0.0)
            self.keys = (wgt * sq_user).astype(F)
            self.indices = sorted(self.indices, key=lambda i: -self.keys[i])      # stable, descending
        elif self.kind == DIFFERENCE:
            first = self.passes_done == 0
            self.passes_done += 1
            if first:
                return
            with np.errstate(divide="ignore", invalid="ignore"):
                I = (self.prev_I / self.weight[..., None]).astype(F); A = (self.half / (self.iterations // 2).astype(F)[..., None]).astype(F)
                e_p = (np.abs(I - A).sum(axis=2) / np.sqrt(I.sum(axis=2))).astype(F)
            skip = (I == 0).all(axis=2) | np.isnan(I).any(axis=2) | np.isnan(A).any(axis=2)
            err = np.where(skip, F(0), np.maximum(e_p, F(1e-2)))
            n_px = self._block_sum(np.ones((self.h, self.w), F))
            with np.errstate(divide="ignore", invalid="ignore"):
                e_blk = np.nan_to_num(self._block_sum(err) / n_px, nan=0.0)
            self.keys = (e_blk * sq_user).astype(F)
            self.indices = sorted(self.indices, key=lambda i: -self.keys[i])
        elif self.kind == UNIFORM:
            if self.n >= 2 and np.any(self.user != 1):
                self.non_zero = True
            self.indices = sorted(self.indices, key=lambda i: -self.user[i])
