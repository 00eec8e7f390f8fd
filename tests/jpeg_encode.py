"""Test helper: a small baseline JPEG ENCODER (ITU T.81, Huffman, 8-bit), so that the product's decoder (csrc/jpeg_decode.cpp) can be
exercised without a third-party codec: greyscale / YCbCr, 4:4:4 / 4:2:2 / 4:2:0 sampling, arbitrary quantisation tables, restart
intervals.  The Huffman tables are the encoder's own (every symbol gets a fixed-length code; the DHT segment carries them), which a
conforming decoder must accept like any other table."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
_C = np.array([[(np.sqrt(0.125) if u == 0 else 0.5) * np.cos((2 * x + 1) * u * np.pi / 16) for x in range(8)] for u in range(8)])


class _Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, value, length):
        for i in range(length - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> i) & 1); self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xFF:
                    self.out.append(0)
                self.acc = 0; self.n = 0

    def flush(self):
        while self.n:
            self.put(1, 1)


def _category(v):
    return 0 if v == 0 else int(abs(v)).bit_length()


def _seg(marker, payload):
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)


def encode(rgb, sampling=(1, 1), quant=None, restart_interval=0, grey=False):
    """rgb: (h, w, 3) uint8 (or (h, w) with grey=True); sampling = chroma sub-sampling factors (h, v) in {1, 2}; quant: 64 ints (natural order)"""
    img = np.asarray(rgb, np.float64)
    h, w = img.shape[:2]
    q = np.ones(64, np.int64) if quant is None else np.asarray(quant, np.int64).reshape(64)
    if grey:
        planes = [img if img.ndim == 2 else img[..., 0]]; fac = [(1, 1)]
    else:
        r, g, b = img[..., 0], img[..., 1], img[..., 2]
        y = 0.299 * r + 0.587 * g + 0.114 * b
        cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
        cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
        planes = [y, cb, cr]; fac = [sampling, (1, 1), (1, 1)]
    hmax, vmax = max(f[0] for f in fac), max(f[1] for f in fac)
    mcuw, mcuh = 8 * hmax, 8 * vmax
    mx, my = (w + mcuw - 1) // mcuw, (h + mcuh - 1) // mcuh
    comp = []
    for p, (fh, fv) in zip(planes, fac):
        sx, sy = hmax // fh, vmax // fv
        pad = np.pad(p, ((0, my * mcuh - h), (0, mx * mcuw - w)), mode="edge")
        if sx > 1 or sy > 1:                                       # box down-sampling of the chroma planes
            pad = pad.reshape(pad.shape[0] // sy, sy, pad.shape[1] // sx, sx).mean(axis=(1, 3))
        comp.append(pad)
    out = bytearray(b"\xFF\xD8")
    out += _seg(0xDB, bytes([0]) + bytes(int(q[z]) for z in ZIGZAG))
    nc = len(comp)
    out += _seg(0xC0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([nc]) + b"".join(bytes([i + 1, (fac[i][0] << 4) | fac[i][1], 0]) for i in range(nc)))
    # Huffman tables: DC symbols 0..11 as 4-bit codes, AC symbols 0..254 as 8-bit codes (canonical: code = index)
    out += _seg(0xC4, bytes([0x00]) + bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12)))
    out += _seg(0xC4, bytes([0x10]) + bytes([0] * 7 + [255] + [0] * 8) + bytes(range(255)))
    if restart_interval:
        out += _seg(0xDD, restart_interval.to_bytes(2, "big"))
    out += _seg(0xDA, bytes([nc]) + b"".join(bytes([i + 1, 0x00]) for i in range(nc)) + bytes([0, 63, 0]))
    bits = _Bits(); pred = [0] * nc; count = 0; rst = 0
    for j in range(my):
        for i in range(mx):
            if restart_interval and count and count % restart_interval == 0:
                bits.flush(); out += bits.out; out += bytes([0xFF, 0xD0 + (rst & 7)]); rst += 1
                bits = _Bits(); pred = [0] * nc
            count += 1
            for k in range(nc):
                fh, fv = fac[k]
                for by in range(fv):
                    for bx in range(fh):
                        y0, x0 = (j * fv + by) * 8, (i * fh + bx) * 8
                        blk = comp[k][y0:y0 + 8, x0:x0 + 8] - 128.0
                        coef = np.rint((_C @ blk @ _C.T).reshape(64) / q).astype(np.int64)[ZIGZAG]
                        diff = int(coef[0]) - pred[k]; pred[k] = int(coef[0])
                        t = _category(diff)
                        bits.put(t, 4)
                        if t:
                            bits.put(diff if diff > 0 else diff + (1 << t) - 1, t)
                        run = 0
                        last = max([z for z in range(1, 64) if coef[z] != 0], default=0)
                        for z in range(1, last + 1):
                            v = int(coef[z])
                            if v == 0:
                                run += 1; continue
                            while run > 15:
                                bits.put(0xF0, 8); run -= 16
                            s = _category(v)
                            bits.put((run << 4) | s, 8); bits.put(v if v > 0 else v + (1 << s) - 1, s); run = 0
                        if last < 63:
                            bits.put(0x00, 8)
    bits.flush(); out += bits.out
    out += b"\xFF\xD9"
    return bytes(out)
