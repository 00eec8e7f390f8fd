#!/usr/bin/env python3
"""Progressive-JPEG fixtures for tests/test_jpeg.py: files written by Pillow's libjpeg encoder (progressive scan scripts with spectral selection and
successive approximation, 4:4:4 / 4:2:0 / greyscale, with and without restart markers) together with libjpeg's own decode of each file.

    python tests/golden/generate_jpeg.py        -> tests/golden/jpeg_progressive.npz
"""
import io, os
import numpy as np
from PIL import Image


def picture(h, w, seed=1):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    rs = np.random.RandomState(seed)
    r = 128 + 100 * np.sin(x / w * 3.1 + 0.3) * np.cos(y / h * 2.2)
    g = 128 + 90 * np.cos(x / w * 2.0 - y / h * 1.5)
    b = 40 + 170 * (x / w) * (1 - y / h) + 10 * np.sin(y / 3.0)
    img = np.stack([r, g, b], axis=2)
    patch = img[h // 3:h // 3 + 6, w // 4:w // 4 + 10]
    patch[...] = rs.randint(0, 256, patch.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


CASES = [  # name, size, mode, save options
    ("p444", (40, 56), "RGB", dict(quality=92, progressive=True, subsampling=0)),
    ("p420", (37, 53), "RGB", dict(quality=85, progressive=True, subsampling=2)),
    ("p422_rst", (48, 35), "RGB", dict(quality=75, progressive=True, subsampling=1, restart_marker_blocks=2)),
    ("pgrey", (33, 41), "L", dict(quality=90, progressive=True)),
    ("seq420_opt", (29, 47), "RGB", dict(quality=80, progressive=False, optimize=True, subsampling=2)),
]

if __name__ == "__main__":
    out = {}
    for name, size, mode, opts in CASES:
        src = picture(*size)
        im = Image.fromarray(src if mode == "RGB" else src[..., 1], mode)
        buf = io.BytesIO(); im.save(buf, "JPEG", **opts)
        data = buf.getvalue()
        dec = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        out[name + "_file"] = np.frombuffer(data, np.uint8)
        out[name + "_rgb"] = dec
        print(name, len(data), "bytes", "SOF2" if b"\xff\xc2" in data else "SOF0", "scans", data.count(b"\xff\xda"))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_progressive.npz"), **out)
