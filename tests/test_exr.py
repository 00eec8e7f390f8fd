"""OpenEXR decoding for environment maps and float textures (csrc/image_io.cpp decode_exr; the reference reads EXR through FreeImage).
Files come from the writer in tests/exr_encode.py (the published file layout); decoded pixels must equal the source values exactly — HALF
channels after the half -> float widening, which is exact."""
import os
import numpy as np
import pytest
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api
import exr_encode as X


def _picture(h, w, seed=2):
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.5 + 0.5 * np.sin(x / 5), 40.0 * np.exp(-((x - w / 2) ** 2 + (y - h / 3) ** 2) / 30), 0.02 + y / h], axis=2).astype(np.float32)
    img[h // 2, : w // 2] = 0.25                                      # a constant stretch: runs for the RLE codec
    img[1, 1] = [65504.0, 6e-8, 0.0]                                  # half's largest finite value, a denormal half, zero
    img += rs.uniform(0, 1e-3, img.shape).astype(np.float32)
    return img


@pytest.mark.parametrize("compression", [X.NONE, X.RLE, X.ZIPS, X.ZIP])
@pytest.mark.parametrize("kind", ["half", "float", "mixed"])
def test_scanline_files_decode_exactly(tmp_path, compression, kind):
    src = _picture(37, 29)
    dt = {"half": [np.float16] * 3, "float": [np.float32] * 3, "mixed": [np.float16, np.float32, np.float16]}[kind]
    with np.errstate(over="ignore"):
        ch = {n: src[..., k].astype(dt[k]) for k, n in enumerate("RGB")}
    ch["A"] = np.ones(src.shape[:2], np.float16)                      # an alpha channel the reader skips (sorted first in the file)
    path = os.path.join(str(tmp_path), "t.exr")
    open(path, "wb").write(X.encode(ch, compression, data_window_origin=(3, -5), line_order=1 if compression == X.ZIPS else 0))
    got = api.decode_image_file(path)
    want = np.stack([ch[n].astype(np.float32) for n in "RGB"], axis=2)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_luminance_only_and_uint_channels(tmp_path):
    src = _picture(20, 33)[..., 1]
    path = os.path.join(str(tmp_path), "y.exr")
    open(path, "wb").write(X.encode({"Y": src.astype(np.float16)}, X.ZIP))
    got = api.decode_image_file(path)
    assert np.array_equal(got[..., 0], src.astype(np.float16).astype(np.float32)) and np.array_equal(got[..., 0], got[..., 2])
    ids = (np.arange(20 * 33, dtype=np.uint32).reshape(20, 33) * 7919) % 100000
    open(path, "wb").write(X.encode({"R": ids, "G": ids, "B": ids}, X.ZIPS))
    assert np.array_equal(api.decode_image_file(path)[..., 1], ids.astype(np.float32))


def test_environment_map_from_an_exr_file(tmp_path):
    """the loader's envmap emitter reads an .exr the way it reads .hdr / .pfm: level 0 is RGBE, texel row 0 = the bottom row of the picture"""
    src = _picture(16, 32)
    d = str(tmp_path); os.makedirs(os.path.join(d, "textures"))
    open(os.path.join(d, "textures", "sky.exr"), "wb").write(X.encode({n: src[..., k].astype(np.float16) for k, n in enumerate("RGB")}, X.ZIP))
    open(os.path.join(d, "s.xml"), "w").write(
        '<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="40"/></sensor><shape type="sphere"><bsdf type="diffuse"/></shape>'
        '<emitter type="envmap"><string name="filename" value="textures/sky.exr"/></emitter></scene>')
    sc = ctl.DynamicScene(); sc.ParseMitsubaScene(os.path.join(d, "s.xml")); desc = sc.UpdateScene()
    im = desc.images[0]
    assert im.texel_type == api.TEXEL_RGBE and (im.width, im.height) == (32, 16)
    tex = np.ctypeslib.as_array(api.C.cast(im.texels, api.C.POINTER(api.C.c_uint32)), shape=(16, 32))
    e = (tex >> 24).astype(np.int32); m = np.stack([tex & 255, (tex >> 8) & 255, (tex >> 16) & 255], axis=2).astype(np.float32)
    back = m * np.exp2(e - 136.0)[..., None]
    want = src.astype(np.float16).astype(np.float32)[::-1]
    assert np.all(np.abs(back - want) <= want.max(axis=2, keepdims=True) / 128 + 1e-6)


def test_rejected_files(tmp_path):
    src = _picture(8, 8)
    ch = {n: src[..., k] for k, n in enumerate("RGB")}
    path = os.path.join(str(tmp_path), "r.exr")
    good = X.encode(ch, X.ZIP)
    piz = good.replace(b"compression\0compression\0\x01\0\0\0\x03", b"compression\0compression\0\x01\0\0\0\x04")
    open(path, "wb").write(piz)
    with pytest.raises(ctl.CtlError) as e:
        api.decode_image_file(path)
    assert e.value.code == -5 and "PIZ" in str(e.value)
    open(path, "wb").write(good[:4] + bytes([2, 2, 0, 0]) + good[8:])  # version flag 0x200: tiled
    with pytest.raises(ctl.CtlError) as e:
        api.decode_image_file(path)
    assert e.value.code == -5 and "tiled" in str(e.value)
    open(path, "wb").write(good[:len(good) - 40])                     # truncated chunk
    with pytest.raises(ctl.CtlError) as e:
        api.decode_image_file(path)
    assert e.value.code == -6
    open(path, "wb").write(b"\0" * 64)
    with pytest.raises(ctl.CtlError) as e:
        api.decode_image_file(path)
    assert e.value.code == -6
