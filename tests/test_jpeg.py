"""JPEG decoding for bitmap textures (csrc/jpeg_decode.cpp; the reference reads JPEG through FreeImage / libjpeg).
Baseline files come from the encoder in tests/jpeg_encode.py; decoded texels are compared with the source picture (quantisation table of
ones: only DCT rounding and, for sub-sampled chroma, the box-down / triangle-up filter pair separate them).  Progressive files (spectral
selection + successive approximation, ten scans) were written by libjpeg (Pillow) and are compared with libjpeg's own decode of the same
file (tests/golden/jpeg_progressive.npz, generator tests/golden/generate_jpeg.py)."""
import os
import numpy as np
import pytest
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api
from jpeg_encode import encode


def _picture(h, w, seed=1):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    rs = np.random.RandomState(seed)
    r = 128 + 100 * np.sin(x / w * 3.1 + 0.3) * np.cos(y / h * 2.2)
    g = 128 + 90 * np.cos(x / w * 2.0 - y / h * 1.5)
    b = 40 + 170 * (x / w) * (1 - y / h) + 10 * np.sin(y / 3.0)
    img = np.stack([r, g, b], axis=2)
    patch = img[h // 3:h // 3 + 5, w // 4:w // 4 + 9]
    patch[...] = rs.randint(0, 256, patch.shape)                     # a patch of noise: exercises long AC runs and big coefficients
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _decode(tmp_path, name, data):
    d = str(tmp_path)
    open(os.path.join(d, name), "wb").write(data)
    open(os.path.join(d, "s.xml"), "w").write(
        '<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="40"/></sensor><shape type="sphere"><bsdf type="diffuse">'
        '<texture type="bitmap" name="reflectance"><string name="filename" value="%s"/></texture></bsdf></shape></scene>' % name)
    sc = ctl.DynamicScene()
    sc.ParseMitsubaScene(os.path.join(d, "s.xml"))
    desc = sc.UpdateScene()
    im = desc.images[0]
    assert im.texel_type == api.TEXEL_RGBCOL
    tex = np.ctypeslib.as_array(api.C.cast(im.texels, api.C.POINTER(api.C.c_uint32)), shape=(im.height, im.width)).copy()
    rgb = np.stack([tex & 0xff, (tex >> 8) & 0xff, (tex >> 16) & 0xff], axis=2).astype(np.int64)
    assert np.all((tex >> 24) == 255)
    return sc, rgb[::-1]                                            # texel row 0 is the bottom row of the picture (MIPMap.cu:565-586)


@pytest.mark.parametrize("sampling,size,restart", [((1, 1), (24, 40), 0), ((2, 2), (37, 53), 0), ((2, 1), (16, 33), 3), ((2, 2), (64, 48), 2), ((1, 1), (9, 7), 1)])
def test_colour_files(tmp_path, sampling, size, restart):
    src = _picture(*size)
    sc, got = _decode(tmp_path, "t.jpg", encode(src, sampling=sampling, restart_interval=restart))
    assert got.shape == src.shape
    err = np.abs(got - src.astype(np.int64))
    smooth = np.ones(size, bool); smooth[max(0, size[0] // 3 - 2):size[0] // 3 + 8, max(0, size[1] // 4 - 2):size[1] // 4 + 12] = False   # away from the noise patch
    if sampling == (1, 1):
        assert err[smooth].max(initial=0) <= 2 and err.max() <= 4   # DCT rounding + the YCbCr round trip (saturated noise colours clip)
    else:
        assert err[smooth].max() <= 6 and np.mean(err[smooth]) < 1.5   # chroma was box-filtered down and comes back through the triangle filter
        assert np.abs(got[~smooth].astype(float).mean() - src[~smooth].astype(float).mean()) < 6


def test_greyscale_and_coarse_quantisation(tmp_path):
    src = _picture(40, 56)[..., 1]
    sc, got = _decode(tmp_path, "g.jpeg", encode(src, grey=True))
    assert np.all(got[..., 0] == got[..., 1]) and np.all(got[..., 1] == got[..., 2])
    assert np.abs(got[..., 0] - src.astype(np.int64)).max() <= 1
    q = (1 + (np.arange(64) // 8 + np.arange(64) % 8) * 3)        # a JPEG-like table: coarser towards high frequencies
    sc2, got2 = _decode(tmp_path, "q.jpg", encode(_picture(32, 32), quant=q))
    assert np.abs(got2 - _picture(32, 32).astype(np.int64)).mean() < 4


def test_rejected_and_damaged_files(tmp_path):
    good = encode(_picture(16, 16))
    lossless = good.replace(b"\xFF\xC0", b"\xFF\xC3", 1)           # pretend lossless (SOF3)
    with pytest.raises(ctl.CtlError) as e:
        _decode(tmp_path, "p.jpg", lossless)
    assert e.value.code == -5 and "lossless" in str(e.value)
    with pytest.raises(ctl.CtlError) as e:
        _decode(tmp_path, "n.jpg", b"not a jpeg at all")
    assert e.value.code == -6
    sc, cut = _decode(tmp_path, "c.jpg", good[:len(good) * 2 // 3])  # truncated entropy data decodes (to grey) instead of crashing
    assert cut.shape == (16, 16, 3)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_progressive.npz")


@pytest.mark.parametrize("name", ["p444", "p420", "p422_rst", "pgrey", "seq420_opt"])
def test_progressive_files_against_libjpeg(tmp_path, name):
    """DC first / refinement scans, AC bands with end-of-band runs, AC refinement (correction bits), non-interleaved scans over the
    component's own block grid, restart markers inside progressive scans; and an optimised-Huffman sequential file.  libjpeg's integer
    IDCT and this decoder's float IDCT round differently: texels agree to two 8-bit steps, 95 % of them to one."""
    g = np.load(GOLDEN)
    data = g[name + "_file"].tobytes(); want = g[name + "_rgb"].astype(np.int64)
    assert (b"\xff\xc2" in data) == name.startswith("p")
    sc, got = _decode(tmp_path, name + ".jpg", data)
    assert got.shape == want.shape
    err = np.abs(got - want)
    assert err.max() <= 2, err.max()
    assert (err <= 1).mean() >= 0.95


def test_progressive_truncated_file_decodes_what_it_has(tmp_path):
    """a progressive file cut after its first scans is a coarse picture, not an error (later scans only refine)"""
    g = np.load(GOLDEN)
    data = g["p444_file"].tobytes(); want = g["p444_rgb"].astype(np.int64)
    cut = data[:data.index(b"\xff\xda", data.index(b"\xff\xda") + 2)]      # keep the first scan (DC of all components) only
    sc, got = _decode(tmp_path, "cut.jpg", cut + b"\xff\xd9")
    assert got.shape == want.shape
    blocks = want.reshape(5, 8, 7, 8, 3).mean(axis=(1, 3)); got_blocks = got.reshape(5, 8, 7, 8, 3).mean(axis=(1, 3))
    assert np.abs(blocks - got_blocks).max() < 12                    # block means survive (DC was sent with a point transform of one bit)
