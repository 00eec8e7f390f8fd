"""File headers that promise more pixels than the file could hold (or than a texture may have) are refused before anything is allocated for them
(csrc/image_io.cpp check_image_size, csrc/jpeg_decode.cpp; found by tools/fuzz_loaders.py: a 60-byte PGM declaring 2^31 x 2^31 pixels asked for exabytes).
Host code only."""
import os
import struct
import zlib

import pytest

import cudatracerlib_amd as ctl
from cudatracerlib_amd import api


def _png(w, h):
    ch = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    return b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + ch(b"IDAT", zlib.compress(b"\x00" * 64)) + ch(b"IEND", b"")


def _bmp(w, h):
    return b"BM" + struct.pack("<IHHI", 54, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, 0, 2835, 2835, 0, 0) + bytes(64)


def _tga(w, h, rle):
    return struct.pack("<BBBHHBHHHHBB", 0, 0, 10 if rle else 2, 0, 0, 0, 0, 0, w, h, 24, 0x20) + bytes(64)


def _exr(w, h):
    a = lambda n, t, d: n + b"\0" + t + b"\0" + struct.pack("<I", len(d)) + d
    chl = b"R\0" + struct.pack("<IBBBBii", 1, 0, 0, 0, 0, 1, 1) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head = a(b"channels", b"chlist", chl) + a(b"compression", b"compression", b"\x03") + a(b"dataWindow", b"box2i", box) + a(b"displayWindow", b"box2i", box) + \
        a(b"lineOrder", b"lineOrder", b"\0") + a(b"pixelAspectRatio", b"float", struct.pack("<f", 1)) + a(b"screenWindowCenter", b"v2f", bytes(8)) + a(b"screenWindowWidth", b"float", struct.pack("<f", 1)) + b"\0"
    return struct.pack("<II", 20000630, 2) + head + bytes(64)


def _jpeg(w, h):
    sof = struct.pack(">BHHB", 8, h, w, 3) + bytes([1, 0x11, 0, 2, 0x11, 0, 3, 0x11, 0])
    return b"\xff\xd8" + b"\xff\xc0" + struct.pack(">H", len(sof) + 2) + sof + b"\xff\xd9"


def _exr_wrapping_offset():
    """a well-formed 4x1 uncompressed header whose ONE offset-table entry is 2^64 - 4: `offset + 8` wraps around"""
    a = lambda n, t, d: n + b"\0" + t + b"\0" + struct.pack("<I", len(d)) + d
    chl = b"R\0" + struct.pack("<IBBBBii", 2, 0, 0, 0, 0, 1, 1) + b"\0"
    box = struct.pack("<iiii", 0, 0, 3, 0)
    head = a(b"channels", b"chlist", chl) + a(b"compression", b"compression", b"\x00") + a(b"dataWindow", b"box2i", box) + a(b"displayWindow", b"box2i", box) + \
        a(b"lineOrder", b"lineOrder", b"\0") + a(b"pixelAspectRatio", b"float", struct.pack("<f", 1)) + a(b"screenWindowCenter", b"v2f", bytes(8)) + a(b"screenWindowWidth", b"float", struct.pack("<f", 1)) + b"\0"
    return struct.pack("<II", 20000630, 2) + head + struct.pack("<Q", 0xFFFFFFFFFFFFFFFC) + bytes(64)


def _jpeg_many_scans():
    """a 16x16 greyscale progressive file header followed by 300 empty DC scans (each one walks every block): refused by the scan limit"""
    sof = struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x11, 0])
    dht = bytes([0x00]) + bytes([1] + [0] * 15) + bytes([0])
    sos = struct.pack(">B", 1) + bytes([1, 0x00]) + bytes([0, 0, 0])
    scan = b"\xff\xda" + struct.pack(">H", len(sos) + 2) + sos
    return b"\xff\xd8" + b"\xff\xc2" + struct.pack(">H", len(sof) + 2) + sof + b"\xff\xc4" + struct.pack(">H", len(dht) + 2) + dht + scan * 300 + b"\xff\xd9"


CASES = {
    "wrap.exr": _exr_wrapping_offset(),
    "scans.jpg": _jpeg_many_scans(),
    "huge.pgm": b"P5\n2147483648 2147483648\n255\n" + bytes(32),
    "huge.ppm": b"P3\n60000 60000\n255\n1 2 3\n",
    "huge.pfm": b"PF\n70000 3\n-1.0\n" + bytes(32),
    "huge.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2000000000 +X 2000000000\n" + bytes(32),
    "tall.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 60000 +X 4\n" + bytes(32),
    "huge.png": _png(1 << 20, 1 << 20),
    "wide.png": _png(60000, 60000),
    "huge.bmp": _bmp(30000, 30000),
    "min.bmp": _bmp(16, -(1 << 31)),
    "huge.tga": _tga(65535, 65535, False),
    "rle.tga": _tga(65535, 65535, True),
    "huge.exr": _exr(65536, 65536),
    "huge.jpg": _jpeg(65535, 65535),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_header_larger_than_the_file_is_refused(tmp_path, name):
    p = os.path.join(str(tmp_path), name)
    open(p, "wb").write(CASES[name])
    with pytest.raises(ctl.CtlError) as e:
        api.decode_image_file(p)
    assert e.value.code in (-5, -6), (e.value.code, str(e.value))     # unsupported / io error — never an allocation failure or a crash
    assert "bad_alloc" not in str(e.value) and "length_error" not in str(e.value)


WRAP = ('<scene version="0.5.0"><sensor type="perspective"><film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>'
        '<shape type="%s"><string name="filename" value="%s"/><integer name="shapeIndex" value="0"/></shape></scene>')
PLY_HEAD = "ply\nformat %s 1.0\nelement vertex %s\nproperty float x\nproperty float y\nproperty float z\nelement face %s\nproperty list uchar int vertex_indices\nend_header\n"


def _serialized(nv, nt):
    payload = struct.pack("<I", 0x1000) + b"m\0" + struct.pack("<QQ", nv, nt) + bytes(36)
    blob = struct.pack("<HH", 0x041C, 4) + zlib.compress(payload)
    return blob + struct.pack("<Q", 0) + struct.pack("<I", 1)


MESHES = {
    "count.ply": ("ply", (PLY_HEAD % ("ascii", "99999999999", "1") + "0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n").encode()),
    "short.ply": ("ply", (PLY_HEAD % ("ascii", "3", "4000000000") + "0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n").encode()),
    "neglist.ply": ("ply", (PLY_HEAD % ("ascii", "3", "1") + "0 0 0\n1 0 0\n0 1 0\n-5 0 1 2\n").encode()),
    "biglist.ply": ("ply", (PLY_HEAD % ("binary_little_endian", "1", "1")).replace("uchar int", "int int").encode() + struct.pack("<fff", 0, 0, 0) + struct.pack("<i", 0x7fffffff) + bytes(12)),
    "huge.serialized": ("serialized", _serialized(1 << 30, 1 << 30)),
    "offset.serialized": ("serialized", struct.pack("<HH", 0x041C, 4) + bytes(16) + struct.pack("<Q", 24) + struct.pack("<I", 1)),
}


@pytest.mark.parametrize("name", sorted(MESHES))
def test_mesh_header_larger_than_the_file_is_refused(tmp_path, name):
    kind, data = MESHES[name]
    d = str(tmp_path)
    open(os.path.join(d, name), "wb").write(data)
    x = os.path.join(d, "s.xml"); open(x, "w").write(WRAP % (kind, name))
    sc = ctl.DynamicScene()
    with pytest.raises(ctl.CtlError) as e:
        sc.ParseMitsubaScene(x)
    assert e.value.code in (-5, -6), (e.value.code, str(e.value))
    assert "bad_alloc" not in str(e.value) and "length_error" not in str(e.value)


SENSOR = '<sensor type="perspective"><film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>'
SCENES = {
    "deep": '<scene version="0.5.0">' + '<bsdf type="twosided">' * 100000 + '</bsdf>' * 100000 + '</scene>',                    # recursion depth of the XML reader
    "include_self": '<scene version="0.5.0">' + SENSOR + '<include filename="scene.xml"/></scene>',                                 # <include> cycle
    "default_self": '<scene version="0.5.0">' + SENSOR + '<default name="x" value="$x"/><shape type="sphere"><float name="radius" value="$x"/></shape></scene>',
    "ref_self": '<scene version="0.5.0">' + SENSOR + '<bsdf type="twosided" id="a"><ref id="a"/></bsdf></scene>',
    "group_self": '<scene version="0.5.0">' + SENSOR + '<shape type="shapegroup" id="g"><shape type="instance"><ref id="g"/></shape></shape></scene>',
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_scene_files_that_refer_to_themselves_are_refused(tmp_path, name):
    """unbounded recursion (nesting, <include>, $defaults, references) ends in an error, not in a stack overflow"""
    x = os.path.join(str(tmp_path), "scene.xml"); open(x, "w").write(SCENES[name])
    sc = ctl.DynamicScene()
    with pytest.raises(ctl.CtlError):
        sc.ParseMitsubaScene(x)
