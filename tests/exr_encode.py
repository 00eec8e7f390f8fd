"""Minimal OpenEXR writer for tests/test_exr.py (single-part scanline files; the published file layout: magic, version, attributes, chunk offset
table, chunks).  Channels are written in alphabetical order as the format requires; `compression` is one of NONE, RLE, ZIPS, ZIP."""
import struct, zlib
import numpy as np

NONE, RLE, ZIPS, ZIP = 0, 1, 2, 3
UINT, HALF, FLOAT = 0, 1, 2


def _attr(name, type_, data):
    return name.encode() + b"\0" + type_.encode() + b"\0" + struct.pack("<i", len(data)) + data


def _reorder_and_predict(raw):
    b = np.frombuffer(raw, np.uint8)
    t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)          # even bytes, then odd bytes
    out = t.copy()
    out[1:] = (t[1:] - t[:-1] + 128 + 256) & 255                     # differences, biased
    return out.astype(np.uint8).tobytes()


def _rle(data):
    out = bytearray(); i = 0; n = len(data)
    while i < n:
        j = i + 1
        while j < n and data[j] == data[i] and j - i < 127:
            j += 1
        if j - i >= 3:                                               # a run: count - 1, byte
            out += struct.pack("b", j - i - 1) + data[i:i + 1]; i = j
        else:                                                        # literals up to the next run of three
            j = i
            while j < n and j - i < 127 and not (j + 2 < n and data[j] == data[j + 1] == data[j + 2]):
                j += 1
            out += struct.pack("b", -(j - i)) + data[i:j]; i = j
    return bytes(out)


def encode(channels, compression=ZIP, data_window_origin=(0, 0), line_order=0, extra_attributes=()):
    """channels: dict name -> 2-D array (uint32 -> UINT, float16 -> HALF, float32 -> FLOAT), all the same shape"""
    names = sorted(channels)
    h, w = channels[names[0]].shape
    kinds = {np.dtype(np.uint32): UINT, np.dtype(np.float16): HALF, np.dtype(np.float32): FLOAT}
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", kinds[channels[n].dtype], 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    x0, y0 = data_window_origin
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = struct.pack("<ii", 20000630, 2)
    hdr += _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([compression])) + _attr("dataWindow", "box2i", box)
    hdr += _attr("displayWindow", "box2i", box) + _attr("lineOrder", "lineOrder", bytes([line_order])) + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    for a in extra_attributes:
        hdr += _attr(*a)
    hdr += b"\0"
    lines = 16 if compression == ZIP else 1
    blocks = []
    for b0 in range(0, h, lines):
        raw = b"".join(np.ascontiguousarray(channels[n][y]).astype(channels[n].dtype.newbyteorder("<")).tobytes() for y in range(b0, min(h, b0 + lines)) for n in names)
        if compression == NONE:
            data = raw
        else:
            t = _reorder_and_predict(raw)
            data = _rle(t) if compression == RLE else zlib.compress(t)
            if len(data) >= len(raw):
                data = raw                                           # the format stores a block raw when the codec does not shrink it
        blocks.append((y0 + b0, data))
    order = list(range(len(blocks)))
    if line_order == 1:
        order.reverse()
    table_pos = len(hdr); pos = table_pos + 8 * len(blocks)
    offsets = [0] * len(blocks); body = b""
    for k in order:
        offsets[k] = pos + len(body)
        body += struct.pack("<ii", blocks[k][0], len(blocks[k][1])) + blocks[k][1]
    return hdr + b"".join(struct.pack("<Q", o) for o in offsets) + body
