#!/usr/bin/env python3
"""Mutation fuzzing of the file front-ends (image decoders, OBJ / PLY / .serialized mesh readers, the Mitsuba XML loader) against the AddressSanitizer
build of the library (tools/asan_check.sh leaves it in /tmp/ctl_asan).  Seeds are valid files written by the test-side encoders; every mutant is handed to
ctl_decode_image_file or ctl_parse_mitsuba_scene in a worker process.  A CtlError is the expected answer to a damaged file; a crash, an ASan report or a
time-out is a finding and the mutant is kept under <out>/.

    tools/asan_check.sh -k nothing          # builds /tmp/ctl_asan/libctl_amd.so
    python tools/fuzz_loaders.py [--per-seed 400] [--procs 8] [--out /tmp/ctl_fuzz]
"""
import argparse
import glob
import os
import struct
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def seeds(d):
    """-> list of (name, kind, bytes, aux files {name: bytes});  kind: image | obj | ply | serialized | xml"""
    import exr_encode as X
    import jpeg_encode as J
    from cudatracerlib_amd import scenes
    rs = np.random.RandomState(3)
    out = []
    img = (rs.rand(12, 20, 3) * 255).astype(np.uint8)
    ch = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    for name, ctype, arr in (("rgb.png", 2, img), ("rgba.png", 6, np.concatenate([img, img[..., :1]], 2)), ("grey.png", 0, img[..., :1])):
        raw = b"".join(b"\x00" + arr[y].tobytes() for y in range(arr.shape[0]))
        out.append((name, "image", b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", arr.shape[1], arr.shape[0], 8, ctype, 0, 0, 0)) + ch(b"IDAT", zlib.compress(raw)) + ch(b"IEND", b""), {}))
    pal = b"".join(bytes([i * 16, 255 - i * 16, i]) for i in range(16))
    raw = b"".join(b"\x00" + bytes(((x + y) % 16) << 4 | ((x * y) % 16) for x in range(10)) for y in range(12))
    out.append(("pal4.png", "image", b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", 20, 12, 4, 3, 0, 0, 0)) + ch(b"PLTE", pal) + ch(b"IDAT", zlib.compress(raw)) + ch(b"IEND", b""), {}))
    out.append(("base.jpg", "image", J.encode(img.astype(np.uint8), sampling=(2, 2), restart_interval=2), {}))
    out.append(("grey.jpg", "image", J.encode(img.astype(np.uint8), grey=True), {}))
    g = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_progressive.npz"))
    for k in ("p420_file", "p422_rst_file", "pgrey_file"):
        out.append((k.replace("_file", ".jpg"), "image", g[k].tobytes(), {}))
    f = (rs.rand(9, 13, 3) * 4).astype(np.float32)
    for comp in (X.NONE, X.RLE, X.ZIPS, X.ZIP):
        out.append(("c%d.exr" % comp, "image", X.encode({"R": f[..., 0].astype(np.float16), "G": f[..., 1], "B": f[..., 2].astype(np.float16)}, comp), {}))
    m = f.max(axis=2); man, ex = np.frexp(m)
    rgbe = np.concatenate([(f * (man * 256.0 / m)[..., None]).astype(np.uint8), (ex + 128).astype(np.uint8)[..., None]], 2)
    out.append(("flat.hdr", "image", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 9 +X 13\n" + rgbe.tobytes(), {}))
    rle = b""
    big = np.tile(rgbe[:1, :1], (2, 16, 1))
    for y in range(2):                      # new-style run-length scanlines
        rle += bytes([2, 2, 0, 16]) + b"".join(bytes([128 + 16, int(big[y, 0, c])]) for c in range(4))
    out.append(("rle.hdr", "image", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 16\n" + rle, {}))
    out.append(("c.pfm", "image", b"PF\n13 9\n-1.0\n" + f[::-1].tobytes(), {}))
    out.append(("g.pfm", "image", b"Pf\n13 9\n1.0\n" + f[::-1, :, 0].astype(">f4").tobytes(), {}))
    out.append(("p6.ppm", "image", b"P6\n# c\n20 12\n255\n" + img.tobytes(), {}))
    out.append(("p3.ppm", "image", ("P3\n4 2\n15\n" + " ".join(str(int(v) % 16) for v in img[:2, :4].ravel()) + "\n").encode(), {}))
    out.append(("p5.pgm", "image", b"P5\n20 12\n65535\n" + img[..., 0].astype(">u2").tobytes(), {}))
    bmp_rows = b"".join(img[y, :, ::-1].tobytes() for y in range(11, -1, -1))
    out.append(("t.bmp", "image", b"BM" + struct.pack("<IHHI", 54 + len(bmp_rows), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 20, 12, 1, 24, 0, len(bmp_rows), 2835, 2835, 0, 0) + bmp_rows, {}))
    out.append(("t.tga", "image", struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 20, 12, 24, 0x20) + img[..., ::-1].tobytes(), {}))
    tga_rle = b"".join(bytes([128 + 19]) + bytes(img[y, 0, ::-1]) for y in range(12))
    out.append(("rle.tga", "image", struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, 20, 12, 24, 0x20) + tga_rle, {}))
    V, F = scenes.icosphere(1)
    N = V / np.linalg.norm(V, axis=1, keepdims=True)
    obj = "mtllib m.mtl\no ball\n" + "".join("v %g %g %g\n" % tuple(v) for v in V) + "".join("vt %g %g\n" % (v[0], v[1]) for v in V) + "".join("vn %g %g %g\n" % tuple(n) for n in N)
    obj += "usemtl a\n" + "".join("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a + 1, a + 1, a + 1, b + 1, b + 1, b + 1, c + 1, c + 1, c + 1) for a, b, c in F[:40])
    obj += "usemtl b\ng part\n" + "".join("f %d//%d %d//%d %d//%d %d//%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1, a + 1, a + 1) for a, b, c in F[40:])
    mtl = "newmtl a\nKd 0.5 0.4 0.3\nKs 0.1 0.1 0.1\nNs 20\nnewmtl b\nKd 0.1 0.2 0.9\nd 0.5\nillum 2\n"
    out.append(("m.obj", "obj", obj.encode(), {"m.mtl": mtl.encode()}))
    ply_a = "ply\nformat ascii 1.0\ncomment x\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(V), len(F))
    ply_a += "".join("%g %g %g %g %g %g\n" % (*v, *n) for v, n in zip(V, N)) + "".join("3 %d %d %d\n" % tuple(t) for t in F)
    out.append(("a.ply", "ply", ply_a.encode(), {}))
    ply_b = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float u\nproperty float v\nelement face %d\nproperty list uchar uint vertex_indices\nend_header\n" % (len(V), len(F))).encode()
    ply_b += np.concatenate([V, V[:, :2]], 1).astype("<f4").tobytes() + b"".join(b"\x03" + np.asarray(t, "<u4").tobytes() for t in F)
    out.append(("b.ply", "ply", ply_b, {}))
    blob = scenes._serialized_mesh(V.astype(np.float32), np.asarray(F, np.uint32), N.astype(np.float32))
    out.append(("m.serialized", "serialized", blob + struct.pack("<Q", 0) + struct.pack("<I", 1), {}))     # one sub-mesh, the offset table, the count
    sd = os.path.join(d, "_interior"); os.makedirs(sd, exist_ok=True)
    xml_path = scenes.write_interior_mitsuba(sd, 32, 18)
    aux = {}
    for p in glob.glob(os.path.join(sd, "**", "*"), recursive=True):
        if os.path.isfile(p) and p != xml_path:
            aux[os.path.relpath(p, sd)] = open(p, "rb").read()
    out.append(("scene.xml", "xml", open(xml_path, "rb").read(), aux))
    return out


def mutate(data, rs):
    b = bytearray(data)
    n = len(b)
    how = rs.randint(0, 7)
    if how == 0 and n > 8:                       # truncate
        del b[rs.randint(1, n):]
    elif how == 1:                               # flip a few bytes
        for _ in range(rs.randint(1, 9)):
            b[rs.randint(0, n)] = rs.randint(0, 256)
    elif how == 2 and n > 8:                     # a 4-byte field gets an extreme value
        p = rs.randint(0, n - 4)
        b[p:p + 4] = [b"\x00\x00\x00\x00", b"\xff\xff\xff\xff", b"\x7f\xff\xff\xff", b"\xff\xff\xff\x7f", b"\x80\x00\x00\x00", b"\x00\x00\x00\x80", b"\x00\x00\x01\x00"][rs.randint(0, 7)]
    elif how == 3:                               # insert noise
        p = rs.randint(0, n + 1); b[p:p] = bytes(rs.randint(0, 256, rs.randint(1, 33)).tolist())
    elif how == 4 and n > 16:                    # delete a block
        p = rs.randint(0, n - 8); del b[p:p + rs.randint(1, min(64, n - p))]
    elif how == 5 and n > 16:                    # duplicate a block
        p = rs.randint(0, n - 8); q = p + rs.randint(1, min(64, n - p)); b[q:q] = b[p:q]
    else:                                        # ASCII digit runs become huge / negative / empty numbers (text formats)
        digits = [i for i in range(n) if 48 <= b[i] <= 57]
        if digits:
            p = digits[rs.randint(0, len(digits))]
            b[p:p + 1] = [b"99999999999", b"-1", b"", b"4294967296", b"1e39", b"nan"][rs.randint(0, 6)]
        else:
            b[rs.randint(0, n)] ^= 0xff
    return bytes(b)


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api
os.environ["CTL_LOADER_LENIENT"] = "0"
XML = '<scene version="0.5.0"><sensor type="perspective"><film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>' \
      '<shape type="%%s"><string name="filename" value="%%s"/><integer name="shapeIndex" value="0"/></shape></scene>'
for line in open(sys.argv[1]):
    kind, path = line.rstrip("\n").split("\t")
    print("BEGIN", path, flush=True)
    try:
        if kind == "image":
            api.decode_image_file(path)
        else:
            if kind != "xml":
                x = path + ".xml"; open(x, "w").write(XML %% (kind, os.path.basename(path))); path = x
            sc = ctl.DynamicScene(); sc.ParseMitsubaScene(path); sc.UpdateScene()
    except ctl.CtlError:
        pass
    print("END", flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-seed", type=int, default=400); ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--out", default="/tmp/ctl_fuzz"); ap.add_argument("--lib", default="/tmp/ctl_asan/libctl_amd.so"); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    work = os.path.join(a.out, "work"); os.makedirs(work, exist_ok=True)
    S = seeds(work)
    rs = np.random.RandomState(a.seed)
    jobs = []
    for name, kind, data, aux in S:
        for k in range(a.per_seed + 1):
            d = os.path.join(work, "%s_%04d" % (name.replace(".", "_"), k)); os.makedirs(d, exist_ok=True)
            for an, ab in aux.items():
                p = os.path.join(d, an); os.makedirs(os.path.dirname(p), exist_ok=True)
                # now and then the auxiliary file (material library, texture, mesh of the scene) is the one that is damaged
                open(p, "wb").write(mutate(ab, rs) if (k and rs.rand() < 0.3) else ab)
            p = os.path.join(d, name)
            open(p, "wb").write(data if k == 0 else mutate(data, rs))     # k == 0: the seed itself must load
            jobs.append((kind, p))
    print("%d seeds, %d inputs" % (len(S), len(jobs)), flush=True)
    env = dict(os.environ)
    if os.path.exists(a.lib):
        asan = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))[0]
        env.update(CTL_AMD_LIB=a.lib, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=4096")
        print("AddressSanitizer build:", a.lib)
    else:
        print("no ASan build at %s: fuzzing the regular library" % a.lib)
    wpath = os.path.join(a.out, "worker.py"); open(wpath, "w").write(WORKER % dict(root=ROOT))
    findings = []
    pending = [jobs[i::a.procs] for i in range(a.procs)]
    rnd = 0
    while any(pending):
        procs = []
        for i, chunk in enumerate(pending):
            if not chunk:
                continue
            lst = os.path.join(a.out, "list_%d_%d.txt" % (rnd, i)); open(lst, "w").write("".join("%s\t%s\n" % j for j in chunk))
            procs.append((i, chunk, subprocess.Popen([sys.executable, wpath, lst], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        nxt = [[] for _ in pending]
        for i, chunk, p in procs:
            try:
                so, se = p.communicate(timeout=1800)
            except subprocess.TimeoutExpired:
                p.kill(); so, se = p.communicate(); se += "\nTIMEOUT"
            began = [l[6:] for l in so.splitlines() if l.startswith("BEGIN ")]
            ends = so.count("\nEND") + (1 if so.startswith("END") else 0)
            if len(began) > ends or p.returncode != 0:       # the worker died inside (or right after) began[-1]
                bad = began[-1] if began else chunk[0][1]
                findings.append((bad, p.returncode, se[-1500:]))
                idx = [j for j, c in enumerate(chunk) if c[1] == bad or c[1] + ".xml" == bad]
                nxt[i] = chunk[idx[0] + 1:] if idx else []
        pending = nxt; rnd += 1
    keep = os.path.join(a.out, "findings"); os.makedirs(keep, exist_ok=True)
    for bad, rc, err in findings:
        print("\n==== FINDING rc=%s %s\n%s" % (rc, bad, err))
        try:
            dst = os.path.join(keep, os.path.basename(os.path.dirname(bad))); os.makedirs(dst, exist_ok=True)
            subprocess.call(["cp", "-r", os.path.dirname(bad) + "/.", dst])
        except Exception:
            pass
    print("\n%d inputs, %d findings" % (len(jobs), len(findings)))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
