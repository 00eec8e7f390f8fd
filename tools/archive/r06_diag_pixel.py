"""round 6: one pixel of a fuzz seed, pass by pass: GPU sample against the oracle's (frame + zero-stop image), and the oracle's vertex log.  python tools/archive/r06_diag_pixel.py SEED X Y"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import cudatracerlib_amd as gpu
from cudatracerlib_amd import scenes
import oracle
W, H, PASSES, DEPTH, RR = 96, 64, 4, 8, 5
seed, X, Y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
orc = oracle.Oracle(shared_math=True); lib = orc.lib
lib.orc_path_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
sc = scenes.fuzz_scene(seed, W, H); d = sc.desc
tables = orc.sequence_tables(PASSES)
names = "depth tri node mat model light fx fy fz pdf stype cfx cfy cfz clx cly clz dist u v ox oy oz dx dy dz".split()
for flatten in (True, False):
    scene = gpu.Scene(d, flatten=flatten)
    for k in range(PASSES):
        zs = np.zeros((H, W, 7), np.float32)
        want, _ = orc.render(d, W, H, n_passes=1, tables=tables[k:k + 1], max_path_length=DEPTH, rr_start=RR, zero_stop=zs, rows=(Y, Y + 1))
        tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", DEPTH); p.setValue("RRStartDepth", RR)
        tr.Resize(W, H); tr.InitializeScene(scene); img = gpu.Image(W, H); tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=True)
        g = img.getPixelData()[Y, X]; w = want[Y, X]; z = zs[Y, X]
        print("flat %d pass %d gpu %s | oracle %s | zero-stop %s" % (flatten, k, [float(v) for v in g[[0, 1, 2, 6]]], [float(v) for v in w[[0, 1, 2, 6]]], [float(v) for v in z[[0, 1, 2, 6]]]))
        if flatten and not np.allclose(g[:3], w[:3] + z[:3], rtol=2e-3, atol=2e-3):
            log = np.zeros(26 * 16, np.float32); rgb = np.zeros(3, np.float32); t1, t2 = tables[k]
            n = lib.orc_path_log(C.addressof(d), W, H, t1.ctypes.data, t2.ctypes.data, X, Y, 1, DEPTH, RR, log.ctypes.data, len(log), rgb.ctypes.data)
            print("   oracle sample", rgb.tolist())
            for r in log[:n].reshape(-1, 26):
                print("   ", {a: round(float(b), 6) for a, b in zip(names, r) if a in ("depth", "tri", "node", "mat", "model", "light", "fx", "pdf", "stype", "cfx", "cfy", "clx", "cly", "clz", "dist")})
            for L in range(1, DEPTH + 1):
                wl, _ = orc.render(d, W, H, n_passes=1, tables=tables[k:k + 1], max_path_length=L, rr_start=RR, rows=(Y, Y + 1))
                tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", L); p.setValue("RRStartDepth", RR)
                tr.Resize(W, H); tr.InitializeScene(scene); img = gpu.Image(W, H); tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=True)
                print("    len %d gpu %s oracle %s" % (L, [round(float(v), 6) for v in img.getPixelData()[Y, X][[0, 1, 2, 6]]], [round(float(v), 6) for v in wl[Y, X][[0, 1, 2, 6]]]))
