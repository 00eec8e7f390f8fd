import sys, json, numpy as np
sys.path.insert(0, '/root/repo')
import cudatracerlib_amd as gpu
from cudatracerlib_amd import scenes
import oracle
W, H, PASSES, DEPTH, RR = 96, 64, 4, 8, 5
seed = 392
orc = oracle.Oracle(shared_math=True)
sc = scenes.fuzz_scene(seed, W, H); d = sc.desc
tables = orc.sequence_tables(PASSES)
zs = np.zeros((H, W, 7), np.float32)
want, _ = orc.render(d, W, H, n_passes=PASSES, tables=tables, max_path_length=DEPTH, rr_start=RR, zero_stop=zs)
scene = gpu.Scene(d, flatten=True)
tr = gpu.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", DEPTH); p.setValue("RRStartDepth", RR)
tr.Resize(W, H); tr.InitializeScene(scene); img = gpu.Image(W, H)
per_pass = []
for k in range(PASSES):
    tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0)); per_pass.append(img.getPixelData().copy())
got = per_pass[-1]
ys, xs = np.nonzero(got[..., 6] != want[..., 6] + zs[..., 6])
for y, x in zip(ys, xs):
    print("pixel", x, y, "gpu w", got[y, x, 6], "oracle w", want[y, x, 6], "zs", zs[y, x, 6], "gpu rgb", got[y, x, :3].tolist(), "oracle rgb", want[y, x, :3].tolist())
    prev = 0
    for k in range(PASSES):
        w1, _ = orc.render(d, W, H, n_passes=1, tables=tables[k:k+1], max_path_length=DEPTH, rr_start=RR)
        print("  pass", k, "gpu w after", per_pass[k][y, x, 6], "oracle this pass w", w1[y, x, 6], "rgb", w1[y, x, :3].tolist())
