"""Where one rank of an N-GPU job spends a DoPasses call: python tools/shard_time_probe.py [world ...]   (runs rank 0's tile shard on this GPU)
Uses the driver's call pattern: 5 warm-up passes, then one timed call of $PASSES passes (default 20; BASELINE config 4 is 256 spp, the headline metric 64)."""
import sys, time
sys.path.insert(0, '/root/repo')
import cudatracerlib_amd as ctl
from cudatracerlib_amd import scenes, api
api.set_cache_dir('/tmp/ctl_amd_cache')
sc = scenes.synthetic_sm(1920, 1080, n_instances=2000)
scene = ctl.Scene(sc.desc, flatten=True)
import os
FUSE = [int(x) for x in os.environ.get("FUSE", "0,1").split(",")]   # FuseTraversal off / on
PASSES = int(os.environ.get("PASSES", "20"))
for world, streams in [(int(a), s) for a in (sys.argv[1:] or [1, 2, 4, 8]) for s in FUSE]:
    tr = ctl.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 8)
    p.setValue("FuseTraversal", bool(streams))
    tr.setTileShard(0, world); tr.Resize(1920, 1080); tr.InitializeScene(scene)
    img = ctl.Image(1920, 1080)
    tr.reservePasses(PASSES)
    tr.DoPasses(img, 5, new_trace=True)
    best = None
    for rep in range(3):
        t = time.perf_counter(); tr.DoPasses(img, PASSES, new_trace=False); dt = time.perf_counter() - t
        st = tr.stats()
        k = st.ms_intersect + st.ms_intersect_any + st.ms_shade + st.ms_raygen + st.ms_fused
        row = (dt * 1e3, st.rays_last_pass / dt / 1e6, st.ms_intersect + st.ms_fused, st.ms_intersect_any, st.ms_shade, st.ms_raygen, dt * 1e3 - k)
        if best is None or row[0] < best[0]: best = row
    print("passes %d world %d FuseTraversal %d: %7.2f ms  %6.0f Mrays/s (x%d = %6.0f)  closest(+fused) %.2f shadow %.2f shade %.2f raygen %.2f  outside kernels %.2f ms" % ((PASSES, world, streams) + best[:2] + (world, best[1] * world) + best[2:]), flush=True)
