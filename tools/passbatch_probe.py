import sys, time
sys.path.insert(0, '/root/repo')
import cudatracerlib_amd as ctl
from cudatracerlib_amd import scenes, api
api.set_cache_dir('/tmp/ctl_amd_cache')
sc = scenes.synthetic_sm(1920, 1080, n_instances=2000)
scene = ctl.Scene(sc.desc, flatten=True)
# usage: passbatch_probe.py [world:pb,pb,... ...]   (default: the sweep quoted in DESIGN.md)
cases = [(int(a.split(':')[0]), tuple(int(x) for x in a.split(':')[1].split(','))) for a in sys.argv[1:]] or [(1, (8, 16, 32, 64)), (2, (16, 32, 64)), (4, (32, 64))]
for world, pbs in cases:
    for pb in pbs:
        tr = ctl.WavefrontPathTracer(); p = tr.getParameters(); p.setValue("MaxPathLength", 8); p.setValue("PassBatch", pb)
        tr.setTileShard(0, world); tr.Resize(1920, 1080); tr.InitializeScene(scene)
        img = ctl.Image(1920, 1080)
        tr.DoPasses(img, max(2, pb), new_trace=True)
        t = time.perf_counter(); tr.DoPasses(img, 64, new_trace=False); dt = time.perf_counter() - t
        st = tr.stats()
        print("world %d PassBatch %2d: %7.1f ms  %.0f Mrays/s  intersect %.1f shadow %.1f shade %.1f raygen %.1f" % (world, pb, dt * 1e3, st.rays_last_pass / dt / 1e6, st.ms_intersect, st.ms_intersect_any, st.ms_shade, st.ms_raygen), flush=True)
