import sys; sys.path.insert(0,'/root/repo')
import raytracingincuda_amd as rt
for prec in (32, 64):
    with rt.Renderer(0, prec) as r:
        r.set_camera(rt.camera(prec, 1, 1, 2000, 50)); r.set_scene(rt.build_scene(3, prec)); r.init_rng(1227); r.set_schedule(rt.SCHED_PERSISTENT)
        segs = r.count_segments(0); ms = min(r.render(0) for _ in range(5))
        print(prec, round(ms*1e3/segs, 3), 'us per segment')
