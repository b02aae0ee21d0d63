import sys; sys.path.insert(0,'/root/repo')
import numpy as np, raytracingincuda_amd as rt
def render(prec, scene, source, sched, W=1920, H=1080, S=100, B=50):
    with rt.Renderer(0, prec) as r:
        r.set_camera(rt.camera(prec, W, H, S, B)); r.set_scene(rt.build_scene(scene, prec)); r.set_scene_source(source); r.set_schedule(sched); r.init_rng(1227)
        ms = r.render(0); return r.read_framebuffer(), ms
for prec, scene in ((32,1),(32,2),(32,3),(64,1),(64,2),(64,3)):
    a, ma = render(prec, scene, rt.SCENE_GRID, rt.SCHED_SORTED)
    b, mb = render(prec, scene, rt.SCENE_LDS_EXACT, rt.SCHED_STATIC)
    c, mc = render(prec, scene, rt.SCENE_GRID, rt.SCHED_STATIC)
    print('f%d scene %d 1920x1080x100: grid+sorted %.1f ms, exact+static %.1f ms, grid+static %.1f ms; identical: %s %s' % (prec, scene, ma, mb, mc, np.array_equal(a.view(np.uint8), b.view(np.uint8)), np.array_equal(c.view(np.uint8), b.view(np.uint8))), flush=True)
