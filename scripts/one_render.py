"""One configuration, a few renders (for rocprofv3 runs)."""
import argparse, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import raytracingincuda_amd as rt
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=3); ap.add_argument("--w", type=int, default=1920); ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--s", type=int, default=100); ap.add_argument("--b", type=int, default=50); ap.add_argument("--prec", type=int, default=32)
ap.add_argument("--sched", type=int, default=1); ap.add_argument("--source", type=int, default=3); ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--wps", type=int, default=0)
ap.add_argument("--shard", default="", help="rank,nranks,strip_rows")
ap.add_argument("--md5", action="store_true", help="also print the md5 of the last framebuffer")
ap.add_argument("--build-id", action="store_true", help="print the library's build id and the render times on lines of their own (scripts/pmc_passes.py)")
a = ap.parse_args()
r = rt.Renderer(0, a.prec); r.set_camera(rt.camera(a.prec, a.w, a.h, a.s, a.b)); r.set_scene(rt.build_scene(a.scene, a.prec))
if a.shard:
    r.set_shard(*[int(x) for x in a.shard.split(",")])
r.set_schedule(a.sched, a.wps); r.set_scene_source(a.source); r.init_rng(1227)
ms = [round(r.render(a.threads), 3) for _ in range(a.reps)]
print(ms, r.stats())
if a.build_id:
    print("build_id", rt.build_id())
    print("render_ms", *ms)
if a.md5:
    import hashlib
    print("md5", hashlib.md5(r.read_framebuffer().tobytes()).hexdigest())
r.close()
