"""Fixed term of the main launch (17 % of the 1080p frame is not steady state): main-launch time over frame sizes at several samples per pixel and bounce
limits; T(n) = a n + b from the 1080p / 2160p pair (n = pixels / 1080p).  profiles/r05/tail_model_probe.txt.      Usage: tail_model_probe.py [bounces ...]"""
import sys, json
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
bounces = [int(x) for x in sys.argv[1:]] or [50]
res = {}
for B in bounces:
    for S in (28, 52, 100, 197):
        for (W, H) in ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)):
            with rt.Renderer(0, 32) as r:
                r.set_camera(rt.camera(32, W, H, S, B)); r.set_scene(rt.build_scene(3, 32)); r.init_rng(1227)
                ms = []
                for _ in range(5):
                    r.render(0); st = r.stats(); ms.append((st["main_ms"], st["prepass_ms"], st["render_ms"]))
                ms.sort()
                chain = 0
                if (W, H) == (1920, 1080):
                    r.count_segments(0); chain = int(r.stats()["max_chain_main"])
                res["b%d_%d_%dx%d" % (B, S, W, H)] = {"main_ms": round(ms[1][0], 3), "prepass_ms": round(ms[1][1], 3), "render_ms": round(ms[1][2], 3), "longest_chain": chain}
print(json.dumps(res, indent=1))
for B in bounces:
    for S in (28, 52, 100, 197):
        t1, t4 = res["b%d_%d_1920x1080" % (B, S)]["main_ms"], res["b%d_%d_3840x2160" % (B, S)]["main_ms"]
        print("bounces %d, %d samples: main launch 1080p %.3f, 2160p %.3f -> per-1080p slope %.3f ms, fixed term %.3f ms; longest chain %d segments" % (
            B, S, t1, t4, (t4 - t1) / 3, (4 * t1 - t4) / 3, res["b%d_%d_1920x1080" % (B, S)]["longest_chain"]))
