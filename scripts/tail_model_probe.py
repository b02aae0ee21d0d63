import sys, json
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
res = {}
for S in (28, 52, 100, 197):
    for (W, H) in ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)):
        with rt.Renderer(0, 32) as r:
            r.set_camera(rt.camera(32, W, H, S, 50)); r.set_scene(rt.build_scene(3, 32)); r.init_rng(1227)
            ms = []
            for _ in range(5):
                r.render(0); st = r.stats(); ms.append((st["main_ms"], st["prepass_ms"], st["render_ms"]))
            ms.sort()
            res["%d_%dx%d" % (S, W, H)] = {"main_ms": round(ms[1][0], 3), "prepass_ms": round(ms[1][1], 3), "render_ms": round(ms[1][2], 3), "vgprs": st["vgprs"]}
print(json.dumps(res, indent=1))
# fixed term of the main launch per S from the 1080p / 2160p pair: T(n) = a n + b, n = pixels / 1080p
for S in (28, 52, 100, 197):
    t1, t4 = res["%d_1920x1080" % S]["main_ms"], res["%d_3840x2160" % S]["main_ms"]
    print(S, "samples: main launch 1080p %.3f, 2160p %.3f -> per-1080p slope %.3f ms, fixed term %.3f ms" % (t1, t4, (t4 - t1) / 3, (4 * t1 - t4) / 3))
