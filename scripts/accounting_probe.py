"""Lane-utilisation accounting of phase B of the sorted schedule on the headline config:
wave-iterations actually executed vs segments/64 (perfect packing)."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
W, H, S, B = 1920, 1080, 100, 50
r = rt.Renderer(0, 32, debug=True); r.set_camera(rt.camera(32, W, H, S, B)); r.set_scene(rt.build_scene(3, 32))
r.init_rng(1227); r.set_schedule(2, 0)
ms = [r.render(0) for _ in range(3)]
seg = r.count_segments(0)
tl = r.debug_timeline(0).astype(np.float64)
tl = tl[tl[:, 2] > 0]
it_n, it_c = tl[:, 3].sum(), tl[:, 4].sum()
dur = (tl[:, 2] - tl[:, 0]) / 100.0
print(json.dumps({"render_ms": [round(m, 3) for m in ms], "segments": int(seg), "waves": len(tl),
                  "wave_iters_normal": int(it_n), "wave_iters_coop": int(it_c),
                  "ideal_wave_iters_all_samples": seg / 64.0,
                  "lane_utilisation_if_phaseB_is_96pct": 0.96 * seg / 64.0 / (it_n + it_c),
                  "wave_duration_us_pcts": [round(float(x), 1) for x in np.percentile(dur, [0, 10, 50, 90, 100])],
                  "us_per_iter_mean": float(dur.sum() / (it_n + it_c))}))
r.close()

# ---- per dispatch-age class (blocks are dispatched in blockIdx order, 256 CUs x 4-wave blocks)
def classes():
    r = rt.Renderer(0, 32, debug=True); r.set_camera(rt.camera(32, W, H, S, B)); r.set_scene(rt.build_scene(3, 32))
    r.init_rng(1227); r.set_schedule(2, 0); r.render(0)
    tl = r.debug_timeline(0).astype(np.float64)
    gid = np.arange(len(tl)); ok = tl[:, 2] > 0
    t0 = tl[ok, 0].min()
    for c in range(int(gid[ok].max() // 1024) + 1):
        m = ok & (gid // 1024 == c)
        end = (tl[m, 2] - t0) / 100.0 / 1000.0; exh = (tl[m, 1] - t0) / 100.0 / 1000.0; start = (tl[m, 0] - t0) / 100.0 / 1000.0
        q = lambda a: [round(float(x), 2) for x in np.percentile(a, [0, 50, 90, 100])]
        print(json.dumps({"age_class": c, "waves": int(m.sum()), "start_ms": q(start), "pool_exhausted_ms": q(exh), "end_ms": q(end),
                          "iters": q(tl[m, 3] + tl[m, 4]), "pixels": q(tl[m, 5]), "us_per_iter": q((tl[m, 2] - tl[m, 0]) / 100.0 / (tl[m, 3] + tl[m, 4]))}))
    r.close()
classes()
