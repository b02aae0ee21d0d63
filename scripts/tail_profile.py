"""How the main launch of the full frame ends: share of the resident waves still running over time, the capacity
lost to waves that have run out, and which waves are last (instrumented kernel: slower than the product by ~10 %)."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
args = dict(a.split("=") for a in sys.argv[1:])
W, H, S, B = int(args.get("w", 1920)), int(args.get("h", 1080)), 100, 50
r = rt.Renderer(0, 32); r.set_camera(rt.camera(32, W, H, S, B)); r.set_scene(rt.build_scene(int(args.get("scene", 3)), 32))
if "shard" in args:
    r.set_shard(*[int(x) for x in args["shard"].split(",")])
r.init_rng(1227); r.set_schedule(2, 0)
ms = r.render(0)
tl = r.debug_timeline(0).astype(np.float64)
tl = tl[tl[:, 2] > 0]
t0 = tl[:, 0].min()
end = (tl[:, 2] - t0) / 100.0
exh = np.where(tl[:, 1] > 0, (tl[:, 1] - t0) / 100.0, np.nan)
T = end.max()
curve = {"%d%%" % p: round(float((end > T * p / 100.0).mean()), 3) for p in (50, 60, 70, 75, 80, 85, 90, 95, 98)}
lost = float(((T - end) / T).mean())
order = np.argsort(end)
last = order[-int(len(end) * 0.02):]
print(json.dumps({"render_ms_uninstrumented": round(ms, 3), "instrumented_launch_us": round(float(T), 1), "waves": len(end),
                  "first_exhaustion_seen_at": round(float(np.nanmin(exh) / T), 3), "median_wave_end": round(float(np.median(end) / T), 3),
                  "share_of_waves_running_at": curve, "capacity_lost_to_finished_waves": round(lost, 4),
                  "last_2pct_waves": {"wave_id_quantiles": [int(x) for x in np.percentile(last, [0, 25, 50, 75, 100])],
                                      "iters_normal_median": float(np.median(tl[last, 3])), "iters_coop_median": float(np.median(tl[last, 4])),
                                      "pixels_median": float(np.median(tl[last, 5]))},
                  "all_waves": {"iters_normal_median": float(np.median(tl[:, 3])), "iters_coop_median": float(np.median(tl[:, 4])), "pixels_median": float(np.median(tl[:, 5]))}}))
r.close()
