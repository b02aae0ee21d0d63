import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
def main(shard=None, lr=0, wps=0, threads=0, sched=2):
    W, H, S, B = 1920, 1080, 100, 50
    r = rt.Renderer(0, 32); r.set_camera(rt.camera(32, W, H, S, B)); r.set_scene(rt.build_scene(3, 32))
    if shard: r.set_shard(*shard)
    r.init_rng(1227); r.set_schedule(sched, wps)
    ms = r.render(threads)
    tl = r.debug_timeline(threads).astype(np.float64)
    tl = tl[tl[:, 2] > 0]
    t0 = tl[:, 0].min()
    start = (tl[:, 0] - t0) / 100.0; exh = np.where(tl[:, 1] > 0, (tl[:, 1] - t0) / 100.0, np.nan); end = (tl[:, 2] - t0) / 100.0   # us
    q = lambda a: [round(float(x), 1) for x in np.atleast_1d(np.nanpercentile(a, [0, 10, 50, 90, 99, 100]))] if np.size(a) and not np.all(np.isnan(a)) else []
    print(json.dumps({"shard": shard, "sched": sched, "wps": wps, "threads": threads, "render_ms": round(ms, 3), "waves": len(tl), "start_us": q(start), "exhausted_us": q(exh), "end_us": q(end),
                      "tail_us(end-exh)": q(end - exh), "iters_normal": q(tl[:, 3]), "iters_coop": q(tl[:, 4]), "pixels_per_wave": q(tl[:, 5]),
                      "us_per_normal_iter": q((np.nan_to_num(exh, nan=0) - start)[tl[:, 3] > 0] / tl[:, 3][tl[:, 3] > 0]),
                      "us_per_coop_iter": q(((end - exh)[tl[:, 4] > 0]) / tl[:, 4][tl[:, 4] > 0])}), flush=True)
    r.close()
main(); main(shard=(1, 2, 8))
