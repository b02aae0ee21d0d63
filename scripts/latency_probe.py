"""Per-iteration latency / throughput probes."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt

def run(W, H, S, B, sched, threads=0, scene=3, reps=3, shard=None):
    sc = rt.build_scene(scene, 32); cam = rt.camera(32, W, H, S, B)
    r = rt.Renderer(0, 32); r.set_camera(cam); r.set_scene(sc); r.set_schedule(sched, 0)
    if shard: r.set_shard(*shard)
    r.init_rng(1227)
    ms = min(r.render(threads) for _ in range(reps))
    seg = r.count_segments(threads)
    st = r.stats()
    npx = W * r.local_rows
    r.close()
    print(json.dumps({"W": W, "H": H, "S": S, "B": B, "sched": sched, "shard": shard, "ms": round(ms, 4),
                      "mean_iters_per_pixel": round(seg / npx, 1), "M_wave_iters_per_s_at_full_util": round(seg / 64 / ms / 1e3, 1),
                      "mrays": round(npx * S / ms / 1e3, 1), "blocks": st["grid_blocks"]}), flush=True)

for sched in (0, 1):
    for B in (1, 2, 3, 4, 8, 50):
        run(1920, 1080, 100, B, sched)
    for B in (2, 4, 50):
        run(1920, 1080, 100, B, sched, shard=(3, 8, 8))
