"""How close a shard's main launch is to its longest chain: main-launch ms / longest per-pixel chain (segments) =
microseconds per trip if that chain alone set the time.  Compare with the idle lone-ray trip (scripts/lone_glass_probe.py)."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
W, H, S, B = 1920, 1080, 100, 50
sc = rt.build_scene(3, 32); cam = rt.camera(32, W, H, S, B)
for shard in (None, (1, 2, 8), (1, 4, 2), (3, 8, 2), (6, 8, 2), (5, 16, 2), (3, 32, 2)):
    r = rt.Renderer(0, 32); r.set_camera(cam); r.set_scene(sc)
    if shard:
        r.set_shard(*shard)
    r.init_rng(1227)
    ms = [r.render(0) for _ in range(5)]
    st = r.stats()
    r.count_segments(0)
    c = r.stats()
    print(json.dumps({"shard": shard, "render_ms": round(float(np.median(ms)), 3), "prepass_ms": round(st["prepass_ms"], 3), "main_ms": round(st["main_ms"], 3),
                      "solo_waves": st["solo_waves"], "max_chain_main": c["max_chain_main"], "segments_main": c["segments_main"],
                      "us_per_trip_if_chain_bound": round(1e3 * st["main_ms"] / max(1, c["max_chain_main"]), 3),
                      "bulk_ms_at_full_frame_rate": round(c["segments_main"] / 38.6e6, 3)}), flush=True)
    r.close()
