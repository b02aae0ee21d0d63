"""Where the prepass's time goes: the prepass is the persistent body over samples [0, 3) in tile order, so a 3-sample frame in the
persistent schedule is the same launch (minus the 52 bytes it parks per pixel).  Per-wave timeline of the COUNT build: when the work
counter runs out and when the last wave ends; longest per-pixel chain.  Usage: prepass_tail_probe.py [S] [W H]"""
import json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracingincuda_amd as rt
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
r = rt.Renderer(0, 32, debug=True); r.set_camera(rt.camera(32, W, H, S, 50)); r.set_scene(rt.build_scene(3, 32))
r.init_rng(1227); r.set_schedule(1, 0)
ms = [round(r.render(0), 3) for _ in range(4)]
tl = r.debug_timeline(0).astype(np.float64)
st = r.stats()
tl = tl[tl[:, 2] > 0]
t0 = tl[:, 0].min()
us = lambda a: (a - t0) / 100.0                      # s_memrealtime ticks at 100 MHz
end, exh, start = us(tl[:, 2]), us(np.where(tl[:, 1] > 0, tl[:, 1], tl[:, 2])), us(tl[:, 0])
q = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])]
print(json.dumps({"config": "scene 3 %dx%d %d spp, persistent schedule (= the prepass launch)" % (W, H, S), "render_ms": ms, "waves": len(tl),
                  "wave_start_us_pcts_0_10_50_90_99_100": q(start), "counter_exhausted_us": q(exh), "wave_end_us": q(end),
                  "iterations_per_wave": q(tl[:, 3] + tl[:, 4]), "drain_iterations_per_wave": q(tl[:, 4]), "pixels_per_wave": q(tl[:, 5]),
                  "longest_chain_segments": int(st.get("max_chain_main", 0)), "segments": int(st.get("segments_main", 0))}))
r.close()
