"""Which pixels end a frame?  Per-pixel take / finish times of the main launch (COUNT build, rtiow_debug_pixel_times) against the pixel's
segments, its rank in the hand-out and the wave that ran it.  Usage: pixel_finish_study.py [scene W H S B]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracingincuda_amd as rt
from raytracingincuda_amd import build
# the per-pixel stamps are compiled into a study build only (-DRTIOW_PIXEL_TIMES; the fp64 counting kernel has no registers to spare for them)
lib = os.path.join(build.LIB, "ab", "pixel_times.so")
if not os.path.exists(lib):
    build.build_variant("pixel_times", ["-DRTIOW_DEBUG_API", "-DRTIOW_PIXEL_TIMES"], verbose=False)
os.environ["RTIOW_HIP_DEBUG_LIBRARY"] = lib
a = sys.argv[1:]
scene, W, H, S, B = (int(x) for x in a[:5]) if len(a) >= 5 else (3, 1920, 1080, 100, 50)
r = rt.Renderer(0, 32, debug=True); r.set_camera(rt.camera(32, W, H, S, B)); r.set_scene(rt.build_scene(scene, 32))
r.init_rng(1227); r.set_schedule(2, 0)
ms = [round(r.render(0), 3) for _ in range(3)]
pt = r.debug_pixel_times(0).reshape(-1, 4).astype(np.int64)
st = r.stats()
r.close()
ok = pt[:, 1] != 0
t0 = pt[ok, 0].min()
take = (pt[:, 0] - t0) / 100.0e3; fin = (pt[:, 1] - t0) / 100.0e3          # ms
seg = pt[:, 2]; wave = pt[:, 3]
end = fin[ok].max()
out = {"config": "scene %d %dx%d %d spp %d bounces fp32, sorted schedule, main launch of the COUNT build" % (scene, W, H, S, B), "render_ms_plain_build": ms,
       "main_launch_ms_count_build": round(float(end), 3), "pixels": int(ok.sum())}
q = lambda x: [round(float(v), 3) for v in np.percentile(x, [0, 10, 50, 90, 99, 100])]
out["segments_pcts_0_10_50_90_99_100"] = q(seg[ok])
# the pixels that finish in the last 10 % / 5 % / 2 % of the launch
for frac in (0.10, 0.05, 0.02):
    m = ok & (fin >= end * (1 - frac))
    dur = fin[m] - take[m]
    out["finish_in_last_%d_pct" % int(frac * 100)] = {
        "pixels": int(m.sum()), "segments": q(seg[m]), "taken_at_ms": q(take[m]), "duration_ms": q(dur), "us_per_segment": q(dur * 1e3 / np.maximum(seg[m], 1)),
        "taken_in_first_5pct_of_launch": round(float((take[m] < 0.05 * end).mean()), 3), "wave_age_class": q(wave[m] // 1024)}
# pace by wave age class: us per segment of pixels with >= 500 segments
m = ok & (seg >= 500)
out["us_per_segment_of_heavy_pixels_by_age_class"] = {int(c): q(((fin - take) * 1e3 / np.maximum(seg, 1))[m & (wave // 1024 == c)]) for c in range(int(wave.max() // 1024) + 1)}
# how late do heavy pixels start?  rank error: pixels >= 1000 segments by take time
m = ok & (seg >= 1000)
out["pixels_ge_1000_segments"] = {"count": int(m.sum()), "taken_at_ms": q(take[m]), "finished_at_ms": q(fin[m]), "share_taken_after_10pct": round(float((take[m] > 0.1 * end).mean()), 4)}
# critical path estimate: for each pixel finish = take + seg * pace; what if every heavy pixel ran at the best class's median pace?
print(json.dumps(out, indent=1))
