"""Latency of a lone ray's trip through glass (what bounds small shards): a 1-pixel frame whose
primary rays all enter the unit glass sphere at (0,1,0) from above, off-axis.  Plain library: microseconds per
segment; with the stats build (RTIOW_HIP_LIBRARY=.../librtiow_hip_stats.so): region shares of the loop cycles."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracingincuda_amd as rt
from raytracingincuda_amd import api

S, B = 2000, 50
cam = rt.camera(32, 1, 1, S, B)
c = np.array([0.3, 6.0, 0.2]); d = np.array([0.62, 1.0, 0.55]) - c          # from above, entering the glass sphere (0,1,0) off-axis: internal reflections
for k in range(3):
    cam.center[k] = c[k]; cam.pixel00_loc[k] = c[k] + d[k]
    cam.pixel_delta_u[k] = 0.0; cam.pixel_delta_v[k] = 0.0; cam.defocus_disk_u[k] = 0.0; cam.defocus_disk_v[k] = 0.0
cam.defocus_angle = 0.0
out = {}
for name, target in (("ground_sky_pixel", None), ("through_glass", cam)):
    with rt.Renderer(0, 32) as r:
        r.set_camera(target if target is not None else rt.camera(32, 1, 1, S, B)); r.set_scene(rt.build_scene(3, 32)); r.init_rng(1227)
        r.set_schedule(rt.SCHED_PERSISTENT)
        segs = r.count_segments(0)
        lib = api.load_hip_library()
        have_regions = hasattr(lib, "rtiow_debug_region_cycles")
        regions = ["refill", "gen_primary", "hit_world", "hit_coop", "shade", "accumulate", "grid_setup", "grid_direct", "grid_walk", "grid_fallback", "ruv_rounds", "loop_total"]
        rbuf = (ctypes.c_ulonglong * len(regions))()
        try:
            lib.rtiow_debug_region_cycles(rbuf, len(rbuf), 1)
        except AttributeError:
            have_regions = False
        ms = min(r.render(0) for _ in range(5))
        e = {"segments": segs, "segments_per_sample": round(segs / S, 2), "us_per_segment": round(ms * 1e3 / segs, 3)}
        if have_regions:
            lib.rtiow_debug_region_cycles(rbuf, len(rbuf), 0)
            tot = float(rbuf[-1]) or 1.0
            e["region_share"] = {n: round(rbuf[k] / tot, 3) for k, n in enumerate(regions)}
        out[name] = e
print(json.dumps(out, indent=1))
