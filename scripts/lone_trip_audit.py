#!/usr/bin/env python3
"""Where does a LONE ray's trip go?  (the chain-latency regime bounds small frames and every shard at N >= 4.)

A 1 x 1 frame leaves one live lane in one wave on an idle GPU: every instruction of its trip is on the critical path, nothing overlaps.
  (a) product library: HIP-event time / segments = us per trip, and the EFFECTIVE shader clock of that launch (rtiow_stats.main_clock_mhz:
      s_memtime / s_memrealtime of the wave itself) -> cycles per trip at the clock the chip really ran;
  (b) instrumented library (-DRTIOW_PATH_STATS, lib/librtiow_hip_stats.so): s_memtime cycles per region of the loop, per trip, and how many
      times each divergent block ran per trip (the stamps cost ~10 %: shares, not absolutes).
Usage: lone_trip_audit.py [scene_id [precision]]      Output: one JSON object.
"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracingincuda_amd as rt
from raytracingincuda_amd import api

scene_id = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S, B = 400, 50
out = {"config": "scene %d, 1x1 frame, %d spp, %d bounces, fp%d" % (scene_id, S, B, prec)}
with rt.Renderer(0, prec) as r:
    r.set_camera(rt.camera(prec, 1, 1, S, B)); r.set_scene(rt.build_scene(scene_id, prec)); r.init_rng(1227)
    segs = r.count_segments(0)
    runs = []
    for _ in range(6):
        ms = r.render(0)
        st = r.stats()
        runs.append((ms, st["main_ms"], st["main_clock_mhz"], st["main_wave0_ms"]))
    ms, main_ms, mhz, w0 = min(runs)
    out["product"] = {"segments": segs, "render_ms": round(ms, 4), "us_per_trip": round(ms * 1e3 / segs, 4), "effective_clock_mhz": round(mhz, 1),
                      "nominal_clock_mhz": st["clock_mhz"], "cycles_per_trip_at_effective_clock": round(ms * 1e-3 * mhz * 1e6 / segs, 1) if mhz else None,
                      "wave_life_ms": round(w0, 4), "all_runs_ms_and_mhz": [(round(a, 4), round(c)) for a, _, c, _ in runs]}

stats_lib = os.environ.get("RTIOW_STATS_LIBRARY") or os.path.join(os.path.dirname(api.lib_paths()["host"]), "librtiow_hip_stats.so")
if prec == 32 and os.path.exists(stats_lib):
    # a second process image of the library: the instrumented build (region clocks + execution counts)
    import subprocess
    code = r'''
import ctypes, json, os, sys
sys.path.insert(0, %r)
import raytracingincuda_amd as rt
from raytracingincuda_amd import api
orig = api.lib_paths
api.lib_paths = lambda: dict(orig(), hip=%r)
names = ["iteration", "ruv_call", "ruv_round", "disk_round", "gen_primary", "shade_hit", "sky", "dielectric", "metal", "exact_block", "finish_call", "ieee_block", "second_div",
         "schlick_draw", "refill", "finish_pixel", "grid_step", "walk_step_1", "walk_step_2", "walk_step_3", "walk_step_4", "walk_step_5_8", "walk_step_9_up", "walk_entered", "cell_second_pair"]
regions = ["refill", "gen_primary", "hit_world", "hit_coop", "shade", "accumulate", "grid_setup", "grid_direct", "grid_walk", "grid_fallback", "ruv_rounds", "loop_total"]
r = rt.Renderer(0, 32); r.set_camera(rt.camera(32, 1, 1, %d, %d)); r.set_scene(rt.build_scene(%d, 32)); r.init_rng(1227); r.set_schedule(2, 0)
lib = api.load_hip_library()
buf = (ctypes.c_ulonglong * (2 * len(names)))(); rbuf = (ctypes.c_ulonglong * len(regions))()
r.render(0)
assert lib.rtiow_debug_path_stats(buf, len(buf), 1) == 0 and lib.rtiow_debug_region_cycles(rbuf, len(rbuf), 1) == 0
ms = r.render(0)
assert lib.rtiow_debug_region_cycles(rbuf, len(rbuf), 0) == 0 and lib.rtiow_debug_path_stats(buf, len(buf), 0) == 0
v = list(buf); it = float(v[0])
print(json.dumps({"trips": int(it), "instrumented_render_ms": ms,
                  "cycles_per_trip_by_region": {n: round(rbuf[k] / it, 1) for k, n in enumerate(regions)},
                  "executions_per_trip": {n: round(v[2 * k] / it, 3) for k, n in enumerate(names) if k and v[2 * k]}}))
r.close()
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), stats_lib, S, B, scene_id)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    out["instrumented"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 and p.stdout.strip() else {"error": p.stderr[-400:]}
print(json.dumps(out, indent=1))
