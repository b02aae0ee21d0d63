"""Execution profile of the render kernel (needs `python -m raytracingincuda_amd.build --stats`):
wave-level executions and active lanes of every divergent region per wave-iteration.
Usage: path_stats_probe.py [scene_id [W H S B]]      (RTIOW_PROBE_PREC=64: the fp64 kernels)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracingincuda_amd as rt
from raytracingincuda_amd import api

orig = api.lib_paths
api.lib_paths = lambda: dict(orig(), hip=os.environ.get("RTIOW_STATS_LIBRARY") or os.path.join(os.path.dirname(orig()["host"]), "librtiow_hip_stats.so"))
a = sys.argv[1:]
scene = int(a[0]) if a else 3
W, H, S, B = (int(x) for x in a[1:5]) if len(a) >= 5 else (1920, 1080, 100, 50)
names = ["iteration", "ruv_call", "ruv_round", "disk_round", "gen_primary", "shade_hit", "sky", "dielectric", "metal",
         "exact_block", "finish_call", "ieee_block", "second_div", "schlick_draw", "refill", "finish_pixel", "grid_step",
         "walk_step_1", "walk_step_2", "walk_step_3", "walk_step_4", "walk_step_5_8", "walk_step_9_up", "walk_entered", "cell_second_pair"]
prec = int(os.environ.get("RTIOW_PROBE_PREC", "32"))
r = rt.Renderer(0, prec); r.set_camera(rt.camera(prec, W, H, S, B)); r.set_scene(rt.build_scene(scene, prec))
sched = int(os.environ.get("RTIOW_PROBE_SCHED", "2"))
r.init_rng(1227); r.set_schedule(sched, 0)
lib = api.load_hip_library()
buf = (ctypes.c_ulonglong * (2 * len(names)))()
assert lib.rtiow_debug_path_stats(buf, len(buf), 1) == 0
regions = ["refill", "gen_primary", "hit_world", "hit_coop", "shade", "accumulate", "grid_setup", "grid_direct", "grid_walk", "grid_fallback", "ruv_rounds", "loop_total"]
rbuf = (ctypes.c_ulonglong * len(regions))()
assert lib.rtiow_debug_region_cycles(rbuf, len(rbuf), 1) == 0
r.render(0)
assert lib.rtiow_debug_region_cycles(rbuf, len(rbuf), 0) == 0
assert lib.rtiow_debug_path_stats(buf, len(buf), 0) == 0
v = list(buf)
it = float(v[0])
out = {"config": "scene %d %dx%d %d spp %d bounces fp%d, schedule %d (all launches together)" % (scene, W, H, S, B, prec, sched), "wave_iterations": int(it),
       "lanes_per_iteration": round(v[1] / it, 2), "per_wave_iteration": {}}
for k, n in enumerate(names[1:], 1):
    out["per_wave_iteration"][n] = {"wave_executions": round(v[2 * k] / it, 3), "active_lanes_each": round(v[2 * k + 1] / max(v[2 * k], 1), 1)}
tot = float(rbuf[len(regions) - 1]) or 1.0
out["region_share_of_loop_cycles"] = {n: round(rbuf[k] / tot, 4) for k, n in enumerate(regions)}
out["region_note"] = "grid_* lie inside hit_world, ruv_rounds inside shade; s_memtime reads cost ~10 %, shares only"
print(json.dumps(out, indent=1))
r.close()
