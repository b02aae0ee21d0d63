"""One-off: wave-level statistics of the fp32 screen (needs a -DRTIOW_SCREEN_STATS build of
rtiow_hip.hip at scripts/librtiow_stats.so)."""
import ctypes, json, os, sys
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
from raytracingincuda_amd import api
here = os.path.dirname(os.path.abspath(__file__))
orig = api.lib_paths
api.lib_paths = lambda: dict(orig(), hip=os.path.join(here, "librtiow_stats.so"))
W, H, S, B = 1920, 1080, 10, 50
r = rt.Renderer(0, 32); r.set_camera(rt.camera(32, W, H, S, B)); r.set_scene(rt.build_scene(3, 32))
r.init_rng(1227); r.set_schedule(1, 0)
lib = api.load_hip_library()
buf = (ctypes.c_ulonglong * 8)()
buf2 = (ctypes.c_ulonglong * 8)()
lib.rtiow_debug_screen_stats(buf, 1); lib.rtiow_debug_finish_stats(buf2, 1)
r.render(0)
lib.rtiow_debug_screen_stats(buf, 0); lib.rtiow_debug_finish_stats(buf2, 0)
v = list(buf)
names = ["wave_trips", "trips_branch_taken", "sphere_blocks", "lane_passes", "sphere_blocks_no_behind", "lane_passes_no_behind",
         "sphere_blocks_no_behind_beyond", "lane_passes_no_behind_beyond"]
out = dict(zip(names, v))
it = v[0] / 32.0
out["per_wave_iteration"] = {n: round(x / it, 2) for n, x in zip(names, v)}
f = list(buf2)
fn = ["finish_calls_wave", "finish_calls_lanes", "ieee_blocks_wave", "ieee_lanes", "second_div_wave", "second_div_lanes", "near_root_ok_wave", "near_root_ok_lanes"]
out["finish_per_wave_iteration"] = {n: round(x / it, 2) for n, x in zip(fn, f)}
print(json.dumps(out))
r.close()
