"""Every rank's shard timed for strip heights 8/4/2/1 at N = 2/4/8 on ONE GPU (profiles/r01_strip_rows_sweep.txt, profiles/r02_strip_rows_sweep.jsonl)."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
W, H, S, B = 1920, 1080, 100, 50
sc = rt.build_scene(3, 32); cam = rt.camera(32, W, H, S, B)
for n in (2, 4, 8):
    for strip in (8, 4, 2, 1):
        times = []
        for rank in range(n):
            r = rt.Renderer(0, 32); r.set_camera(cam); r.set_scene(sc); r.set_shard(rank, n, strip); r.init_rng(1227)
            times.append(float(np.median([r.render(0) for _ in range(3)]))); r.close()
        print(json.dumps({"n_gpus": n, "strip_rows": strip, "rank_ms": [round(t, 2) for t in times], "slowest": round(max(times), 2), "mean": round(float(np.mean(times)), 2)}), flush=True)
