"""Single-GPU estimate of the strong-scaling curve: time of the slowest rank's shard for N = 1, 2, 4, 8
(the real multi-GPU run adds one RCCL gather of 24.9 MB / N per rank)."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
W, H, S, B = 1920, 1080, 100, 50
sc = rt.build_scene(3, 32); cam = rt.camera(32, W, H, S, B)
base = None
for n in (1, 2, 4, 8):
    worst = 0
    for rank in range(n):                      # every rank: the slowest one is what a real run waits for
        r = rt.Renderer(0, 32); r.set_camera(cam); r.set_scene(sc); r.set_shard(rank, n, 8 if n <= 2 else 2); r.init_rng(1227)
        ms = float(np.median([r.render(0) for _ in range(5)])); r.close()
        worst = max(worst, ms)
    base = base or worst
    print(json.dumps({"n_gpus": n, "slowest_rank_ms": round(worst, 3), "speedup_vs_1": round(base / worst, 2), "efficiency": round(base / worst / n, 3)}), flush=True)
