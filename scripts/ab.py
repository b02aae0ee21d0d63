"""A/B timing of kernel variants in ONE process, interleaved rounds (median/min)."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt

def main(scene_id=3, W=1920, H=1080, S=100, B=50, prec=32, rounds=5, shard=None, variants=None):
    sc = rt.build_scene(scene_id, prec); cam = rt.camera(prec, W, H, S, B)
    r = rt.Renderer(0, prec); r.set_camera(cam); r.set_scene(sc)
    if shard: r.set_shard(*shard)
    r.init_rng(1227)
    # (source, threads, sched, waves_per_simd)
    variants = variants or [(2, 0, 2, 0), (0, 0, 2, 0), (0, 8, 2, 0)]
    times = {v: [] for v in variants}
    ref = None
    for rd in range(rounds):
        for v in variants:
            r.set_scene_source(v[0]); r.set_schedule(v[2], v[3])
            times[v].append(r.render(v[1]))
            if rd == 0:
                img = r.read_framebuffer()
                if ref is None: ref = img
                assert np.array_equal(ref.view(np.uint8), img.view(np.uint8)), v
    rows = r.local_rows
    for v in variants:
        t = np.array(times[v])
        print(json.dumps({"scene": scene_id, "prec": prec, "shard": shard, "source": v[0], "threads": v[1], "sched": v[2], "wps": v[3],
                          "ms_median": round(float(np.median(t)), 3), "ms_min": round(float(t.min()), 3),
                          "mrays": round(W * rows * S / float(np.median(t)) / 1e3, 1)}), flush=True)
    r.close()

if __name__ == "__main__":
    main(3)
    main(3, shard=(3, 8, 8), variants=[(2, 0, 2, 0), (0, 0, 2, 0)])
    main(3, shard=(1, 2, 8), variants=[(2, 0, 2, 0), (0, 0, 2, 0)])
    main(1, rounds=3, variants=[(2, 0, 1, 0), (0, 0, 1, 0), (2, 0, 2, 0), (0, 0, 2, 0)])
    main(3, prec=64, rounds=3, variants=[(2, 0, 2, 0), (0, 0, 2, 0), (0, 0, 1, 0)])
