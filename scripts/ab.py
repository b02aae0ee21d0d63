"""A/B timing of kernel variants in ONE process, interleaved rounds (median/min)."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt

def main(scene_id=3, W=1920, H=1080, S=100, B=50, prec=32, rounds=5):
    sc = rt.build_scene(scene_id, prec); cam = rt.camera(prec, W, H, S, B)
    r = rt.Renderer(0, prec); r.set_camera(cam); r.set_scene(sc); r.init_rng(1227)
    variants = [(a, s, t) for a in (rt.ALGO_FILTERED, rt.ALGO_DIRECT) for s in (rt.SCENE_LDS, rt.SCENE_SCALAR) for t in (0, 8)]
    times = {v: [] for v in variants}
    ref = None
    for rd in range(rounds):
        for v in variants:
            r.set_algorithm(v[0]); r.set_scene_source(v[1])
            times[v].append(r.render(v[2]))
            if rd == 0:
                img = r.read_framebuffer()
                if ref is None: ref = img
                assert np.array_equal(ref.view(np.uint8), img.view(np.uint8)), v
    for v in variants:
        t = np.array(times[v])
        print(json.dumps({"scene": scene_id, "prec": prec, "algo": v[0], "source": v[1], "threads": v[2], "ms_median": float(np.median(t)),
                          "ms_min": float(t.min()), "mrays": W * H * S / float(np.median(t)) / 1e3, "vgprs": r.stats()["vgprs"]}), flush=True)
    r.close()

if __name__ == "__main__":
    main(3); main(1, rounds=3); main(3, prec=64, rounds=3)
