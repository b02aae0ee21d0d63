"""Scratch GPU probe: arithmetic parity, RNG init parity, render parity vs oracle, timings."""
import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
from tests.oracle_lib import Oracle

orc = Oracle()
out = {}
rng = np.random.default_rng(0)
for prec, dt in ((32, np.float32), (64, np.float64)):
    r = rt.Renderer(0, prec)
    a = (rng.standard_normal(1 << 16) * 10 ** rng.uniform(-20, 20, 1 << 16)).astype(dt)
    b = (rng.standard_normal(1 << 16) * 10 ** rng.uniform(-20, 20, 1 << 16)).astype(dt)
    c = (rng.standard_normal(1 << 16) * 10 ** rng.uniform(-20, 20, 1 << 16)).astype(dt)
    with np.errstate(all='ignore'):
        out[f'div{prec}'] = bool(np.array_equal(r.debug_ops(0, a, b).view(np.uint8), (a / b).view(np.uint8)))
        out[f'sqrt{prec}'] = bool(np.array_equal(r.debug_ops(1, np.abs(a)).view(np.uint8), np.sqrt(np.abs(a)).view(np.uint8)))
    r.close()
print(json.dumps(out), flush=True)

def run(prec, scene_id, W, H, S, B, threads=8, source=rt.SCENE_LDS, check=True):
    dt = np.float32 if prec == 32 else np.float64
    sc = rt.build_scene(scene_id, prec)
    cam = rt.camera(prec, W, H, S, B)
    r = rt.Renderer(0, prec)
    r.set_camera(cam); r.set_scene(sc); r.set_scene_source(source)
    r.init_rng(1227)
    ms = r.render(threads)
    ms2 = r.render(threads)
    img = r.read_framebuffer()
    st = r.stats()
    res = {'prec': prec, 'scene': scene_id, 'W': W, 'H': H, 'S': S, 'B': B, 'threads': threads, 'source': source,
           'ms': ms, 'ms2': ms2, 'mrays': W * H * S / ms2 / 1e3, 'rng_ms': st['rng_init_ms'], 'vgprs': st['vgprs']}
    if check:
        states = r.debug_read_rng()
        exp = orc.xorwow_states(1227, np.arange(W * H))
        res['rng_equal'] = bool(np.array_equal(states, exp))
        t = time.time()
        ref, stats = orc.render(prec, rt.compact_scene(sc), cam, 1227)
        res['oracle_s'] = time.time() - t
        res['bit_equal'] = bool(np.array_equal(ref.view(np.uint8), img.view(np.uint8)))
        d = np.abs(ref.astype(np.float64) - img.astype(np.float64))
        res['max_abs'] = float(d.max()); res['n_diff_px'] = int((d.max(axis=2) > 0).sum())
        res['sbar'] = stats[1] / stats[0]
    r.close()
    print(json.dumps(res), flush=True)
    return res

run(32, 3, 64, 40, 4, 10)
run(32, 3, 320, 192, 10, 25)
run(32, 1, 320, 192, 10, 25)
run(64, 3, 160, 96, 4, 25)
run(32, 3, 100, 60, 3, 8, threads=16)
run(32, 3, 100, 60, 3, 8, threads=4)
run(32, 3, 100, 60, 3, 8, threads=0)
run(32, 3, 320, 192, 10, 25, source=rt.SCENE_SCALAR)
for src in (rt.SCENE_LDS, rt.SCENE_SCALAR):
    for th in (0, 8):
        run(32, 3, 1920, 1080, 100, 50, threads=th, source=src, check=False)
run(32, 1, 1920, 1080, 100, 50, threads=0, check=False)
run(64, 3, 1920, 1080, 100, 50, threads=0, check=False)
