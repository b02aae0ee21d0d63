"""Soak run for the grid walk (not a test; run by hand on a GPU box): many random scenes, each probed with adversarial
rays through rtiow_debug_hit_world -- grid walk vs the exact loop, ray by ray.  Prints one line per scene and a
summary; exits non-zero on the first mismatch.      python tests/studies/grid_soak.py [n_scenes] [rays_per_family] [first_seed] [max_spheres]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import raytracingincuda_amd as rt  # noqa: E402
from tests.test_gpu_parity import _adversarial_rays, _random_field  # noqa: E402
from tests.test_grid_plan import _plan  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_each = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
first_seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
max_n = int(sys.argv[4]) if len(sys.argv) > 4 else 900        # up to ~2000 spheres still get a grid (larger scenes leave LDS): the longest walks, 80-90 cells
used = rays_total = 0
for seed in range(n_scenes):
    rng = np.random.default_rng(first_seed + seed)
    prec = 32 if seed % 2 == 0 else 64
    n = int(rng.integers(40, max_n))
    half = float(np.exp(rng.uniform(np.log(2), np.log(60))))
    centre = rng.uniform(-80, 80, 3) * (seed % 3 == 0)
    rscale = float(np.exp(rng.uniform(np.log(0.02), np.log(1.0))))
    rlaw = [lambda g: rscale, lambda g: float(g.uniform(0.2, 1.5)) * rscale, lambda g: float(np.exp(g.uniform(-2, 1))) * rscale][seed % 3]
    yspread = float(rng.choice([0.0, 0.3, 2.0, 10.0])) * rscale * 5
    sc = _random_field(rng, prec, n, (centre[0] - half, centre[0] + half), (centre[2] - half, centre[2] + half), rlaw,
                       lambda g, r: centre[1] + r + float(g.uniform(0, yspread)) if yspread else centre[1] + r,
                       ground=seed % 4 != 3, big=[(centre[0], centre[1] + 3 * rscale, centre[2], 3 * rscale)] if seed % 5 == 0 else [])
    cr = sc["center_radius"].astype(np.float64)
    if sc["center_radius"][0][3] > 100:                    # keep the ground under the field
        sc["center_radius"][0][0] = centre[0]; sc["center_radius"][0][1] = centre[1] - sc["center_radius"][0][3]; sc["center_radius"][0][2] = centre[2]
        cr = sc["center_radius"].astype(np.float64)
    pl = _plan(rt, cr)
    if pl["usable"]:
        rays = _adversarial_rays(rng, cr, pl, n_each)
    else:                                                   # no grid: still compare the screened loop
        fake = dict(pl, x0=centre[0] - half, z0=centre[2] - half, cell=max(half / 8, 1e-3), nx=16, nz=16)
        rays = _adversarial_rays(rng, cr, fake, n_each)
    dt = np.float32 if prec == 32 else np.float64
    with np.errstate(over="ignore", invalid="ignore"):
        rays = rays.astype(dt)
    out = {}
    for source in (rt.SCENE_GRID, rt.SCENE_LDS_EXACT):
        with rt.Renderer(0, prec, debug=True) as r:
            r.set_camera(rt.camera(prec, 64, 64, 1, 1)); r.set_scene(sc); r.set_scene_source(source)
            out[source] = r.debug_hit_world(rays)
            if source == rt.SCENE_GRID:
                grid = r.stats()["scene_source"] == rt.SCENE_GRID
    (t, i), (tr, ir) = out[rt.SCENE_GRID], out[rt.SCENE_LDS_EXACT]
    bad = np.nonzero((i != ir) | (t.view(np.uint8).reshape(len(t), -1) != tr.view(np.uint8).reshape(len(t), -1)).any(axis=1))[0]
    used += grid; rays_total += len(rays)
    print("scene %3d f%d n=%3d half=%6.2f r~%.3f grid=%d hits %.2f mismatches %d" % (seed, prec, n, half, rscale, grid, float((ir >= 0).mean()), len(bad)), flush=True)
    if len(bad):
        print(rays[bad[:5]], t[bad[:5]], tr[bad[:5]], i[bad[:5]], ir[bad[:5]])
        sys.exit(1)
print("%d scenes (%d with a grid), %d rays: grid walk == exact loop everywhere" % (n_scenes, used, rays_total))
