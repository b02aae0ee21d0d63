"""Soak run for the hand-out (not a test; run by hand on a GPU box): random frame sizes, sample / bounce counts,
scenes, precisions and shards through the default sorted schedule (prepass, cost sort, main launch; solo waves on a
partly filled GPU) against the static schedule (one lane per pixel, no hand-out), bit for bit.  A hand-out bug --
a slot skipped, handed out twice, a state parked or unparked wrongly -- changes the image.  Prints one line per case;
exits non-zero on the first mismatch.      python tests/studies/schedule_soak.py [n_cases] [--large]
--large: frames of 1.4-3 M pixels, unsharded -- at least four 64-pixel pools per resident wave, where the fp32 launches take the kernels with
the bounded rejection loop (render_persistent_kernel<float, SRC, COUNT, true>; launch_render)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import raytracingincuda_amd as rt  # noqa: E402

large = "--large" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_cases = int(args[0]) if args else 60
rng = np.random.default_rng(77)
solo_cases = plain_cases = unsorted_cases = 0
for case in range(n_cases):
    prec = 32 if case % 3 else 64
    scene = int(rng.choice([1, 2, 3]))
    W = int(rng.integers(33, 1400)); H = int(rng.integers(17, 800))
    S = int(rng.choice([3, 23, 24, 25, 40, 64, 70])); B = int(rng.choice([1, 5, 25, 32, 33, 50]))
    if large:
        W = int(rng.integers(1500, 2400)); H = int(rng.integers(950, 1300)); S = int(rng.choice([24, 25, 40]))
    if W * H * S > (1.3e8 if large else 6e7):
        S = max(24, int((1.3e8 if large else 6e7) / (W * H)))
    shard = None
    if case % 2 and not large:
        n = int(rng.integers(2, 9))
        shard = (int(rng.integers(0, n)), n, int(rng.choice([1, 2, 3, 8])))
    sc = rt.build_scene(scene, prec)
    images, stats = [], None
    for sched in (rt.SCHED_SORTED, rt.SCHED_STATIC):
        with rt.Renderer(0, prec) as r:
            r.set_camera(rt.camera(prec, W, H, S, B)); r.set_scene(sc); r.set_schedule(sched, 0)
            if shard:
                r.set_shard(*shard)
            r.init_rng(1227)
            r.render(8)
            images.append(r.read_framebuffer())
            if sched == rt.SCHED_SORTED:
                stats = r.stats()
    same = images[0].shape == images[1].shape and np.array_equal(images[0].view(np.uint8), images[1].view(np.uint8))
    solo_cases += stats["solo_waves"] > 0; plain_cases += stats["phases"] == 2 and stats["solo_waves"] == 0; unsorted_cases += stats["phases"] == 1
    print("case %3d fp%d scene %d %4dx%-4d %3d spp %2d bounces shard %-12s phases %d solo %3d x %d  %s" % (
        case, prec, scene, W, H, S, B, shard, stats["phases"], stats["solo_waves"], stats["solo_lanes"], "same" if same else "DIFFERENT"), flush=True)
    if not same:
        sys.exit(1)
print("%d cases, all bit-identical to the static schedule: %d with solo waves, %d sorted without, %d unsorted (frame or sample count too small)" % (
    n_cases, solo_cases, plain_cases, unsorted_cases))
