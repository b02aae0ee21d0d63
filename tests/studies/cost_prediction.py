"""Study behind the sort key of the sorted schedule (not a test, not collected by pytest):
how well does a pixel's prepass cost predict the cost of its remaining samples?  Uses the CPU oracle's
per-pixel segment maps (test infrastructure), so it lives under tests/.

For each scene at 640x360: segments of samples [0,3) ("prepass") and of samples [3,100) ("rest") per pixel;
predictors: the pixel's own prepass cost and box means of it; reported: correlation with the rest, and where in
the heavy-first order the truly heaviest 0.5 % of the pixels would start (fraction of the pixels handed out before).

    python tests/studies/cost_prediction.py > profiles/archive/r02_cost_prediction_study.txt
"""
import os
import sys

import numpy as np
from scipy.ndimage import uniform_filter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import raytracingincuda_amd as rt  # noqa: E402  (host library only: scene tables and camera)
from tests.conftest import compact  # noqa: E402
from tests.oracle_lib import Oracle  # noqa: E402

o = Oracle()
W, H = 640, 360
for scene in (3, 1):
    sc = compact(o.build_scene(scene, 32))
    full = o.render(32, sc, rt.camera(32, W, H, 100, 50), 1227, segments=True)[2].astype(np.float64)
    pre = o.render(32, sc, rt.camera(32, W, H, 3, 50), 1227, segments=True)[2].astype(np.float64)
    rest = full - pre
    n = rest.size
    heavy = np.argsort(-rest.ravel())[: n // 200]
    print("scene %d, %dx%d, 100 spp: segments per pixel mean %.0f, p99 %.0f, max %.0f" % (scene, W, H, full.mean(), np.percentile(full, 99), full.max()))
    for name, pred in (("own 3 samples", pre), ("3x3 mean", uniform_filter(pre, 3)), ("5x5 mean", uniform_filter(pre, 5)),
                       ("9x9 mean", uniform_filter(pre, 9)), ("13x13 mean", uniform_filter(pre, 13)), ("21x21 mean", uniform_filter(pre, 21)),
                       ("the rest itself", rest)):
        order = np.argsort(-pred.ravel(), kind="stable")
        pos = np.empty(n)
        pos[order] = np.arange(n) / n
        print("  %-16s correlation %.3f   heaviest 0.5 %% start at: median %.3f  p90 %.3f  max %.3f" % (
            name, np.corrcoef(pred.ravel(), rest.ravel())[0, 1], np.median(pos[heavy]), np.percentile(pos[heavy], 90), pos[heavy].max()))
