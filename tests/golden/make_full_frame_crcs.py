"""Full-frame goldens for the BASELINE.json configurations, from the CPU oracle.

    python tests/golden/make_full_frame_crcs.py [name ...]      # all configs, or the named ones

The oracle (oracle/rtiow_oracle.cpp, CUDA policy: iterative loop, sky from the primary ray,
XORWOW per global pixel index) renders every pixel of the frame on all host threads; what is
committed is one CRC-32 per image row of the raw float bits plus the SHA-256 of the whole
image (tests/golden/full_frame_crcs.json, ~10 bytes per row).  The -m gpu test
`test_full_frames_match_the_oracle_row_by_row` renders the same frames with the HIP path and
compares every row, so the BASELINE sizes are checked bit for bit in full, not by spot rows.
Minutes of CPU time per frame (the headline frame is ~5e8 path segments x 125 spheres).
"""
import hashlib
import json
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.oracle_lib import Oracle  # noqa: E402
from tests.conftest import compact  # noqa: E402

CONFIGS = [  # name, precision, scene_id, W, H, S, B
    ("config2_scene3_1280x720_100spp_50b_f32", 32, 3, 1280, 720, 100, 50),
    ("config3_headline_scene3_1920x1080_100spp_50b_f32", 32, 3, 1920, 1080, 100, 50),
    ("scene3_1920x1080_100spp_50b_f64", 64, 3, 1920, 1080, 100, 50),
    ("scene1_1920x1080_100spp_50b_f32", 32, 1, 1920, 1080, 100, 50),
    ("scene1_1280x768_100spp_25b_f32", 32, 1, 1280, 768, 100, 25),      # largest frame of the reference's benchmark grid
    ("config4_scene3_1920x1080_500spp_50b_f64", 64, 3, 1920, 1080, 500, 50),
]
OUT = os.path.join(HERE, "full_frame_crcs.json")


def row_crcs(img):
    return [zlib.crc32(np.ascontiguousarray(img[j]).view(np.uint8).tobytes()) & 0xffffffff for j in range(img.shape[0])]


def main():
    import raytracingincuda_amd as rt     # host library only: camera::initialize in T (no GPU needed)
    want = set(sys.argv[1:])
    gold = json.load(open(OUT)) if os.path.exists(OUT) else {}
    orc = Oracle()
    orc.set_threads(os.cpu_count() or 1)
    for name, prec, scene_id, W, H, S, B in CONFIGS:
        if want and name not in want:
            continue
        t0 = time.time()
        scene = compact(orc.build_scene(scene_id, prec))
        img, stats = orc.render(prec, scene, rt.camera(prec, W, H, S, B), 1227)
        gold[name] = {"precision": prec, "scene_id": scene_id, "width": W, "height": H, "samples": S, "bounces": B, "seed": 1227,
                      "sha256": hashlib.sha256(np.ascontiguousarray(img).view(np.uint8).tobytes()).hexdigest(),
                      "row_crc32": row_crcs(img), "oracle_stats": stats, "oracle_seconds": round(time.time() - t0, 1),
                      "oracle_threads": os.cpu_count()}
        print(name, gold[name]["sha256"][:16], "%.0f s" % (time.time() - t0), flush=True)
        with open(OUT, "w") as f:
            json.dump(gold, f, indent=0, separators=(",", ":"))
            f.write("\n")


if __name__ == "__main__":
    main()
