"""Regenerates the fixtures under tests/golden/.  Run in the build container (needs
/root/reference for the serial pins and hipcc for the rocRAND known answers):

    python tests/golden/make_golden.py [--full]

What each fixture is and where it comes from:
  serial_ref.json          md5/length of the P3 text printed by the REFERENCE's serial tracer
                           (oracle/_ref/ref_serial_driver = the reference's own headers built by
                           oracle/Makefile) for small configs; --full also re-runs the unmodified
                           reference binary (oracle/_ref/inOneWeekend, ~55 s).
  xorwow_rocrand_kat.json  rocRAND host-engine XORWOW known answers (make_xorwow_kat.cpp).
  scene_tables.npz         scene tables 1/2/3 x f32/f64 from the oracle (the f64 tables are what
                           the byte-exact serial pin renders; three f32 entries are also listed in
                           SURVEY.md Appendix A.3, dumped there from the reference's structs).
  cuda_sem_*.npy           small CUDA-semantics renders from the oracle (regression + the
                           golden the -m gpu tests compare the HIP path with).
  ref_serial_320x192_100spp_50b.npz
                           the images (uint8 levels of the P3 text) the REFERENCE's serial tracer
                           (oracle/_ref/ref_serial_driver) prints for scenes 1/2/3 at 320x192, 100 spp,
                           depth 50, + ref_serial_320x192_100spp_50b.json: md5 of each P3 text and the
                           ppm_diff noise floor of each scene (two oracle renders that differ only
                           in the RNG seed; SURVEY.md A.5 lists the same floors for scenes 1 and 3).
                           Used by the sky-mode / policy pins of the CUDA-semantics oracle.
No reference source text is stored here: only inputs (flags) and outputs (hashes, arrays).
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.oracle_lib import Oracle  # noqa: E402

SERIAL_CONFIGS = [  # scene_id, W, H, S, depth
    (3, 64, 36, 4, 10), (1, 96, 54, 3, 12), (2, 80, 48, 5, 25), (3, 160, 96, 10, 50), (1, 120, 72, 2, 20),
]
CUDA_SEM_CONFIGS = [  # name, precision, scene_id, W, H, S, B
    ("cuda_sem_s3_64x40_4spp_10b_f32", 32, 3, 64, 40, 4, 10),
    ("cuda_sem_s1_48x32_2spp_8b_f32", 32, 1, 48, 32, 2, 8),
    ("cuda_sem_s2_48x32_3spp_25b_f64", 64, 2, 48, 32, 3, 25),
]


def main():
    full = "--full" in sys.argv
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_serial_driver")
    serial = {"driver": []}
    for cfg in SERIAL_CONFIGS:
        out = subprocess.run([drv] + [str(x) for x in cfg], capture_output=True, check=True).stdout
        serial["driver"].append({"scene_id": cfg[0], "width": cfg[1], "height": cfg[2], "samples": cfg[3], "depth": cfg[4],
                                 "md5": hashlib.md5(out).hexdigest(), "bytes": len(out)})
    path = os.path.join(HERE, "serial_ref.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    serial["unmodified_binary"] = old.get("unmodified_binary")
    if full:
        out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "inOneWeekend")], capture_output=True, check=True).stdout
        serial["unmodified_binary"] = {"scene_id": 1, "width": 1280, "height": 768, "samples": 10, "depth": 20,
                                       "md5": hashlib.md5(out).hexdigest(), "bytes": len(out)}
    json.dump(serial, open(path, "w"), indent=1)

    from tests.oracle_lib import p3_levels, to_levels, diff_stats, SKY_CURRENT
    orc = Oracle()
    import raytracingincuda_amd as rt
    imgs, meta = {}, {"config": {"width": 320, "height": 192, "samples": 100, "depth": 50}, "scenes": {}}
    for sid in (1, 2, 3):
        out = subprocess.run([drv, str(sid), "320", "192", "100", "50"], capture_output=True, check=True).stdout
        imgs["s%d" % sid] = p3_levels(out)
        sc = orc.build_scene(sid, 64)
        keep = sc["valid"] != 0
        comp = {k: (v[keep] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
        cam = rt.camera(64, 320, 192, 100, 50)
        a, _ = orc.render(64, comp, cam, 1227, sky_mode=SKY_CURRENT)
        b, _ = orc.render(64, comp, cam, 99, sky_mode=SKY_CURRENT)
        floor = diff_stats(to_levels(a), to_levels(b))
        fb = diff_stats(to_levels(a)[..., 2], to_levels(b)[..., 2])
        meta["scenes"][str(sid)] = {"md5": hashlib.md5(out).hexdigest(), "bytes": len(out),
                                    "floor_mean": floor["mean"], "floor_p99": floor["p99"], "floor_max": floor["max"],
                                    "floor_mean_blue": fb["mean"], "floor_p99_blue": fb["p99"]}
    np.savez_compressed(os.path.join(HERE, "ref_serial_320x192_100spp_50b.npz"), **imgs)
    json.dump(meta, open(os.path.join(HERE, "ref_serial_320x192_100spp_50b.json"), "w"), indent=1)

    kat_exe = "/tmp/rtiow_make_xorwow_kat"
    subprocess.run(["hipcc", "-O1", "-o", kat_exe, os.path.join(HERE, "make_xorwow_kat.cpp")], check=True)
    kat = subprocess.run([kat_exe], capture_output=True, check=True).stdout
    open(os.path.join(HERE, "xorwow_rocrand_kat.json"), "wb").write(kat)

    tables = {}
    for prec in (32, 64):
        for sid in (1, 2, 3):
            sc = orc.build_scene(sid, prec)
            for k in ("center_radius", "albedo_fuzz", "refraction_index", "type", "valid"):
                tables["s%d_f%d_%s" % (sid, prec, k)] = sc[k]
    np.savez_compressed(os.path.join(HERE, "scene_tables.npz"), **tables)

    for name, prec, sid, W, H, S, B in CUDA_SEM_CONFIGS:
        sc = orc.build_scene(sid, prec)
        keep = sc["valid"] != 0
        comp = {k: (v[keep] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
        img, _ = orc.render(prec, comp, rt.camera(prec, W, H, S, B), 1227)
        np.save(os.path.join(HERE, name + ".npy"), img)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
