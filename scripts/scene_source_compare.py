"""The AMD analogue of the reference's global / const / tex comparison (README.md:7-12): the same
frame with every scene source -- uniform grid over the LDS tables (default), screened loop over the LDS tables,
exact loop over the LDS tables, wave-uniform scalar loads through the scalar cache -- side by side, fp32 and fp64.
Every source gives the same image (the md5 column).     python scripts/scene_source_compare.py > profiles/archive/r02_scene_source_comparison.md"""
import hashlib, sys
import numpy as np
sys.path.insert(0, '.')
import raytracingincuda_amd as rt
SOURCES = [(rt.SCENE_GRID, "grid (default)"), (rt.SCENE_LDS, "lds: screened loop"), (rt.SCENE_LDS_EXACT, "lds_exact: exact loop"), (rt.SCENE_SCALAR, "scalar: exact loop, scalar cache")]
print("| frame | precision | scene source | kernel ms (median of 5) | Grays/s | LDS bytes / workgroup | image md5 |")
print("|---|---|---|---|---|---|---|")
for scene, W, H, S, B in ((3, 1920, 1080, 100, 50), (1, 1920, 1080, 100, 50), (1, 1280, 768, 100, 25)):
    for prec in (32, 64):
        sc = rt.build_scene(scene, prec); cam = rt.camera(prec, W, H, S, B)
        for src, name in SOURCES:
            with rt.Renderer(0, prec) as r:
                r.set_camera(cam); r.set_scene(sc); r.set_scene_source(src); r.init_rng(1227)
                r.render(0)
                ms = float(np.median([r.render(0) for _ in range(5)]))
                st = r.stats()
                md5 = hashlib.md5(r.read_framebuffer().tobytes()).hexdigest()[:12]
            print("| scene %d %dx%d %d spp %d b | fp%d | %s | %.2f | %.2f | %d | %s |" % (scene, W, H, S, B, prec, name, ms, W * H * S / ms / 1e6, st["lds_bytes"], md5), flush=True)
