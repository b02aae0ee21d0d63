"""A/B of two or more builds of librtiow_hip.so on ONE box (boxes differ by ~5 %): interleaved
subprocess runs of scripts/one_render.py; every library must render the same image (md5 of the framebuffer).  Usage: ab_libs.py libA.so libB.so [-- one_render args]"""
import json, os, re, subprocess, sys
import numpy as np
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--"); extra = args[k + 1:]; args = args[:k]
libs = args
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
times = {l: [] for l in libs}
md5 = {l: set() for l in libs}
for rd in range(3):
    for l in libs:
        env = dict(os.environ, RTIOW_HIP_LIBRARY=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "one_render.py"), "--sched", "2", "--reps", "6", "--md5", *extra],
                             env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(l, "FAILED", out.stderr[-400:]); sys.exit(1)
        t = [float(x) for x in re.findall(r"[\d.]+", out.stdout.split("]")[0])]
        times[l] += t[1:]                     # first repetition allocates
        md5[l] |= set(re.findall(r"md5 ([0-9a-f]{32})", out.stdout))
for l in libs:
    t = np.array(times[l])
    print(json.dumps({"lib": os.path.basename(l), "args": extra, "ms_median": round(float(np.median(t)), 3), "ms_min": round(float(t.min()), 3), "n": len(t),
                      "image_md5": sorted(md5[l])}), flush=True)
if len(set(frozenset(v) for v in md5.values())) != 1:
    print("IMAGES DIFFER between the libraries"); sys.exit(2)
