"""Sweep of the sorted schedule's prepass length and deal granularity with the tuning build
(lib/ab/tuning.so, -DRTIOW_TUNING: reads RTIOW_TUNE_SA / RTIOW_TUNE_DEAL).  One box, interleaved."""
import json, os, re, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "raytracingincuda_amd", "lib", "ab", "tuning.so")
extra = sys.argv[1:]
cases = [("smooth=%d,SA=%d" % (sm, sa), {"RTIOW_TUNE_SMOOTH": str(sm), "RTIOW_TUNE_SA": str(sa)}) for sm in (4, 6, 8) for sa in (1, 2, 3)] + [("smooth=0,SA=3", {"RTIOW_TUNE_SMOOTH": "0"})]
times = {c[0]: [] for c in cases}
for rd in range(2):
    for name, env in cases:
        env = dict(env)
        wps = ["--wps", env.pop("RTIOW_TUNE_WPS")] if "RTIOW_TUNE_WPS" in env else []
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "one_render.py"), "--sched", "2", "--reps", "6", *extra, *wps],
                             env=dict(os.environ, RTIOW_HIP_LIBRARY=lib, **env), capture_output=True, text=True)
        if out.returncode != 0:
            print(name, "FAILED", out.stderr[-400:]); sys.exit(1)
        t = [float(x) for x in re.findall(r"[\d.]+", out.stdout.split("]")[0])]
        times[name] += t[1:]
for name, _ in cases:
    t = np.array(times[name])
    print(json.dumps({"case": name, "args": extra, "ms_median": round(float(np.median(t)), 3), "ms_min": round(float(t.min()), 3)}), flush=True)
