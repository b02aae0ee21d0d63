"""Interleaved timing of schedule knobs with the tuning build (lib/ab/tuning.so, -DRTIOW_TUNING: launch.h reads RTIOW_TUNE_<KNOB> from the
environment).  Every case must render the same image.  Usage:
    tune_cases.py "PRE_STRIDE=1,SA=3;PRE_STRIDE=2,SA=3;..." [--rounds 3] [-- one_render args]"""
import json, os, re, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("RTIOW_TUNING_LIBRARY") or os.path.join(root, "raytracingincuda_amd", "lib", "ab", "tuning.so")
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--"); extra = args[k + 1:]; args = args[:k]
rounds = 3
if "--rounds" in args:
    k = args.index("--rounds"); rounds = int(args[k + 1]); del args[k:k + 2]
cases = []
for c in args[0].split(";"):
    c = c.strip()
    env = {}
    for kv in c.split(","):
        if kv:
            k, v = kv.split("="); env["RTIOW_TUNE_" + k] = v
    cases.append((c or "builtin", env))
times = {c[0]: [] for c in cases}; stats = {}; md5 = {}
for rd in range(rounds):
    for name, env in cases:
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "one_render.py"), "--sched", "2", "--reps", "6", "--md5", *extra],
                             env=dict(os.environ, RTIOW_HIP_LIBRARY=lib, **env), capture_output=True, text=True)
        if out.returncode != 0:
            print(name, "FAILED", out.stderr[-400:]); sys.exit(1)
        t = [float(x) for x in re.findall(r"[\d.]+", out.stdout.split("]")[0])]
        times[name] += t[1:]
        m = re.search(r"'prepass_ms': ([\d.]+), 'main_ms': ([\d.]+)", out.stdout)
        stats.setdefault(name, []).append((float(m.group(1)), float(m.group(2))))
        md5.setdefault(name, set()).update(re.findall(r"md5 ([0-9a-f]{32})", out.stdout))
for name, _ in cases:
    t = np.array(times[name]); st = np.array(stats[name])
    print(json.dumps({"case": name, "args": extra, "ms_median": round(float(np.median(t)), 3), "ms_min": round(float(t.min()), 3),
                      "prepass_ms": round(float(np.median(st[:, 0])), 3), "main_ms": round(float(np.median(st[:, 1])), 3), "md5": sorted(md5[name])}), flush=True)
if len(set(frozenset(v) for v in md5.values())) != 1:
    print("IMAGES DIFFER between the cases"); sys.exit(2)
