"""Sweep of the hand-out knobs of the tuning build (lib/ab/tuning.so, -DRTIOW_TUNING): solo waves for the heaviest
pixels (RTIOW_TUNE_SOLO_WAVES / _SOLO_LANES / _SOLO_PRIO) and the late-pixel protection (RTIOW_TUNE_PROTECT /
_PROTECT_BETA / _PROTECT_MIN).  Cases come from SWEEP_CASES = "name:KEY=V,KEY=V;name:..." (keys without the
RTIOW_TUNE_ prefix; the first case is the reference for the image check).  One box, interleaved rounds; every case
prints the framebuffer's md5, which must not change.  Arguments are passed on to scripts/one_render.py."""
import json, os, re, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "raytracingincuda_amd", "lib", "ab", "tuning.so")
extra = sys.argv[1:]
spec = os.environ.get("SWEEP_CASES", "off:;solo256x2:SOLO_WAVES=256,SOLO_LANES=2;protect:PROTECT=1")
cases = []
for c in spec.split(";"):
    name, _, kv = c.partition(":")
    env, key = {}, None
    for tok in kv.split(","):                      # a token without "=" continues the previous value (comma-separated lists)
        if "=" in tok:
            key = "RTIOW_TUNE_" + tok.split("=")[0]
            env[key] = tok.split("=", 1)[1]
        elif tok and key:
            env[key] += "," + tok
    cases.append((name, env))
times = {c[0]: [] for c in cases}
md5 = {}
for rd in range(2):
    for name, env in cases:
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "one_render.py"), "--sched", "2", "--reps", "6", "--md5", *extra],
                             env=dict(os.environ, RTIOW_HIP_LIBRARY=lib, **env), capture_output=True, text=True)
        if out.returncode != 0:
            print(name, "FAILED", out.stderr[-400:]); sys.exit(1)
        t = [float(x) for x in re.findall(r"[\d.]+", out.stdout.split("]")[0])]
        times[name] += t[1:]
        md5[name] = out.stdout.strip().split("md5 ")[-1]
ref = md5[cases[0][0]]
for name, _ in cases:
    t = np.array(times[name])
    print(json.dumps({"case": name, "args": extra, "ms_median": round(float(np.median(t)), 3), "ms_min": round(float(t.min()), 3), "same_image": md5[name] == ref}), flush=True)
