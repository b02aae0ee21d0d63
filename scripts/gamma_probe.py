import os, subprocess, sys
base = dict(os.environ)
cfgs = [("off", {"RTIOW_DEBUG_TAKE_GAMMA": "0"}),
        ("phi .0,.005,.01,.03,.06", {"RTIOW_DEBUG_TAKE_PHI": "0,0.005,0.01,0.03,0.06"}),
        ("phi .0,.01,.03,.08,.15", {"RTIOW_DEBUG_TAKE_PHI": "0,0.01,0.03,0.08,0.15"}),
        ("phi .0,.0,.02,.06,.12", {"RTIOW_DEBUG_TAKE_PHI": "0,0,0.02,0.06,0.12"}),
        ("phi .0,.0,.0,.05,.10", {"RTIOW_DEBUG_TAKE_PHI": "0,0,0,0.05,0.10"}),
        ("phi .0,.0,.0,.10,.20", {"RTIOW_DEBUG_TAKE_PHI": "0,0,0,0.10,0.20"}),
        ("phi .0,.0,.0,.0,.10", {"RTIOW_DEBUG_TAKE_PHI": "0,0,0,0,0.10"}),
        ("phi .0,.0,.0,.0,.25", {"RTIOW_DEBUG_TAKE_PHI": "0,0,0,0,0.25"}),
        ("off", {"RTIOW_DEBUG_TAKE_GAMMA": "0"})]
for name, e in cfgs:
    env = dict(base, **e)
    for extra, label in (((), "full"), (("--h", "135"), "1920x135")):
        out = subprocess.run([sys.executable, "scripts/one_render.py", "--sched", "2", "--reps", "5", *extra], env=env, capture_output=True, text=True)
        print(name, label, out.stdout.strip()[:48], out.stderr.strip()[-200:], flush=True)
