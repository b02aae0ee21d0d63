import os, subprocess, sys
for g in ("0", "1"):
    env = dict(os.environ, RTIOW_DEBUG_TAKE_GAMMA=g)
    out = subprocess.run([sys.executable, "scripts/accounting_probe.py"], env=env, capture_output=True, text=True)
    print("gamma", g); print(out.stdout, out.stderr[-300:], flush=True)
