"""scripts/scaling_probe.py for several library builds on one box, interleaved.
Usage: ab_scaling.py libA.so libB.so"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rd in range(2):
    for l in sys.argv[1:]:
        env = dict(os.environ, RTIOW_HIP_LIBRARY=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "scaling_probe.py")], env=env, capture_output=True, text=True, cwd=root)
        rows = [json.loads(x) for x in out.stdout.splitlines() if x.startswith("{")]
        print(os.path.basename(l), "round", rd, [(r["n_gpus"], r["slowest_rank_ms"]) for r in rows], out.stderr[-200:], flush=True)
