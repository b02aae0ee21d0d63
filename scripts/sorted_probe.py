import json, os, subprocess, sys
def run(env_extra, label, extra=()):
    env = dict(os.environ); env.update(env_extra)
    out = subprocess.run([sys.executable, "scripts/one_render.py", "--sched", "2", "--reps", "5", *extra], env=env, capture_output=True, text=True).stdout
    print(label, out.strip()[:70], flush=True)
run({}, "sorted default            ")
run({"RTIOW_DEBUG_NO_SORT": "1"}, "two-phase, tile order (first_pools)")
run({"RTIOW_DEBUG_NO_SORT": "1", "RTIOW_DEBUG_NO_FIRST": "1"}, "two-phase, tile order, atomic first")
run({"RTIOW_DEBUG_NO_FIRST": "1"}, "sorted, atomic first pools")
run({"RTIOW_DEBUG_POOLS_PER_BLOCK": "1"}, "sorted ppb=1")
run({"RTIOW_DEBUG_POOLS_PER_BLOCK": "1", "RTIOW_DEBUG_NO_FIRST": "1"}, "sorted ppb=1 atomic first")
out = subprocess.run([sys.executable, "scripts/one_render.py", "--sched", "1", "--reps", "5"], capture_output=True, text=True).stdout
print("single-phase persistent", out.strip()[:70])
