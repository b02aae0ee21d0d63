"""Runs the reference's benchmark grid (global_float_benchmark.sh:5-11: scene 1, five frame sizes,
100 spp, 25 bounces, --threads 4 8 16 32, 5 runs) through tools/hip_benchmark.sh + bin/csv_avg and
tabulates it next to the reference's own published averages (tests/golden/csv/, RTX 3070 Laptop).
Writes profiles/<tag>_harness_*.csv and <tag>_harness_vs_reference.md.   Usage: harness_compare.py [tag]"""
import csv, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(root, "gpurun_out", "harness"); os.makedirs(out, exist_ok=True)
raw = os.path.join(out, "hip_global_float_timing_100sample.csv")
avg = os.path.join(out, "avg_hip_global_float_timing_100sample.csv")
subprocess.check_call(["bash", os.path.join(root, "tools", "hip_benchmark.sh"), "float", raw], stderr=subprocess.DEVNULL)
subprocess.check_call([os.path.join(root, "raytracingincuda_amd", "bin", "csv_avg"), raw, avg])
ref = {}
for r in csv.DictReader(open(os.path.join(root, "tests", "golden", "csv", "250427_avg_gpu_global_float_timing_100sample.csv"))):
    ref[(r["width"], r["height"], r["threads"])] = (float(r["avg_render_only_time_ms"]), float(r["avg_end_to_end_time_ms"]))
lines = ["| frame | --threads | reference render / end-to-end ms (RTX 3070 Laptop) | this build render / end-to-end ms (MI355X) | render speed-up | Mrays/s |",
         "|---|---|---|---|---|---|"]
for r in csv.DictReader(open(avg)):
    k = (r["width"], r["height"], r["threads"])
    mine = (float(r["avg_render_only_time_ms"]), float(r["avg_end_to_end_time_ms"]))
    rays = int(r["width"]) * int(r["height"]) * int(r["samples"])
    rf = ref.get(k)
    lines.append("| %sx%s | %s | %s | %.2f / %.1f | %s | %.0f |" % (
        r["width"], r["height"], r["threads"], "%.1f / %.1f" % rf if rf else "launch fails (csv:122)", mine[0], mine[1],
        "%.0fx" % (rf[0] / mine[0]) if rf else "-", rays / mine[0] / 1e3))
print("\n".join(lines))
open(os.path.join(out, "harness_vs_reference.md"), "w").write("\n".join(lines) + "\n")
