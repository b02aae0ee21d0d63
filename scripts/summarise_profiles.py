"""Turns gpurun_out/final/ (scripts/refresh_profiles.sh) into the committed files under profiles/:
bench lines, the rocprofv3 kernel-stats table, the per-launch agreement check between rocprofv3 and bench.py's HIP
events, and the counter records (profiles/pmc_records.json: what `bench.py --pmc committed` reads -- each record
carries the build id of the library it was measured on).
Usage: summarise_profiles.py [round_tag]   (default r05; files land in profiles/<round_tag>/)"""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "final")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
top = os.path.join(root, "profiles")
dst = os.path.join(top, tag)                      # one directory per round; profiles/pmc_records.json (what bench.py --pmc committed reads) stays at the top
os.makedirs(dst, exist_ok=True)

def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])

for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    name = os.path.basename(f)[:-5]
    try:
        json.dump(last_json(f), open(os.path.join(dst, "%s.json" % name), "w"), indent=1)
    except (ValueError, IndexError):
        print("skipped (no JSON line):", name)
stats = glob.glob(os.path.join(src, "ktrace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
b = last_json(os.path.join(src, "bench_under_rocprof.json"))
rows = list(csv.DictReader(open(stats)))
def row(part):
    hit = [r for r in rows if part in r["Name"]]
    return hit[0] if hit else None
# (the fp32 kernels exist with and without the bounded rejection loop -- fourth template argument -- and a launch uses one of them)
main, pre, place = row("render_persistent_kernel<float, 0, false"), row("render_prepass_kernel<float, 0, false"), row("place_pixels_kernel<float>")
agree = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-probe --pmc off",
         "rocprof_main_launch_avg_ms": float(main["AverageNs"]) / 1e6, "rocprof_main_launch_calls": int(main["Calls"]),
         "bench_roofline_launch_ms_mean": b["roofline"]["launch_ms_mean"],
         "rocprof_prepass_avg_ms": float(pre["AverageNs"]) / 1e6 if pre else None, "bench_prepass_ms": b["step"]["prepass_ms"],
         "rocprof_place_pixels_avg_ms": float(place["AverageNs"]) / 1e6 if place else None, "bench_place_ms": b["step"].get("place_ms"),
         "rocprof_all_kernels_per_step_ms": (float(main["AverageNs"]) + (float(pre["AverageNs"]) if pre else 0.0) + (float(place["AverageNs"]) if place else 0.0)
                                             + sum(float(r["AverageNs"]) for r in rows if "cost_" in r["Name"])) / 1e6,
         "bench_step_kernel_ms_mean": b["step"]["kernel_ms_mean"]}
json.dump(agree, open(os.path.join(dst, "rocprof_vs_bench.json"), "w"), indent=1)
print(json.dumps(agree, indent=1))
if os.path.exists(os.path.join(src, "pmc_records.json")):
    rec = json.load(open(os.path.join(src, "pmc_records.json")))
    json.dump(rec, open(os.path.join(top, "pmc_records.json"), "w"), indent=1, sort_keys=True)
    for k, r in rec.items():
        print(k, r.get("build_id", "")[:12], json.dumps(r.get("derived_main")))
for name in ("lone_trip_audit_scene3.json", "lone_trip_audit_scene1.json", "batch_queue_cost.json", "pmc_shard_records.log"):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
for name, out in (("path_stats.json", "path_stats.json"), ("path_stats_scene1.json", "path_stats_scene1.json"), ("scaling_estimate.jsonl", "scaling_estimate_single_gpu.jsonl"), ("accounting.jsonl", "accounting_by_age_class.jsonl")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, out))
for f in sorted(glob.glob(os.path.join(src, "harness_*")) + glob.glob(os.path.join(src, "*timing_100sample.csv")) + glob.glob(os.path.join(src, "e2e_*.log"))):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
