"""Turns gpurun_out/final/ (scripts/refresh_profiles.sh) into the committed files under profiles/:
bench lines, the rocprofv3 kernel-stats table, per-launch agreement check, HBM traffic, SQ counters.
Usage: summarise_profiles.py [round_tag]   (default r01)"""
import csv, glob, json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "final")
dst = os.path.join(root, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])

for name, out in (("bench_n1.json", "bench_n1"), ("bench_n1_f64.json", "bench_n1_f64"), ("bench_n1_scene1.json", "bench_n1_scene1"),
                  ("bench_n1_static_t8.json", "bench_n1_static_t8"), ("bench_under_rocprof.json", "bench_under_rocprof"),
                  ("bench_n1_screen_only.json", "bench_n1_screen_only"), ("bench_n1_scene1_screen_only.json", "bench_n1_scene1_screen_only"),
                  ("bench_config2_scene1_320x192.json", "bench_config2_scene1_320x192"), ("bench_config3_1280x720.json", "bench_config3_1280x720"),
                  ("bench_config5_f64_500spp.json", "bench_config5_f64_500spp")):
    if not os.path.exists(os.path.join(src, name)):
        continue
    json.dump(last_json(os.path.join(src, name)), open(os.path.join(dst, "%s_%s.json" % (tag, out)), "w"), indent=1)
stats = glob.glob(os.path.join(src, "ktrace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
b = last_json(os.path.join(src, "bench_under_rocprof.json"))
rows = list(csv.DictReader(open(stats)))
main = [r for r in rows if "render_persistent_kernel<float, 0, false>" in r["Name"]][0]
pre = [r for r in rows if "render_prepass_kernel<float, 0, false>" in r["Name"]]
agree = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-probe",
         "rocprof_main_launch_avg_ms": float(main["AverageNs"]) / 1e6, "rocprof_main_launch_calls": int(main["Calls"]),
         "bench_roofline_launch_ms_mean": b["roofline"]["launch_ms_mean"],
         "rocprof_prepass_avg_ms": float(pre[0]["AverageNs"]) / 1e6 if pre else None, "bench_prepass_ms": b["step"]["prepass_ms"],
         "rocprof_all_kernels_per_step_ms": (float(main["AverageNs"]) + (float(pre[0]["AverageNs"]) if pre else 0.0)
                                             + sum(float(r["AverageNs"]) for r in rows if "cost_" in r["Name"])) / 1e6,
         "bench_step_kernel_ms_mean": b["step"]["kernel_ms_mean"]}
json.dump(agree, open(os.path.join(dst, "%s_rocprof_vs_bench.json" % tag), "w"), indent=1)
print(json.dumps(agree, indent=1))
for prec, suffix in ((32, ""), (64, "_f64")):
    subprocess.check_call([sys.executable, os.path.join(root, "scripts", "collect_traffic.py"), os.path.join(src, "pmc_fetch" + suffix),
                           os.path.join(src, "pmc_write" + suffix), "s3_1920x1080_100spp_50b_f%d" % prec])
sq = {}
for d in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(src, d, "**", "*_counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            k = "main" if "render_persistent_kernel" in r["Kernel_Name"] else ("prepass" if "render_prepass_kernel" in r["Kernel_Name"] else None)
            if k: per.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in per.items():
            sq.setdefault(k, {})[c] = sum(v) / len(v)
# what the counters say about issue: vector instructions per launch and SIMD cycles per instruction
m = sq.get("main", {})
if m.get("SQ_INSTS_VALU") and m.get("GRBM_GUI_ACTIVE"):
    simd_cycles = m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0             # GRBM_GUI_ACTIVE sums the 8 XCDs; 256 CUs x 4 SIMDs
    sq["derived_main"] = {"valu_insts_per_launch": m["SQ_INSTS_VALU"], "simd_cycles_per_valu_inst": simd_cycles / m["SQ_INSTS_VALU"],
                          "lane_insts_per_launch": m["SQ_INSTS_VALU"] * 64.0,
                          "note": "SQ_INSTS_VALU counts wave-instructions; a wave64 VALU op occupies a SIMD-32 for >= 2 cycles"}
json.dump({"config": "scene 3 1920x1080 100spp 50b fp32, sorted schedule; mean per dispatch", "counters": sq},
          open(os.path.join(dst, "%s_pmc_sq_final.json" % tag), "w"), indent=1)
print(json.dumps(sq.get("main", {}), indent=1))

for name, out in (("path_stats.json", "path_stats.json"), ("path_stats_scene1.json", "path_stats_scene1.json"), ("scaling_estimate.jsonl", "scaling_estimate_single_gpu.jsonl"), ("accounting.jsonl", "accounting_by_age_class.jsonl")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, out)))
