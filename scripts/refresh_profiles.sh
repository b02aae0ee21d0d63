#!/bin/bash
# End-of-round measurement set on ONE GPU box (run through gpurun from the repo root):
#   bench lines of every BASELINE.json configuration (each runs its own rocprofv3 --pmc passes: roofline.frac is the
#   vector-issue fraction of THAT run's library), the rocprofv3 kernel trace + stats of the headline bench command,
#   the committed counter records (profiles/pmc_records.json, keyed by configuration, tied to the build id),
#   the in-library group at N = 2 on one device, the execution profile and the single-GPU scaling estimate.
# Results land in gpurun_out/final/; scripts/summarise_profiles.py rNN turns them into profiles/.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_n1.json"
echo "bench fp32 done"
python3 bench.py --steps 10 --warmup 2 --precision 64 --no-cpu-baseline --no-extra-configs > "$OUT/bench_n1_f64.json"
python3 bench.py --steps 5 --warmup 2 --scene_id 1 --no-cpu-baseline --no-extra-configs > "$OUT/bench_n1_scene1.json"
python3 bench.py --steps 10 --warmup 2 --schedule static --threads 8 --no-cpu-baseline --no-extra-configs > "$OUT/bench_n1_static_t8.json"
python3 bench.py --steps 10 --warmup 2 --scene_source lds --no-cpu-baseline --no-extra-configs --pmc off > "$OUT/bench_n1_screen_only.json"
# the other BASELINE.json configs: [1] scene 1 320x192 10 spp 25 bounces, [2] 1280x720, [4] fp64 at 500 spp
python3 bench.py --steps 20 --warmup 3 --scene_id 1 --width 320 --height 192 --samples 10 --bounces 25 --threads 8 --no-cpu-baseline --no-extra-configs > "$OUT/bench_config2_scene1_320x192.json"
python3 bench.py --steps 20 --warmup 3 --width 1280 --height 720 --threads 8 --no-cpu-baseline --no-extra-configs > "$OUT/bench_config3_1280x720.json"
python3 bench.py --steps 5 --warmup 1 --precision 64 --samples 500 --no-cpu-baseline --no-extra-configs > "$OUT/bench_config5_f64_500spp.json"
python3 bench.py --steps 5 --warmup 1 --width 3840 --height 2160 --no-cpu-baseline --no-extra-configs > "$OUT/bench_3840x2160.json"
# under torch.distributed.run at world size 1 (N = 2 without a launcher follows the per-shard counter records below)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > "$OUT/bench_n1_rccl_world_size_1.json"
echo "bench variants done"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/ktrace" -o kt --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-scaling-probe --pmc off > "$OUT/bench_under_rocprof.json" 2> "$OUT/ktrace.log"
echo "kernel trace done"
cd "$ROOT"
for cfg in "" "--scene_id 1" "--width 1280 --height 720" "--scene_id 1 --width 320 --height 192 --samples 10 --bounces 25 --threads 8"; do
  python3 scripts/pmc_passes.py --out "$OUT/pmc_records.json" $cfg > /dev/null
done
for cfg in "--precision 64" "--precision 64 --samples 500"; do      # fp64: + the executed double-precision instruction counts
  python3 scripts/pmc_passes.py --out "$OUT/pmc_records.json" --passes sq,fetch,write,f64 $cfg > /dev/null
done
# every rank's shard of the headline frame for N = 2 / 4 / 8 (one sq pass each): what the N > 1 contract line rates each rank's launch against
python3 scripts/pmc_shard_records.py --out "$OUT/pmc_records.json" --ns 2,4,8 > "$OUT/pmc_shard_records.log"
echo "pmc records done"
# N = 2 / 4 without a launcher (the in-library group; the ranks share the one device), each rank rated against ITS shard's record
RTIOW_PMC_RECORDS="$OUT/pmc_records.json" python3 bench.py --gpus 2 --devices 0,0 --steps 10 --warmup 2 > "$OUT/bench_group_n2_one_device.json"
RTIOW_PMC_RECORDS="$OUT/pmc_records.json" python3 bench.py --gpus 4 --devices 0,0,0,0 --steps 5 --warmup 1 > "$OUT/bench_group_n4_one_device.json"
if [ -f raytracingincuda_amd/lib/librtiow_hip_stats.so ]; then python3 scripts/path_stats_probe.py > "$OUT/path_stats.json"; python3 scripts/path_stats_probe.py 1 > "$OUT/path_stats_scene1.json"; fi
python3 scripts/scaling_probe.py > "$OUT/scaling_estimate.jsonl"
# the drop-in executables: the reference's own benchmark grid and the BASELINE.json configurations (one cold process per run), text and binary files
python3 scripts/harness_compare.py final > "$OUT/harness_compare.log" 2>&1 || true
cp gpurun_out/harness/*.csv gpurun_out/harness/harness_vs_reference.md "$OUT/" 2>/dev/null || true
BASELINE_CONFIGS=1 RUNS=3 STATS_JSONL="$OUT/harness_baseline_configs_stats.jsonl" bash tools/hip_benchmark.sh float "$OUT/harness_baseline_configs_float.csv" 2> /dev/null
BASELINE_CONFIGS=1 RUNS=2 STATS_JSONL="$OUT/harness_baseline_configs_stats_double.jsonl" bash tools/hip_benchmark.sh double "$OUT/harness_baseline_configs_double.csv" 2> /dev/null
mkdir -p /tmp/e2e_p6 && (cd /tmp/e2e_p6 && for i in 1 2 3; do "$ROOT/raytracingincuda_amd/bin/global-float-hip-raytrace" --scene_id 3 --width 1920 --height 1080 --samples 100 --bounces 50 --threads 8 --stats --ppm_format p6; done) > "$OUT/e2e_1080p_p6.log" 2>&1
echo "harness done"
python3 scripts/accounting_probe.py > "$OUT/accounting.jsonl"
python3 scripts/lone_trip_audit.py 3 > "$OUT/lone_trip_audit_scene3.json"
python3 scripts/lone_trip_audit.py 1 > "$OUT/lone_trip_audit_scene1.json"
timeout -k 10 300 raytracingincuda_amd/bin/batch_queue_cost > "$OUT/batch_queue_cost.json" || true
echo "all done"
