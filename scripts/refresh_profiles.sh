#!/bin/bash
# End-of-round measurement set on ONE GPU box (run through gpurun from the repo root):
#   bench lines (headline fp32, fp64, scene 1), rocprofv3 kernel trace + stats of the SAME bench
#   command, and the two PMC passes for HBM traffic (separate passes, as MI355X_MICROARCH.md asks).
# Results land in gpurun_out/final/; scripts/summarise_profiles.py turns them into profiles/.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_n1.json"
echo "bench fp32 done"
python3 bench.py --steps 10 --warmup 2 --precision 64 --no-cpu-baseline > "$OUT/bench_n1_f64.json"
python3 bench.py --steps 5 --warmup 2 --scene_id 1 --no-cpu-baseline > "$OUT/bench_n1_scene1.json"
python3 bench.py --steps 10 --warmup 2 --schedule static --threads 8 --no-cpu-baseline > "$OUT/bench_n1_static_t8.json"
python3 bench.py --steps 10 --warmup 2 --scene_source lds --no-cpu-baseline > "$OUT/bench_n1_screen_only.json"
python3 bench.py --steps 5 --warmup 2 --scene_id 1 --scene_source lds --no-cpu-baseline > "$OUT/bench_n1_scene1_screen_only.json"
# the other BASELINE.json configs: [1] scene 1 320x192 10 spp 25 bounces, [2] 1280x720, [4] fp64 at 500 spp
python3 bench.py --steps 20 --warmup 3 --scene_id 1 --width 320 --height 192 --samples 10 --bounces 25 --threads 8 --no-cpu-baseline > "$OUT/bench_config2_scene1_320x192.json"
python3 bench.py --steps 20 --warmup 3 --width 1280 --height 720 --threads 8 --no-cpu-baseline > "$OUT/bench_config3_1280x720.json"
python3 bench.py --steps 5 --warmup 1 --precision 64 --samples 500 --no-cpu-baseline > "$OUT/bench_config5_f64_500spp.json"
echo "bench variants done"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/ktrace" -o kt --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/ktrace.log"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch_f64" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 --prec 64 > "$OUT/pmc_fetch_f64.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write_f64" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 --prec 64 > "$OUT/pmc_write_f64.log" 2>&1
echo "pmc done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_sq" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/pmc_sq2" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 > "$OUT/pmc_sq2.log" 2>&1
cd "$ROOT"
if [ -f raytracingincuda_amd/lib/librtiow_hip_stats.so ]; then python3 scripts/path_stats_probe.py > "$OUT/path_stats.json"; python3 scripts/path_stats_probe.py 1 > "$OUT/path_stats_scene1.json"; fi
python3 scripts/scaling_probe.py > "$OUT/scaling_estimate.jsonl"
python3 scripts/accounting_probe.py > "$OUT/accounting.jsonl"
echo "all done"
