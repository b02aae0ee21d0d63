#!/bin/bash
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"; cd "$ROOT"
A=raytracingincuda_amd/lib/librtiow_hip.so
B=raytracingincuda_amd/lib/ab/cell_all_slots.so
C=raytracingincuda_amd/lib/ab/cell_pairs_joint.so
: > "$OUT/ab_cell_pairs2.jsonl"
for cfg in "" "--scene 1" "--prec 64" "--w 3840 --h 2160" "--w 1280 --h 720" "--scene 1 --w 1280 --h 768 --b 25" "--prec 64 --s 500"; do
  python3 scripts/ab_libs.py $A $B $C -- $cfg >> "$OUT/ab_cell_pairs2.jsonl"
  echo "ab $cfg done"
done
: > "$OUT/pmc_cell_pairs2.jsonl"
for cfg in "" "--scene 1" "--prec 64"; do
  python3 scripts/ab_pmc.py $A $B -- $cfg >> "$OUT/pmc_cell_pairs2.jsonl"
  echo "pmc $cfg done"
done
python3 scripts/path_stats_probe.py > "$OUT/path_stats.json"
python3 scripts/path_stats_probe.py 1 > "$OUT/path_stats_scene1.json"
python3 -m pytest tests/test_group.py tests/test_gpu_parity.py -m gpu -x -q -k "bench or launcher" > "$OUT/gpu_tests_bench.log" 2>&1 || { tail -40 "$OUT/gpu_tests_bench.log"; exit 1; }
tail -3 "$OUT/gpu_tests_bench.log"
python3 bench.py --steps 10 --warmup 2 > "$OUT/bench_try1.json" 2> "$OUT/bench_try1.err"
python3 scripts/pmc_shard_records.py --out "$OUT/pmc_records_try.json" --ns 2 > "$OUT/shard_records_try.log"
RTIOW_PMC_RECORDS="$OUT/pmc_records_try.json" python3 bench.py --gpus 2 --devices 0,0 --steps 5 --warmup 1 > "$OUT/bench_group_try1.json"
echo all done
