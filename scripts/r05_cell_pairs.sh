#!/bin/bash
# Round 5, VERDICT r04 #1a: cell slots 2-3 behind one wave-level branch (product) against every slot always (lib/ab/cell_all_slots.so).
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"; cd "$ROOT"
A=raytracingincuda_amd/lib/librtiow_hip.so
B=raytracingincuda_amd/lib/ab/cell_all_slots.so
python3 -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_cell_pairs.log" 2>&1 || { tail -30 "$OUT/gpu_tests_cell_pairs.log"; exit 1; }
tail -3 "$OUT/gpu_tests_cell_pairs.log"
: > "$OUT/ab_cell_pairs.jsonl"
for cfg in "" "--scene 1" "--prec 64" "--w 3840 --h 2160" "--w 1280 --h 720" "--scene 1 --w 1280 --h 768 --b 25"; do
  python3 scripts/ab_libs.py $A $B -- $cfg >> "$OUT/ab_cell_pairs.jsonl"
  echo "ab $cfg done"
done
: > "$OUT/pmc_cell_pairs.jsonl"
for cfg in "" "--scene 1" "--prec 64"; do
  python3 scripts/ab_pmc.py $A $B -- $cfg >> "$OUT/pmc_cell_pairs.jsonl"
  echo "pmc $cfg done"
done
python3 scripts/path_stats_probe.py > "$OUT/path_stats.json"
python3 scripts/path_stats_probe.py 1 > "$OUT/path_stats_scene1.json"
echo all done
