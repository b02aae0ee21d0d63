#!/bin/bash
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"; cd "$ROOT"
python3 -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_job4.log" 2>&1 || { tail -40 "$OUT/gpu_tests_job4.log"; exit 1; }
tail -3 "$OUT/gpu_tests_job4.log"
timeout -k 10 300 raytracingincuda_amd/bin/batch_queue_cost > "$OUT/batch_queue_cost.json"
echo "queue probe done"
bash scripts/cold_process_study.sh "$OUT/cold_process_study.jsonl" > "$OUT/cold_process_summary.jsonl"
cat "$OUT/cold_process_summary.jsonl" | cut -c1-400
python3 scripts/lone_trip_audit.py 3 > "$OUT/lone_trip_audit_scene3.json"
python3 scripts/lone_trip_audit.py 1 > "$OUT/lone_trip_audit_scene1.json"
python3 scripts/lone_trip_audit.py 3 64 > "$OUT/lone_trip_audit_scene3_f64.json"
cat "$OUT/lone_trip_audit_scene3.json"
A=raytracingincuda_amd/lib/librtiow_hip.so
B=raytracingincuda_amd/lib/ab/solo_four_sites.so
: > "$OUT/ab_solo_one_finish.jsonl"
for cfg in "--scene 1 --w 320 --h 192 --s 100 --b 25" "--scene 1 --w 320 --h 192 --s 10 --b 25" "--shard 3,8,2" "--shard 1,4,2" "--w 1 --h 1 --s 400" "--w 640 --h 360" "--scene 1 --w 640 --h 384 --b 25"; do
  python3 scripts/ab_libs.py $A $B -- $cfg >> "$OUT/ab_solo_one_finish.jsonl"
done
cat "$OUT/ab_solo_one_finish.jsonl" | cut -c1-200
C=raytracingincuda_amd/lib/ab/dda_closed_form.so
: > "$OUT/ab_dda_incremental.jsonl"
for cfg in "" "--scene 1" "--prec 64" "--w 3840 --h 2160" "--w 1280 --h 720" "--scene 1 --w 1280 --h 768 --b 25"; do
  python3 scripts/ab_libs.py $A $C -- $cfg >> "$OUT/ab_dda_incremental.jsonl"
done
cat "$OUT/ab_dda_incremental.jsonl" | cut -c1-200
: > "$OUT/pmc_dda_incremental.jsonl"
for cfg in "" "--scene 1"; do python3 scripts/ab_pmc.py $A $C -- $cfg >> "$OUT/pmc_dda_incremental.jsonl"; done
echo all done
