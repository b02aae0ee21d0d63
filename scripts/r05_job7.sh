#!/bin/bash
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"; cd "$ROOT"
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_frames or oracle or degenerate or smoke or parity" > "$OUT/gpu_tests_job7.log" 2>&1 || { tail -40 "$OUT/gpu_tests_job7.log"; exit 1; }
tail -2 "$OUT/gpu_tests_job7.log"
A=raytracingincuda_amd/lib/librtiow_hip.so
B=raytracingincuda_amd/lib/ab/camera_two_fetches.so
: > "$OUT/ab_camera_one_fetch.jsonl"
for cfg in "--w 1 --h 1 --s 400" "--shard 3,8,2" "--shard 1,4,2" "--scene 1 --w 320 --h 192 --s 100 --b 25" "--scene 1 --w 320 --h 192 --s 10 --b 25" "" "--prec 64" "--w 1280 --h 720"; do
  python3 scripts/ab_libs.py $A $B -- $cfg >> "$OUT/ab_camera_one_fetch.jsonl"
done
cat "$OUT/ab_camera_one_fetch.jsonl" | cut -c1-200
echo all done
