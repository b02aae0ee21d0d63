#!/bin/bash
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"; cd "$ROOT"
python3 -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_job3.log" 2>&1 || { tail -40 "$OUT/gpu_tests_job3.log"; exit 1; }
tail -3 "$OUT/gpu_tests_job3.log"
timeout -k 10 300 raytracingincuda_amd/bin/batch_queue_cost > "$OUT/batch_queue_cost.json"
echo "queue probe done"; cat "$OUT/batch_queue_cost.json" | head -c 600; echo
bash scripts/cold_process_study.sh "$OUT/cold_process_study.jsonl" > "$OUT/cold_process_summary.jsonl"
cat "$OUT/cold_process_summary.jsonl"
echo all done
