#!/bin/bash
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 300 raytracingincuda_amd/bin/batch_queue_cost > "$OUT/batch_queue_cost.json"
python3 scripts/lone_trip_counters.py 3 > "$OUT/lone_trip_counters_scene3.json"
python3 scripts/lone_trip_counters.py 1 > "$OUT/lone_trip_counters_scene1.json"
cat "$OUT/lone_trip_counters_scene3.json"
echo all done
