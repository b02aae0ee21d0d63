#!/bin/bash
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"; cd "$ROOT"
A=raytracingincuda_amd/lib/librtiow_hip.so
B=raytracingincuda_amd/lib/ab/tail_prio_by_age.so
C=raytracingincuda_amd/lib/ab/tail_prio_all.so
: > "$OUT/ab_tail_priority.jsonl"
for cfg in "" "--scene 1" "--w 1280 --h 720" "--prec 64"; do
  python3 scripts/ab_libs.py $A $B $C -- $cfg >> "$OUT/ab_tail_priority.jsonl"
done
cat "$OUT/ab_tail_priority.jsonl" | cut -c1-200
echo all done
