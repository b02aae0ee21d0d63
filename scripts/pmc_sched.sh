#!/bin/bash
# SQ counters of the three schedules on one box: scripts/pmc_sched.sh OUTDIR [one_render args]
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for sched in 0 1 2; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d "$OUT/sched$sched" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched $sched --reps 3 "$@" > "$OUT/sched$sched.log" 2>&1
done
echo done
