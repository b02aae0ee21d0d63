#!/bin/bash
# Instruction counts (VALU, SALU, branches) of library builds on one box: scripts/pmc_count.sh OUTDIR lib1.so lib2.so ... [-- one_render args]
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for lib in "${LIBS[@]}"; do
  name=$(basename "$lib" .so)
  export RTIOW_HIP_LIBRARY="$ROOT/$lib"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD -d "$OUT/$name" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 2 "$@" > "$OUT/$name.log" 2>&1
done
python3 "$ROOT/scripts/pmc_summary.py" $(for lib in "${LIBS[@]}"; do echo "$OUT/$(basename "$lib" .so)"; done)
