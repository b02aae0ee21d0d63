#!/bin/bash
# PMC comparison of library builds on one box: scripts/pmc_ab.sh OUTDIR lib1.so lib2.so ...
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so)
  export RTIOW_HIP_LIBRARY="$ROOT/$lib"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/$name.a" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 > "$OUT/$name.a.log" 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_ANY -d "$OUT/$name.b" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 3 > "$OUT/$name.b.log" 2>&1 || echo "pass b failed for $name"
done
echo done
