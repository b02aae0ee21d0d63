#!/bin/bash
# Instruction cost of the kernel's components by DOUBLE EXECUTION: each -DRTIOW_PROBE_<X> build runs component
# X twice (same image), so SQ_INSTS_VALU(X build) - SQ_INSTS_VALU(plain build) = vector instructions of X per
# launch.  Step 1 (here, no GPU): scripts/cost_probe.sh build.  Step 2 (GPU box): scripts/cost_probe.sh run OUTDIR [one_render args]
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PROBES="PLAIN HIT DIRECT RUV GEN SHADE"
if [ "$1" = build ]; then
  cd "$ROOT/raytracingincuda_amd"; mkdir -p lib/ab
  for v in $PROBES; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -fno-gpu-flush-denormals-to-zero -fno-fast-math -I../include -DRTIOW_PROBE_$v -o lib/ab/probe_$v.so csrc/rtiow_hip.hip csrc/rtiow_group.hip -ldl
  done
  exit 0
fi
OUT=$2; shift 2
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for v in $PROBES; do
  export RTIOW_HIP_LIBRARY="$ROOT/raytracingincuda_amd/lib/ab/probe_$v.so"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d "$OUT/$v" -o pmc --output-format csv -- python3 "$ROOT/scripts/one_render.py" --sched 2 --reps 2 "$@" > "$OUT/$v.log" 2>&1
done
python3 "$ROOT/scripts/pmc_summary.py" $(for v in $PROBES; do echo "$OUT/$v"; done) > "$OUT/summary.jsonl"
cat "$OUT/summary.jsonl"
