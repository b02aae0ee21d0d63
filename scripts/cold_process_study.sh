#!/bin/bash
# why do some cold-process renders take 16-19 ms where the warm bench takes 11.3?  One process per run (the reference's CSV
# contract, global_float_benchmark.sh:53-74), the drop-in executable with --stats: render_ms, the three launches' own event times, and the
# EFFECTIVE shader clock of the prepass and of the main launch (rtiow_stats.main_clock_mhz: s_memtime / s_memrealtime stamps of one wave).
#   series A: 10 processes back to back (binary P6 output: ~5 ms between the end of one render and the start of the next process)
#   series B: 10 processes with 2 s of idle GPU between them
#   series C: 10 processes back to back, each rendering TWICE in the process (RTIOW_RENDER_TWICE=1: the first render is the warm-up) -- what a
#             warm clock and loaded code objects give the same process
# Output: one JSON line per run in $1 (default gpurun_out/r05/cold_process_study.jsonl).
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/r05/cold_process_study.jsonl}
EXE=$ROOT/raytracingincuda_amd/bin/global-float-hip-raytrace
ARGS="--scene_id 3 --width 1920 --height 1080 --samples 100 --bounces 50 --threads 8 --stats --ppm_format p6"
mkdir -p "$(dirname "$OUT")" /tmp/cold_study; cd /tmp/cold_study
: > "$OUT"
run() {   # series name, run index
  local line
  line=$("$EXE" $ARGS 2>&1 >/dev/null | grep '^{' | tail -1)
  echo "{\"series\": \"$1\", \"run\": $2, \"stats\": $line}" >> "$OUT"
}
for i in 1 2 3 4 5 6 7 8 9 10; do run back_to_back $i; done
for i in 1 2 3 4 5 6 7 8 9 10; do sleep 2; run idle_2s_before $i; done
export RTIOW_RENDER_TWICE=1
for i in 1 2 3 4 5 6 7 8 9 10; do run second_render_of_the_process $i; done
unset RTIOW_RENDER_TWICE
#   series D..: back to back, with the clock warm-up knob (RTIOW_CLOCK_WARMUP_US: every SIMD busy for that long in front of the start event)
for us in 500 1000 2000 4000 8000; do
  export RTIOW_CLOCK_WARMUP_US=$us
  for i in 1 2 3 4 5 6 7 8 9 10; do run warmup_${us}us $i; done
done
unset RTIOW_CLOCK_WARMUP_US
# the same three series' summary
python3 - "$OUT" <<'PY'
import json, sys
import statistics as st
rows = [json.loads(l) for l in open(sys.argv[1])]
for s in ("back_to_back", "idle_2s_before", "second_render_of_the_process", "warmup_500us", "warmup_1000us", "warmup_2000us", "warmup_4000us", "warmup_8000us"):
    r = [x["stats"] for x in rows if x["series"] == s]
    ms = [x["render_ms"] for x in r]
    print(json.dumps({"series": s, "render_ms": [round(v, 2) for v in ms], "median": round(st.median(ms), 3), "max": round(max(ms), 3),
                      "main_clock_mhz": [round(x["clock_mhz"]["main"]) for x in r], "prepass_clock_mhz": [round(x["clock_mhz"]["prepass"]) for x in r],
                      "main_ms": [round(x["launch_ms"]["main"], 2) for x in r], "prepass_ms": [round(x["launch_ms"]["prepass"], 2) for x in r],
                      "not_in_a_launch_ms": [round(x["render_ms"] - sum(x["launch_ms"].values()), 3) for x in r]}))
PY
