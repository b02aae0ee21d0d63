#!/usr/bin/bash
# hip_benchmark.sh -- the reference's benchmark loop (global_float_benchmark.sh:1-86) over the HIP
# executables: nested loops threads x scene x samples x bounces x (W,H) x RUNS, one CSV row per run
# with the reference's schema (global_float_benchmark.sh:25,74):
#   scene_id,width,height,samples,bounces,threads,run,render_only_time_ms,end_to_end_time_ms
# so timing-benchmarks/process.py, or bin/csv_avg, averages it unchanged.
#
#   tools/hip_benchmark.sh [float|double] [output.csv]
# Grids are overridable from the environment, e.g.
#   SCENE_IDS="1 3" WIDTHS="320 1280" HEIGHTS="192 768" SAMPLES="100" BOUNCES="25 50" THREADS="8 16" RUNS=5
# STATS_JSONL=file adds one JSON line per run next to the CSV: the CSV keys plus what
# `--stats` reports (Mrays/s, kernel time, registers, LDS, scene source, solo waves, wall time of every phase).
# BASELINE_CONFIGS=1 runs the BASELINE.json configurations instead of the grid (the variant picks fp32 or fp64):
#   float : scene 1 320x192 10 spp 25 bounces; scene 3 1280x720 and 1920x1080, 100 spp, 50 bounces
#   double: scene 3 1920x1080, 500 spp, 50 bounces
set -u
VARIANT="${1:-float}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
EXE="$HERE/raytracingincuda_amd/bin/global-${VARIANT}-hip-raytrace"
CSV="${2:-$HERE/benchmarks/hip_global_${VARIANT}_timing.csv}"
read -r -a SCENE_IDS <<< "${SCENE_IDS:-1}"
read -r -a WIDTHS <<< "${WIDTHS:-320 480 640 960 1280}"
read -r -a HEIGHTS <<< "${HEIGHTS:-192 288 384 576 768}"
read -r -a SAMPLES <<< "${SAMPLES:-100}"
read -r -a BOUNCES <<< "${BOUNCES:-25}"
read -r -a THREADS <<< "${THREADS:-4 8 16 32}"
RUNS="${RUNS:-5}"
[ -x "$EXE" ] || { echo "missing $EXE (run: python -m raytracingincuda_amd.build)" >&2; exit 1; }
mkdir -p "$(dirname "$CSV")"
WORK="$(mktemp -d)"; trap 'rm -rf "$WORK"' EXIT      # the .ppm of every run is overwritten, as in the reference
echo "scene_id,width,height,samples,bounces,threads,run,render_only_time_ms,end_to_end_time_ms" > "$CSV"
STATS_JSONL="${STATS_JSONL:-}"
[ -n "$STATS_JSONL" ] && : > "$STATS_JSONL"
one_run() {   # scene width height samples bounces threads run
  local OUT
  if [ -n "$STATS_JSONL" ]; then
    OUT=$(cd "$WORK" && "$EXE" --scene_id "$1" --width "$2" --height "$3" --samples "$4" --bounces "$5" --threads "$6" --stats 2> "$WORK/stats.json")
    printf '{"scene_id": %s, "width": %s, "height": %s, "samples": %s, "bounces": %s, "threads": %s, "run": %s, "stats": %s}\n' \
           "$1" "$2" "$3" "$4" "$5" "$6" "$7" "$(tail -n 1 "$WORK/stats.json")" >> "$STATS_JSONL"
  else
    OUT=$(cd "$WORK" && "$EXE" --scene_id "$1" --width "$2" --height "$3" --samples "$4" --bounces "$5" --threads "$6")
  fi
  echo "$1,$2,$3,$4,$5,$6,$7,${OUT}" >> "$CSV"
}
if [ "${BASELINE_CONFIGS:-0}" = "1" ]; then
  if [ "$VARIANT" = "double" ]; then CONFIGS=("3 1920 1080 500 50")
  else CONFIGS=("1 320 192 10 25" "3 1280 720 100 50" "3 1920 1080 100 50"); fi
  for cfg in "${CONFIGS[@]}"; do
    read -r scene_id width height samples bounces <<< "$cfg"
    echo "--- scene=$scene_id ${width}x${height} samples=$samples bounces=$bounces threads=8 ---" >&2
    for run in $(seq 1 "$RUNS"); do one_run "$scene_id" "$width" "$height" "$samples" "$bounces" 8 "$run"; done
  done
  echo "BASELINE configurations complete. Results saved in '$CSV'." >&2
  exit 0
fi
for threads in "${THREADS[@]}"; do
  for scene_id in "${SCENE_IDS[@]}"; do
    for samples in "${SAMPLES[@]}"; do
      for bounces in "${BOUNCES[@]}"; do
        for i in "${!WIDTHS[@]}"; do
          width="${WIDTHS[$i]}"; height="${HEIGHTS[$i]}"
          echo "--- scene=$scene_id ${width}x${height} samples=$samples bounces=$bounces threads=$threads ---" >&2
          for run in $(seq 1 "$RUNS"); do
            one_run "$scene_id" "$width" "$height" "$samples" "$bounces" "$threads" "$run"
          done
        done
      done
    done
  done
done
echo "All combinations and runs complete. All results saved in '$CSV'." >&2
