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
for threads in "${THREADS[@]}"; do
  for scene_id in "${SCENE_IDS[@]}"; do
    for samples in "${SAMPLES[@]}"; do
      for bounces in "${BOUNCES[@]}"; do
        for i in "${!WIDTHS[@]}"; do
          width="${WIDTHS[$i]}"; height="${HEIGHTS[$i]}"
          echo "--- scene=$scene_id ${width}x${height} samples=$samples bounces=$bounces threads=$threads ---" >&2
          for run in $(seq 1 "$RUNS"); do
            OUT=$(cd "$WORK" && "$EXE" --scene_id "$scene_id" --width "$width" --height "$height" \
                   --samples "$samples" --bounces "$bounces" --threads "$threads")
            echo "${scene_id},${width},${height},${samples},${bounces},${threads},${run},${OUT}" >> "$CSV"
          done
        done
      done
    done
  done
done
echo "All combinations and runs complete. All results saved in '$CSV'." >&2
