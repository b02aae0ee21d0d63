#!/usr/bin/bash
# ppm_diff_dirs.sh -- batch comparison of two directories of .ppm renders, the workflow of the
# reference's timing-benchmarks/ppm_diff.sh (README.md:101-116): render the float and the double
# variant into two directories, then diff them pair by pair.
#
#   tools/ppm_diff_dirs.sh <input_dir1> <input_dir2> <output_dir> [ppm_diff gate options...]
#
# Same contract as the reference script: the files of each directory are taken oldest first
# (`ls -tr`), paired by position (a name mismatch only warns: global_float_* vs global_double_*),
# and each pair's difference image goes to <output_dir>/float_double_diff_<name of the first file>.
# Additions: bin/ppm_diff prints mean / p99 / max per pair, and any gate option given after the three
# directories (--max-mean X, --max-p99 X, --max-abs X) makes a pair fail; the script then exits 2.
set -u
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
PPM_DIFF="${PPM_DIFF:-$HERE/raytracingincuda_amd/bin/ppm_diff}"
if [ "$#" -lt 3 ]; then
    echo "Usage: $0 <input_dir1> <input_dir2> <output_dir>"
    echo "  <input_dir1>: The first directory containing PPM files."
    echo "  <input_dir2>: The second directory containing PPM files."
    echo "  <output_dir>: The directory where output files will be saved."
    exit 1
fi
DIR1="$1"; DIR2="$2"; OUT="$3"; shift 3
for d in "$DIR1" "$DIR2"; do
    [ -d "$d" ] || { echo "Error: Input directory '$d' not found."; exit 1; }
done
[ -x "$PPM_DIFF" ] || { echo "Error: Executable '$PPM_DIFF' not found or not executable (run: python -m raytracingincuda_amd.build)."; exit 1; }
mkdir -p "$OUT" || { echo "Error: Could not create output directory '$OUT'."; exit 1; }
mapfile -t A < <(ls -tr "$DIR1"/*.ppm 2>/dev/null)
mapfile -t B < <(ls -tr "$DIR2"/*.ppm 2>/dev/null)
[ "${#A[@]}" -gt 0 ] || { echo "Error: No .ppm files found in input directory 1 '$DIR1'."; exit 1; }
[ "${#B[@]}" -gt 0 ] || { echo "Error: No .ppm files found in input directory 2 '$DIR2'."; exit 1; }
if [ "${#A[@]}" -ne "${#B[@]}" ]; then
    echo "Error: Number of files in input directories do not match."
    echo "Files found in '$DIR1': ${#A[@]}"
    echo "Files found in '$DIR2': ${#B[@]}"
    exit 1
fi
echo "Found ${#A[@]} pairs of files to process."
failed=0
for i in "${!A[@]}"; do
    a="${A[$i]}"; b="${B[$i]}"
    na="$(basename "$a")"; nb="$(basename "$b")"
    if [ "$na" != "$nb" ]; then
        echo "Warning: Filenames do not match for pair $i based on sorted order:"
        echo "  '$a' vs '$b'"
        echo "  Processing anyway, assuming sorted order is correct."
    fi
    out="$OUT/float_double_diff_$na"
    echo "Processing pair: '$a' and '$b'"
    echo "Output will be saved to: '$out'"
    if ! "$PPM_DIFF" "$a" "$b" "$out" "$@"; then
        echo "Error: '$PPM_DIFF' failed for pair '$na'."
        echo "Output file '$out' may be incomplete or missing."
        failed=1
    fi
    echo "Finished processing '$na'."
    echo ""
done
[ "$failed" -eq 0 ] || exit 2
