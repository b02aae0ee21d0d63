#!/bin/bash
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out/r05
python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_final.log 2>&1 || { tail -40 gpurun_out/r05/gpu_tests_final.log; exit 1; }
tail -3 gpurun_out/r05/gpu_tests_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
python3 bench.py > gpurun_out/r05/bench_default_final.json 2> gpurun_out/r05/bench_default_final.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05/bench_default_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['counters_from'][:80]); print([ (e['name'], e['ms_per_step'], e['frac']) for e in d['extra_configs']]); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
