set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for s in 1 2 3 6; do timeout -k 10 120 python scripts/prepass_tail_probe.py $s; done 2>&1 | tee gpurun_out/r04/prepass_tail_probe.jsonl
