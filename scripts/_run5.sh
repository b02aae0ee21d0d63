set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 500 python scripts/tune_cases.py "PRE_STRIDE=1,SA=3;PRE_STRIDE=2,SA=3;PRE_STRIDE=2,SA=4;PRE_STRIDE=2,SA=6;PRE_STRIDE=3,SA=3;PRE_STRIDE=3,SA=6;PRE_STRIDE=4,SA=4;PRE_STRIDE=4,SA=8;PRE_STRIDE=2,SA=3,SMOOTH=8;PRE_STRIDE=3,SA=4,SMOOTH=9" --rounds 2 2>&1 | tee gpurun_out/r04/sparse_prepass_sweep_headline.jsonl
timeout -k 10 300 python scripts/tune_cases.py "PRE_STRIDE=1,SA=3;PRE_STRIDE=2,SA=3;PRE_STRIDE=3,SA=4;PRE_STRIDE=4,SA=4" --rounds 2 -- --scene 1 2>&1 | tee gpurun_out/r04/sparse_prepass_sweep_scene1.jsonl
timeout -k 10 300 python scripts/tune_cases.py "PRE_STRIDE=1,SA=3;PRE_STRIDE=2,SA=3;PRE_STRIDE=3,SA=4;PRE_STRIDE=4,SA=4" --rounds 2 -- --w 1280 --h 720 2>&1 | tee gpurun_out/r04/sparse_prepass_sweep_720p.jsonl
