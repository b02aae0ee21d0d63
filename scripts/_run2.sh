set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=raytracingincuda_amd/lib/ab
timeout -k 10 600 python scripts/ab_pmc.py $L/r03_head.so $L/pooled.so $L/pooled_kr1.so --sets sq,lds 2>&1 | tee gpurun_out/r04/pmc_pooled_headline.jsonl
