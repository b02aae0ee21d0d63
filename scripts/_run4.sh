set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=raytracingincuda_amd/lib/ab
timeout -k 10 300 python scripts/ab_libs.py $L/r03_head.so $L/rot_bar.so $L/pooled.so -- --w 3840 --h 2160 2>&1 | tee gpurun_out/r04/ab_pooled_4k.jsonl
timeout -k 10 300 python scripts/ab_libs.py $L/r03_head.so $L/rot_bar.so $L/pooled.so -- --scene 1 2>&1 | tee gpurun_out/r04/ab_pooled_scene1.jsonl
timeout -k 10 300 python scripts/ab_libs.py $L/r03_head.so $L/pooled.so -- --prec 64 2>&1 | tee gpurun_out/r04/ab_pooled_fp64.jsonl
timeout -k 10 600 python scripts/ab_pmc.py $L/r03_head.so $L/pooled.so --sets sq,lds -- --w 3840 --h 2160 2>&1 | tee gpurun_out/r04/pmc_pooled_4k.jsonl
timeout -k 10 600 python scripts/ab_pmc.py $L/r03_head.so $L/pooled.so --sets sq,lds -- --scene 1 2>&1 | tee gpurun_out/r04/pmc_pooled_scene1.jsonl
