set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=raytracingincuda_amd/lib/ab
timeout -k 10 300 python scripts/ab_libs.py $L/r03_head.so $L/rot_nosvc.so $L/rot_bar.so $L/pooled.so 2>&1 | tee gpurun_out/r04/ab_rotated_barrier_pooled.jsonl
timeout -k 10 600 python scripts/ab_pmc.py $L/rot_nosvc.so $L/rot_bar.so --sets sq,lds 2>&1 | tee gpurun_out/r04/pmc_rotated.jsonl
RTIOW_STATS_LIBRARY=$PWD/$L/stats_head.so timeout -k 10 200 python scripts/path_stats_probe.py > gpurun_out/r04/path_stats_head.json
RTIOW_STATS_LIBRARY=$PWD/$L/stats_pooled.so timeout -k 10 200 python scripts/path_stats_probe.py > gpurun_out/r04/path_stats_pooled.json
