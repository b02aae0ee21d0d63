set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=raytracingincuda_amd/lib/ab
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r04/gpu_tests_a.log
timeout -k 10 300 python scripts/ab_libs.py $L/r03_head.so $L/bitwise_inside.so $L/r04_a.so 2>&1 | tee gpurun_out/r04/ab_r04_a.jsonl
