set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r04/gpu_tests_b.log
