set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 /tmp/e2e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rng or golden or drop_in or shard or full_frames" 2>&1 | tail -4
cd /tmp/e2e
B=$GRAFT_REPO_ROOT/raytracingincuda_amd/bin
for i in 1 2 3; do $B/global-float-hip-raytrace --scene_id 3 --width 1920 --height 1080 --samples 100 --bounces 50 --threads 8 --stats; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r04/e2e_rng_table.log
for i in 1 2; do $B/global-float-hip-raytrace --scene_id 3 --width 1920 --height 1080 --samples 100 --bounces 50 --threads 8 --stats --ppm_format p6; done 2>&1 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r04/e2e_rng_table.log
