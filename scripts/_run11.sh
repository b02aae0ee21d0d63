set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 /tmp/e2e && cd /tmp/e2e
B=$GRAFT_REPO_ROOT/raytracingincuda_amd/bin
for i in 1 2 3 4; do $B/global-float-hip-raytrace --scene_id 3 --width 1920 --height 1080 --samples 100 --bounces 50 --threads 8 --stats; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r04/e2e_before.log
for i in 1 2; do $B/global-float-hip-raytrace --scene_id 3 --width 1920 --height 1080 --samples 100 --bounces 50 --threads 8 --stats --ppm_format p6; done 2>&1 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r04/e2e_before.log
nproc; df -h /tmp | tail -1
