set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=raytracingincuda_amd/lib/ab
for v in r03_head pooled pooled_kr1 pooled_kr3; do
  echo "== $v small"; RTIOW_HIP_LIBRARY=$PWD/$L/$v.so timeout -k 10 120 python scripts/one_render.py --sched 2 --w 640 --h 360 --s 100 --reps 2 --md5
done 2>&1 | tee gpurun_out/r04/pooled_small.log
timeout -k 10 700 python scripts/ab_libs.py $L/r03_head.so $L/rotated_only.so $L/pooled.so $L/pooled_kr1.so $L/pooled_kr3.so 2>&1 | tee gpurun_out/r04/ab_pooled_headline.jsonl
