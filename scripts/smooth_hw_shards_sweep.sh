set -e
out=gpurun_out/smoothhw_shards
mkdir -p $out
export SWEEP_CASES="hw6:;hw9:SMOOTH=9;hw12:SMOOTH=12;hw16:SMOOTH=16;hw24:SMOOTH=24;hw4:SMOOTH=4;hw3:SMOOTH=3"
for sh in 1,4,2 3,4,2 0,4,2 3,8,2 6,8,2 5,8,2 0,8,2; do
timeout -k 10 300 python scripts/solo_sweep.py --shard $sh > $out/shard_${sh//,/_}.jsonl 2>&1
done
echo done
