set -e
out=gpurun_out/tier1
mkdir -p $out
export SWEEP_CASES="off:SOLO_WAVES=0;rule:;t256x8:TIER_WAVES=256,TIER_LANES=8;t512x8:TIER_WAVES=512,TIER_LANES=8;t1024x8:TIER_WAVES=1024,TIER_LANES=8;t512x4:TIER_WAVES=512,TIER_LANES=4;t1024x4:TIER_WAVES=1024,TIER_LANES=4;t512x16:TIER_WAVES=512,TIER_LANES=16;t1024x16:TIER_WAVES=1024,TIER_LANES=16"
for sh in 3,8,2 5,8,2 6,8,2 0,8,2 1,4,2 3,4,2 1,2,8; do
  timeout -k 10 200 python scripts/solo_sweep.py --shard $sh > $out/shard_${sh//,/_}.jsonl 2>&1
done
timeout -k 10 200 python scripts/solo_sweep.py --w 960 --h 540 > $out/s3_540p.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --w 1280 --h 720 > $out/s3_720p.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --scene 1 --w 640 --h 384 --b 25 > $out/scene1_640.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --scene 1 --w 320 --h 192 --b 25 > $out/scene1_320.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --prec 64 --shard 3,8,2 > $out/f64_shard_3_8_2.jsonl 2>&1
echo done
