# All frames / shards of the hand-out study (scripts/solo_sweep.py), one jsonl per configuration under gpurun_out/$1.
set -e
out=gpurun_out/${1:-solo}
mkdir -p $out
for sh in 0,8,2 1,8,2 2,8,2 3,8,2 4,8,2 5,8,2 6,8,2 7,8,2 0,4,2 1,4,2 2,4,2 3,4,2 0,2,8 1,2,8; do
  timeout -k 10 200 python scripts/solo_sweep.py --shard $sh > $out/shard_${sh//,/_}.jsonl 2>&1
done
timeout -k 10 200 python scripts/solo_sweep.py > $out/s3_1080p.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --w 1280 --h 720 > $out/s3_720p.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --w 960 --h 540 > $out/s3_540p.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --scene 1 --w 320 --h 192 --b 25 > $out/scene1_320.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --scene 1 --w 640 --h 384 --b 25 > $out/scene1_640.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --scene 1 --w 960 --h 576 --b 25 > $out/scene1_960.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --scene 1 --w 1920 --h 1080 > $out/scene1_1080p.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --prec 64 --shard 1,4,2 > $out/f64_shard_1_4_2.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --prec 64 --shard 3,8,2 > $out/f64_shard_3_8_2.jsonl 2>&1
timeout -k 10 200 python scripts/solo_sweep.py --prec 64 > $out/f64_1080p.jsonl 2>&1
echo done
