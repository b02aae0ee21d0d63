set -e
out=gpurun_out/age1
mkdir -p $out
export SWEEP_CASES="off:;y2:AGE_RESERVE=0,0,0,0.01,0.02;y4:AGE_RESERVE=0,0,0.01,0.02,0.04;y8:AGE_RESERVE=0,0.01,0.02,0.04,0.08;y16:AGE_RESERVE=0,0.02,0.04,0.08,0.16;y30:AGE_RESERVE=0,0.04,0.08,0.16,0.30;flat4:AGE_RESERVE=0,0.04,0.04,0.04,0.04;flat10:AGE_RESERVE=0,0.1,0.1,0.1,0.1;last10:AGE_RESERVE=0,0,0,0,0.1;last25:AGE_RESERVE=0,0,0,0,0.25"
timeout -k 10 300 python scripts/solo_sweep.py > $out/s3_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 > $out/scene1_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --w 1280 --h 720 > $out/s3_720p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --prec 64 > $out/f64_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --shard 1,2,8 > $out/shard_1_2_8.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 --w 1280 --h 768 --b 25 > $out/scene1_1280.jsonl 2>&1
echo done
