set -e
out=gpurun_out/knobs
mkdir -p $out
export SWEEP_CASES="default:;deal32:DEAL=32;deal16:DEAL=16;deal1:DEAL=1;sa1:SA=1;sa2:SA=2;sa4:SA=4;sa6:SA=6"
timeout -k 10 300 python scripts/solo_sweep.py > $out/s3_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 > $out/scene1_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --w 1280 --h 720 > $out/s3_720p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --prec 64 > $out/f64_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 --w 1280 --h 768 --b 25 > $out/scene1_1280.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --shard 1,2,8 > $out/shard_1_2_8.jsonl 2>&1
echo done
