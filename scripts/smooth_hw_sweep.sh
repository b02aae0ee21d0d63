set -e
out=gpurun_out/smoothhw
mkdir -p $out
export SWEEP_CASES="hw6:;hw7:SMOOTH=7;hw8:SMOOTH=8;hw9:SMOOTH=9;hw10:SMOOTH=10;hw12:SMOOTH=12;hw16:SMOOTH=16;hw5:SMOOTH=5"
timeout -k 10 300 python scripts/solo_sweep.py > $out/s3_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 > $out/scene1_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --w 1280 --h 720 > $out/s3_720p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --w 960 --h 540 > $out/s3_540p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --w 2560 --h 1440 --reps 4 > $out/s3_1440p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --prec 64 > $out/f64_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 --w 1280 --h 768 --b 25 > $out/scene1_1280.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 --w 640 --h 384 --b 25 > $out/scene1_640.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 --w 320 --h 192 --b 25 > $out/scene1_320.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --shard 1,2,8 > $out/shard_1_2_8.jsonl 2>&1
echo done
