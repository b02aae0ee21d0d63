set -e
out=gpurun_out/smoothstrip
mkdir -p $out
# SMOOTH_STRIP: rows of one smoothing strip (local rows); 0 = the whole local image
export SWEEP_CASES="default:;strip8:SMOOTH_STRIP=8;strip2:SMOOTH_STRIP=2;strip4:SMOOTH_STRIP=4;strip16:SMOOTH_STRIP=16;whole:SMOOTH_STRIP=0;whole_hw4:SMOOTH_STRIP=0,SMOOTH=4;whole_hw9:SMOOTH_STRIP=0,SMOOTH=9"
timeout -k 10 300 python scripts/solo_sweep.py > $out/s3_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 > $out/scene1_1080p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --w 1280 --h 720 > $out/s3_720p.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --prec 64 > $out/f64_1080p.jsonl 2>&1
for sh in 1,2,8 1,4,2 3,4,2 3,8,2 6,8,2; do
timeout -k 10 300 python scripts/solo_sweep.py --shard $sh > $out/shard_${sh//,/_}.jsonl 2>&1
done
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 --w 1280 --h 768 --b 25 > $out/scene1_1280.jsonl 2>&1
timeout -k 10 300 python scripts/solo_sweep.py --scene 1 --w 640 --h 384 --b 25 > $out/scene1_640.jsonl 2>&1
echo done
