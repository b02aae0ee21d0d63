set -e
out=gpurun_out/bounce
mkdir -p $out
export SWEEP_CASES="off:SOLO_WAVES=0;rule:"
for b in 10 25 50 100; do
timeout -k 10 100 python scripts/solo_sweep.py --scene 1 --w 640 --h 384 --b $b > $out/scene1_640_b$b.jsonl 2>&1
timeout -k 10 100 python scripts/solo_sweep.py --scene 3 --w 640 --h 384 --b $b > $out/scene3_640_b$b.jsonl 2>&1
timeout -k 10 100 python scripts/solo_sweep.py --scene 1 --shard 1,4,2 --b $b > $out/scene1_s4_b$b.jsonl 2>&1
timeout -k 10 100 python scripts/solo_sweep.py --scene 3 --shard 1,4,2 --b $b > $out/scene3_s4_b$b.jsonl 2>&1
done
echo done
