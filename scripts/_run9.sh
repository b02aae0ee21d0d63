set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=raytracingincuda_amd/lib/ab
timeout -k 10 400 python scripts/ab_libs.py $L/r03_head.so $L/second_root_nopretest.so $L/bitwise_inside.so 2>&1 | tee gpurun_out/r04/ab_bitwise_inside.jsonl
timeout -k 10 400 python scripts/ab_libs.py $L/r03_head.so $L/second_root_nopretest.so $L/bitwise_inside.so -- --w 3840 --h 2160 2>&1 | tee -a gpurun_out/r04/ab_bitwise_inside.jsonl
timeout -k 10 400 python scripts/ab_libs.py $L/r03_head.so $L/second_root_nopretest.so $L/bitwise_inside.so -- --scene 1 2>&1 | tee -a gpurun_out/r04/ab_bitwise_inside.jsonl
timeout -k 10 400 python scripts/ab_libs.py $L/r03_head.so $L/bitwise_inside.so -- --prec 64 2>&1 | tee -a gpurun_out/r04/ab_bitwise_inside.jsonl
timeout -k 10 600 python scripts/ab_pmc.py $L/bitwise_inside.so --sets sq 2>&1 | tee gpurun_out/r04/pmc_bitwise_inside.jsonl
