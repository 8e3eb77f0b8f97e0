#!/bin/bash
O=gpurun_out/r3c8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16x3.py -q -m gpu -x -k "bf16x3 or geglu or x3 or conv1d or tile" > $O/1_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/1_tests.log
python scripts/gemm_x3_plan_sweep.py > $O/2_plan_sweep.txt 2>&1; grep -E "w1g" $O/2_plan_sweep.txt
for b in 2 8; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline > $O/3_bench_vamp_b$b.json 2> $O/3_bench_vamp_b$b.err; head -c 250 $O/3_bench_vamp_b$b.json; echo; done
