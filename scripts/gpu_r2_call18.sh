#!/bin/bash
# Round 2, GPU call 18: 192-row tile of gemm_x3 (2 x 4 wave grid): kernel + model tests, plan sweep, bench.
O=gpurun_out/r2c18
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3 or x3" > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; grep -E "passed|failed|FAILED" $O/1_kernels.log | tail -12
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu > $O/2_bf16x3.log 2>&1
echo "bf16x3 tests rc=$?"; grep -E "passed|failed|FAILED" $O/2_bf16x3.log | tail -8
timeout 400 python scripts/gemm_x3_plan_sweep.py > $O/3_plan_sweep.txt 2> $O/3_plan_sweep.err; echo "sweep rc=$?"; cat $O/3_plan_sweep.txt; tail -3 $O/3_plan_sweep.err
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/4_bench.json 2> $O/4_bench.err; python - <<PY
import json
b = json.load(open("$O/4_bench.json")); a = b["roofline"]["attention"]
print("bench", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
PY
