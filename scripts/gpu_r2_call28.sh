#!/bin/bash
# Round 2, GPU call 28: tiled A planes (norm / attention / GEGLU producers write the tiled layout): tests + same-box A/B bench.
O=gpurun_out/r2c28
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3 or x3 or splitk_reduce or attention or rmsnorm" > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; grep -E "passed|failed|FAILED" $O/1_kernels.log | tail -8
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu > $O/2_bf16x3.log 2>&1
echo "bf16x3 tests rc=$?"; grep -E "passed|failed|FAILED" $O/2_bf16x3.log | tail -5
for t in 1 0 1 0; do VN_X3_ATILED=$t timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/3_bench_at$t.json 2> $O/3_bench_at$t.err; python - <<PY
import json
b = json.load(open("$O/3_bench_at$t.json"))
print("VN_X3_ATILED=$t", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM frac", round(b["roofline"]["frac"], 3), "avg", round(b["roofline"]["avg_launch_us"], 1), "us")
PY
done
