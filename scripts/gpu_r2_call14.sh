#!/bin/bash
# Round 2, GPU call 14: pipelined attention_x3 (bitwise A/B vs the plain loop + speed), fused reduce+norm A/B again, bench.
O=gpurun_out/r2c14
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "splitk_reduce_rmsnorm or attention or rmsnorm" > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; grep -E "fused vs|passed|failed|FAILED" $O/1_kernels.log | tail -12
ATTN_PROBE_QUICK=1 timeout 300 python scripts/attn_probe.py > $O/2_attn_probe.txt 2> $O/2_attn_probe.err; echo "probe rc=$?"; cat $O/2_attn_probe.txt; tail -3 $O/2_attn_probe.err
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -s > $O/3_bf16x3.log 2>&1
echo "bf16x3 tests rc=$?"; grep -E "fused vs|passed|failed|FAILED" $O/3_bf16x3.log | tail -8
for pipe in 1 0; do VN_ATTN_X3_PIPE=$pipe timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/4_bench_pipe$pipe.json 2> $O/4_bench_pipe$pipe.err; python - <<PY
import json
b = json.load(open("$O/4_bench_pipe$pipe.json")); a = b["roofline"]["attention"]
print("pipe $pipe", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
PY
done
