#!/bin/bash
# Round 2, GPU call 23: 96-query attention blocks (4 per CU) vs 128-query blocks: bitwise A/B, probe, bench.
O=gpurun_out/r2c23
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; grep -E "passed|failed|FAILED" $O/1_kernels.log | tail -12
ATTN_PROBE_QUICK=1 timeout 300 python scripts/attn_probe.py > $O/2_attn_probe.txt 2> $O/2_attn_probe.err; echo "probe rc=$?"; cat $O/2_attn_probe.txt; tail -3 $O/2_attn_probe.err
for nw in 0 4; do VN_ATTN_X3_NW=$nw timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/3_bench_nw$nw.json 2> $O/3_bench_nw$nw.err; python - <<PY
import json
b = json.load(open("$O/3_bench_nw$nw.json")); a = b["roofline"]["attention"]
print("VN_ATTN_X3_NW=$nw", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
PY
done
