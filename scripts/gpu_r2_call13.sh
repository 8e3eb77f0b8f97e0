#!/bin/bash
# Round 2, GPU call 13: fused reduce+norm single-op A/B, attention_x3 with scalar-based DMA addressing (tests + probe), device
# build_mask, bf16x3 model tests, bench.
O=gpurun_out/r2c13
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "splitk_reduce_rmsnorm or attention" > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; grep -E "fused vs|passed|failed" $O/1_kernels.log | tail -8
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "build_mask_on_device or torch_device" > $O/2_mask.log 2>&1
echo "mask tests rc=$?"; tail -3 $O/2_mask.log
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -s > $O/3_bf16x3.log 2>&1
echo "bf16x3 tests rc=$?"; grep -E "fused vs|passed|failed|FAILED" $O/3_bf16x3.log | tail -8
ATTN_PROBE_QUICK=1 timeout 300 python scripts/attn_probe.py > $O/4_attn_probe.txt 2> $O/4_attn_probe.err; echo "probe rc=$?"; cat $O/4_attn_probe.txt; tail -3 $O/4_attn_probe.err
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/5_bench.json 2> $O/5_bench.err; python - <<PY
import json
b = json.load(open("$O/5_bench.json")); a = b["roofline"]["attention"]
print("bench", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
PY
