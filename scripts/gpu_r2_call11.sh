#!/bin/bash
# Round 2, GPU call 11: attention_x3 phase order A/B (kernel-only), parity, bench; stream-K-free gemm_x3 regression.
O=gpurun_out/r2c11
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or bf16x3" -x > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; tail -3 $O/1_kernels.log
for o in 0 1 2; do echo "ORD=$o"; VN_ATTN_X3_ORD=$o timeout 120 python scripts/attn_bench.py 2>/dev/null | head -3; done > $O/2_attn_ord.txt; cat $O/2_attn_ord.txt
for o in 0 1 2; do VN_ATTN_X3_ORD=$o timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_attention and bf16x3" -x 2>&1 | tail -1; done
timeout 400 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -x > $O/3_model_bf16x3.log 2>&1
echo "bf16x3 model tests rc=$?"; tail -2 $O/3_model_bf16x3.log
for o in 0 1; do VN_ATTN_X3_ORD=$o timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/4_bench_ord$o.json 2> $O/4_bench_ord$o.err; python - <<PY
import json
b = json.load(open("$O/4_bench_ord$o.json")); a = b["roofline"]["attention"]
print("ord $o", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
PY
done
