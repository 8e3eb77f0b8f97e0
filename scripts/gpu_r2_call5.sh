#!/bin/bash
# Round 2, GPU call 5: attention_x3 tuning (block size by cost model, last-tile masking, rescale skip): kernel-only timing per
# block size, parity, end-to-end bench.
O=gpurun_out/r2c5
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" -x > $O/1_kernels.log 2>&1
echo "attention tests rc=$?"; tail -3 $O/1_kernels.log
for w in 4; do timeout 120 python scripts/attn_bench.py 2>/dev/null; done > $O/2_attn_bench.txt; cat $O/2_attn_bench.txt
timeout 400 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -x > $O/3_model_bf16x3.log 2>&1
echo "bf16x3 model tests rc=$?"; tail -3 $O/3_model_bf16x3.log
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/4_bench.json 2> $O/4_bench.err
python - <<PY
import json
b = json.load(open("$O/4_bench.json")); a = b["roofline"]["attention"]
print(round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF-eq frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF-eq", round(a["avg_launch_us"], 1), "us")
PY
timeout 200 python bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/5_bench_cfg1.json 2> $O/5_bench_cfg1.err
head -c 400 $O/5_bench_cfg1.json; echo
