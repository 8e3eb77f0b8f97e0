#!/bin/bash
# Round 2, GPU call 12: fused split-K reduce + RMSNorm (A/B test + bench), attention_x3 probes (blocks per CU, stagger, phase trace).
O=gpurun_out/r2c12
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -x > $O/1_bf16x3.log 2>&1
echo "bf16x3 tests rc=$?"; tail -3 $O/1_bf16x3.log
timeout 300 python scripts/attn_probe.py > $O/2_attn_probe.txt 2> $O/2_attn_probe.err; echo "probe rc=$?"; cat $O/2_attn_probe.txt; tail -3 $O/2_attn_probe.err
for f in 0 1; do VN_X3_FUSE_NORM=$f timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/3_bench_fuse$f.json 2> $O/3_bench_fuse$f.err; python - <<PY
import json
b = json.load(open("$O/3_bench_fuse$f.json")); a = b["roofline"]["attention"]
print("fuse $f", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
PY
done
