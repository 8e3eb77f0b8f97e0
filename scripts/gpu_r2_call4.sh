#!/bin/bash
# Round 2, GPU call 4: bf16x3 attention (attention_x3.hip + the QK3 / VT3 GEMM epilogues): kernel and model parity, bench A/B.
O=gpurun_out/r2c4
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or bf16x3 or split3" -s -x > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; tail -8 $O/1_kernels.log
timeout 400 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -x > $O/2_model_bf16x3.log 2>&1
echo "bf16x3 model tests rc=$?"; tail -4 $O/2_model_bf16x3.log
for cfg in "1 0" "0 0" "1 2" "1 4"; do
  set -- $cfg
  VN_ATTN_X3=$1 VN_ATTN_X3_WAVES=$2 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/3_bench_ax$1_w$2.json 2> $O/3_bench_ax$1_w$2.err
  python - <<PY
import json
try:
    b = json.load(open("$O/3_bench_ax$1_w$2.json"))
    a = b["roofline"]["attention"]
    print("attn_x3 $1 waves $2:", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF-eq frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF-eq", round(a["avg_launch_us"], 1), "us")
except Exception as e:
    print("attn_x3 $1 waves $2 failed:", e)
PY
done
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "full_size_vamp and bf16x3" -s > $O/4_full_size.log 2>&1
echo "full-size rc=$?"; grep -E "agreement|passed|failed|rror" $O/4_full_size.log | tail
