#!/bin/bash
# Round 2, GPU call 3: stream-K v2 (per-XCD round-robin whole tiles + k-split tail): correctness, per-shape speed, bench.
O=gpurun_out/r2c3
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3 or split3" -s -x > $O/1_x3_kernels.log 2>&1
echo "x3 kernel tests rc=$?"; tail -6 $O/1_x3_kernels.log
timeout 300 python scripts/gemm_x3_sched.py > $O/2_sched.txt 2>&1; head -12 $O/2_sched.txt
for cfg in "128 1" "256 1"; do
  set -- $cfg
  VN_X3_BM=$1 VN_X3_SK=$2 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/3_bench_bm$1_sk$2.json 2> $O/3_bench_bm$1_sk$2.err
  python - <<PY
import json
try:
    b = json.load(open("$O/3_bench_bm$1_sk$2.json"))
    print("bm $1 sk $2:", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF-eq frac", round(b["roofline"]["frac"], 3), "attn", round(b["roofline"]["attention"]["achieved"], 1))
except Exception as e:
    print("bm $1 sk $2 failed:", e)
PY
done
