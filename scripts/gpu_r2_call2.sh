#!/bin/bash
# Round 2, GPU call 2: stream-K bf16x3 GEMM (correctness, per-shape speed vs the data-parallel form, ablations incl. the
# full-line DMA probe), end-to-end bench per tile height, model-level parity in bf16x3, rocprofv3 evidence of the headline
# command, and the staged training step on the x3 kernel.
O=gpurun_out/r2c2
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3 or split3" -s -x > $O/1_x3_kernels.log 2>&1
echo "x3 kernel tests rc=$?"; tail -6 $O/1_x3_kernels.log
timeout 300 python scripts/gemm_x3_sched.py > $O/2_sched.txt 2>&1; cat $O/2_sched.txt
for cfg in "128 1" "256 1" "128 0"; do
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
timeout 400 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -x > $O/4_model_bf16x3.log 2>&1
echo "bf16x3 model tests rc=$?"; tail -3 $O/4_model_bf16x3.log
bash scripts/gpu_bench_prof.sh r2c2 > $O/5_prof.txt 2>&1; tail -30 $O/5_prof.txt
VN_EXPERIMENTAL=1 timeout 500 python -m pytest tests/test_gpu_train.py -q -m gpu -k bf16x3_gemms -s > $O/6_train_x3.log 2>&1
echo "train x3 rc=$?"; tail -5 $O/6_train_x3.log
for x in 0 1; do
  VN_TRAIN_X3=$x timeout 200 python bench.py --workload train --no-cpu-baseline > $O/6_bench_train_x3_$x.json 2> $O/6_bench_train_x3_$x.err
  head -c 300 $O/6_bench_train_x3_$x.json; echo
done
