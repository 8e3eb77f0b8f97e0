#!/bin/bash
# First GPU call of round 2: validate and time everything that was staged after the round-1 GPU budget ended, cheapest first.
# Every step writes its own log under gpurun_out/r2/ and the script keeps going after a failure (|| true).
mkdir -p gpurun_out/r2
O=gpurun_out/r2
export VN_EXPERIMENTAL=1
# 1. f16x2 single ops: split formula, GEMM vs float64 at both tile heights, split-K, epilogues
timeout 120 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "f16x2 or split2h" -s > $O/1_h2_kernels.log 2>&1 || true
tail -4 $O/1_h2_kernels.log
# 2. speed against bf16x3 / fp32 MFMA per model shape (+ errors vs float64)
timeout 90 python scripts/gemm_h2_bench.py > $O/2_h2_bench.txt 2>&1 || true
cat $O/2_h2_bench.txt
# 3. model-level parity in precision="f16x2" (the bf16x3 test bodies), unfused then fused producers
timeout 200 python -m pytest tests/test_gpu_f16x2.py -q -m gpu -s > $O/3_h2_model.log 2>&1 || true
tail -6 $O/3_h2_model.log
VN_H2_FUSE=1 timeout 200 python -m pytest tests/test_gpu_f16x2.py -q -m gpu > $O/3b_h2_model_fused.log 2>&1 || true
tail -3 $O/3b_h2_model_fused.log
# 4. headline in f16x2: tile 128 / 256, fused producers
for cfg in "VN_H2_TILE=128 VN_H2_FUSE=0" "VN_H2_TILE=128 VN_H2_FUSE=1" "VN_H2_TILE=256 VN_H2_FUSE=1"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout 120 python bench.py --dtype f16x2 --no-cpu-baseline > $O/4_bench_$tag.json 2> $O/4_bench_$tag.err || true
  python - <<PY
import json
try:
    b = json.load(open("$O/4_bench_$tag.json"))
    print("$cfg", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF-eq")
except Exception as e:
    print("$cfg failed:", e)
PY
done
# 5. training step with every GEMM on the bf16x3 kernel (child process of the gated test), then its bench line
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -k bf16x3_gemms -s > $O/5_train_x3.log 2>&1 || true
tail -5 $O/5_train_x3.log
VN_TRAIN_X3=1 timeout 200 python bench.py --workload train --no-cpu-baseline > $O/5_bench_train_x3.json 2> $O/5_bench_train_x3.err || true
cat $O/5_bench_train_x3.json | head -c 400; echo
