#!/bin/bash
# quick GPU regression: all gpu tests + bench (N=1) in both precisions
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v -E "^(RCCL|HIP|ROCm) version|^Hostname|^Librccl" | tail -8 | tee gpurun_out/test_gpu.log
timeout 1200 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -2 gpurun_out/bench.err; cat gpurun_out/bench.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dtype bf16 > gpurun_out/bench_bf16.json 2>> gpurun_out/bench.err
cat gpurun_out/bench_bf16.json
