#!/bin/bash
# usage (through gpurun): bash scripts/gpu_tests.sh  -> logs under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/gpu_info.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/test_kernels.log
cat gpurun_out/test_kernels.log | tail -15
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu -s 2>&1 | tail -80 > gpurun_out/test_model.log
cat gpurun_out/test_model.log | tail -40
