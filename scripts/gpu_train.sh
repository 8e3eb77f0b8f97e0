#!/bin/bash
# training-step parity tests on the GPU
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -s ${PYTEST_ARGS} 2>&1 | tail -40 | tee gpurun_out/test_train.log
