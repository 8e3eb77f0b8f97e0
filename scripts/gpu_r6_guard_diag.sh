#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
mode=start
echo "== guarded ragged cases + kernels ($mode)"
VN_GUARD_ALLOC=$mode timeout 1800 python -m pytest tests/test_gpu_guard_cases.py tests/test_gpu_kernels.py -q -s -p no:cacheprovider > $O/r06_guard_kernels_$mode.log 2>&1
echo rc=$?; grep -E "^(FAILED|ERROR|GUARD)|passed|failed|Memory access|fault" $O/r06_guard_kernels_$mode.log | cut -c1-220
bash scripts/gpu_guard_full.sh both
