#!/bin/bash
# round 6, first contact of the guard-page harness with the hardware
mkdir -p gpurun_out; O=gpurun_out
echo "== harness self-test" ; timeout 600 python -m pytest tests/test_gpu_guard.py -x -q -s -k harness > $O/r06_guard_selftest.log 2>&1; echo rc=$?; tail -15 $O/r06_guard_selftest.log
for mode in end start; do
  echo "== guarded kernels + ragged cases ($mode)"
  VN_GUARD_ALLOC=$mode timeout 1500 python -m pytest tests/test_gpu_guard_cases.py tests/test_gpu_kernels.py -x -q -s -p no:cacheprovider --durations=15 > $O/r06_guard_kernels_$mode.log 2>&1
  echo rc=$?; tail -30 $O/r06_guard_kernels_$mode.log
done
echo "== bench"; timeout 600 python bench.py > $O/r06_first_bench.json 2> $O/r06_first_bench.err; echo rc=$?; cut -c1-400 $O/r06_first_bench.json
