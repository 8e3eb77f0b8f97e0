#!/bin/bash
# the whole `-m gpu` suite, smoke() and the headline bench line (with the CPU leg) on the current commit; $1 = tag
TAG=${1:-head}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/1_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/1_pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/1_pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/2_smoke.log
timeout 600 python bench.py > $O/3_bench_n1.json 2> $O/3_bench_n1.err; echo "bench rc=$?"; head -c 600 $O/3_bench_n1.json; echo
