#!/bin/bash
# Round 2, GPU call 17: the whole -m gpu suite + smoke on the build with the fma-chain RMSNorm, the fused reduce+norm, the scalar-based
# attention DMA addressing (no s_setprio) and build_mask on the device by default; then the headline bench line.
O=gpurun_out/r2c17
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/1_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -4 $O/1_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/2_smoke.log
timeout 600 python bench.py > $O/3_bench_n1.json 2> $O/3_bench_n1.err; echo "bench rc=$?"; head -c 600 $O/3_bench_n1.json; echo
