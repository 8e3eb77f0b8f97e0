#!/bin/bash
O=gpurun_out/r3c9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" > $O/1_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/1_tests.log
timeout 300 python scripts/attn_bench.py > $O/2_attn_bench.txt 2>&1; cat $O/2_attn_bench.txt
