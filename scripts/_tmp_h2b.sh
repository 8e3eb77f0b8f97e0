mkdir -p gpurun_out/h2b
timeout 600 python -m pytest tests/test_gpu_f16x2.py -q -x -m gpu 2>&1 | tail -3
timeout 300 python scripts/gemm_h2_bench.py > gpurun_out/h2b/bench.txt 2>&1; cat gpurun_out/h2b/bench.txt
