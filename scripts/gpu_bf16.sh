#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k bf16 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k bf16 2>&1 | grep -E "bf16 fast|passed|failed|Error"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dtype bf16 2>&1 | tail -1 | cut -c1-1400
