#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k attention 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "full_size or forward" 2>&1 | grep -E "dlogit|passed|failed"
