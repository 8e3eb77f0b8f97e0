#!/bin/bash
O=gpurun_out/r3c5
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_codec.py -q -m gpu -x > $O/1_codec_tests.log 2>&1; echo "codec tests rc=$?"; tail -4 $O/1_codec_tests.log
timeout 300 python scripts/codec_bench.py > $O/2_codec_bench.txt 2>&1; cat $O/2_codec_bench.txt
bash scripts/gpu_codec_trace.sh r3c5/trace_x3 bf16x3 > /dev/null 2>&1
bash scripts/gpu_codec_trace.sh r3c5/trace_f32 f32 > /dev/null 2>&1
paste <(awk '{print $NF" "$(NF-1)}' $O/trace_x3/codec_kernel_trace.txt | head -90) <(cut -c1-40,60- $O/trace_f32/codec_kernel_trace.txt | head -90) | head -95
