#!/bin/bash
# Round 2, GPU call 6: attention_x3 ablations (which phase bounds the kernel) + rocprofv3 kernel stats of the headline command.
O=gpurun_out/r2c6
mkdir -p $O
for a in 0 1 6 7; do echo "ABL=$a (1: no exp/split, 6: no MFMAs, 7: neither)"; VN_ATTN_X3_WAVES=4 VN_ATTN_X3_ABL=$a timeout 120 python scripts/attn_bench.py 2>/dev/null | head -2; done > $O/1_attn_ablate.txt; cat $O/1_attn_ablate.txt
bash scripts/gpu_bench_prof.sh r2c6 > $O/2_prof.txt 2>&1; tail -32 $O/2_prof.txt
