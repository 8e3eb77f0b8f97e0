#!/bin/bash
# Round 2, GPU call 7: XCD-aware attention grids (x3 and f32): parity, kernel-only timing, ablations, bench.
O=gpurun_out/r2c7
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" -x > $O/1_kernels.log 2>&1
echo "attention tests rc=$?"; tail -3 $O/1_kernels.log
for w in 0 4 6; do VN_ATTN_X3_WAVES=$w timeout 120 python scripts/attn_bench.py 2>/dev/null; done > $O/2_attn_bench.txt; cat $O/2_attn_bench.txt
for a in 1 6 7; do echo "ABL=$a"; VN_ATTN_X3_WAVES=4 VN_ATTN_X3_ABL=$a timeout 120 python scripts/attn_bench.py 2>/dev/null | head -1; done > $O/3_attn_ablate.txt; cat $O/3_attn_ablate.txt
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/4_bench.json 2> $O/4_bench.err
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dtype f32 > $O/4_bench_f32.json 2> $O/4_bench_f32.err
python - <<PY
import json
for f in ("4_bench", "4_bench_f32"):
    b = json.load(open("$O/" + f + ".json")); a = b["roofline"]["attention"]
    print(f, round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
PY
