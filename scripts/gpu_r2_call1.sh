#!/bin/bash
# Round 2, GPU call 1: the new bf16x3 schedules (correctness, per-shape speed, ablations, end-to-end), PMC counters per
# schedule, and the full-size parity tests in the bench precision (B = 1 and BASELINE configs[2]'s B = 8).
O=gpurun_out/r2c1
mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $O/gpu_info.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3 or split3" -s -x > $O/1_x3_kernels.log 2>&1
echo "x3 kernel tests rc=$?"; tail -5 $O/1_x3_kernels.log
timeout 240 python scripts/gemm_x3_sched.py > $O/2_sched.txt 2>&1; cat $O/2_sched.txt
for p in 3 4 5; do
  VN_X3_PIPE=$p timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/3_bench_pipe$p.json 2> $O/3_bench_pipe$p.err
  python - <<PY
import json
try:
    b = json.load(open("$O/3_bench_pipe$p.json"))
    print("pipe $p:", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF-eq frac", round(b["roofline"]["frac"], 3), "attn", round(b["roofline"]["attention"]["achieved"], 1))
except Exception as e:
    print("pipe $p failed:", e)
PY
done
for p in 3 4 5; do bash scripts/gpu_pmc_x3.sh $p > $O/4_pmc_p$p.txt 2>&1; done
tail -n 8 $O/4_pmc_p4.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "full_size" -s > $O/5_full_size.log 2>&1
echo "full-size tests rc=$?"; grep -E "agreement|passed|failed|Error|error" $O/5_full_size.log | tail -20
