#!/bin/bash
# Round 2, GPU call 9: one QKV GEMM with the plane epilogue (V transposed through the LDS image): parity, bench, kernel stats.
O=gpurun_out/r2c9
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or bf16x3 or split3" -x > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; tail -3 $O/1_kernels.log
timeout 400 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -x > $O/2_model_bf16x3.log 2>&1
echo "bf16x3 model tests rc=$?"; tail -3 $O/2_model_bf16x3.log
timeout 200 python scripts/gemm_x3_sched.py > $O/3_sched.txt 2>&1; head -12 $O/3_sched.txt
for st in 1 0; do
  VN_X3_STAGED=$st timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/4_bench_staged$st.json 2> $O/4_bench_staged$st.err
done
python - <<PY
import json
for f in ("4_bench_staged1", "4_bench_staged0"):
    try:
        b = json.load(open("$O/" + f + ".json")); a = b["roofline"]["attention"]
        print(f, round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM", round(b["roofline"]["achieved"], 1), "TF frac", round(b["roofline"]["frac"], 3), "| attn", round(a["achieved"], 1), "TF", round(a["avg_launch_us"], 1), "us")
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/px && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -o vamp -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; for f in $(find /tmp/px -name "*kernel_trace.csv"); do python scripts/kstats_last_step.py $f vn_embed_kernel 16 20 > $O/5_last_vamp_kernel_stats.txt 2>&1; done; head -14 $O/5_last_vamp_kernel_stats.txt
