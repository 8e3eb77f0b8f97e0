#!/bin/bash
# Round 2, GPU call 10: tile height by shape; B = 1 (configs[1]) variants; bench lines for configs 1 and 2.
O=gpurun_out/r2c10
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3" -x > $O/1_kernels.log 2>&1
echo "kernel tests rc=$?"; tail -2 $O/1_kernels.log
say() { python - "$1" <<'PY'
import json, sys
try:
    b = json.load(open(sys.argv[1])); r = b.get("roofline") or {}; a = r.get("attention") or {}
    print(sys.argv[1].split("/")[-1], round(b["value"]), "tok/s", round(b["ms_per_step"], 2), "ms; GEMM", round(r.get("achieved") or 0, 1), "TF frac", round(r.get("frac") or 0, 3), "| attn", round(a.get("achieved") or 0, 1), "TF")
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for bm in 0 128; do VN_X3_BM=$bm timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/2_bench_bm$bm.json 2> $O/2_bench_bm$bm.err; say $O/2_bench_bm$bm.json; done
for v in "sk-1_ax1" "sk0_ax1" "sk1_ax1" "sk-1_ax0"; do
  sk=$(echo $v | sed 's/sk\(.*\)_ax.*/\1/'); ax=$(echo $v | sed 's/.*_ax//')
  VN_X3_SK=$sk VN_ATTN_X3=$ax timeout 200 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/3_cfg1_$v.json 2> $O/3_cfg1_$v.err; say $O/3_cfg1_$v.json
done
for b in 1 2 4; do timeout 200 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/4_vamp_b$b.json 2> $O/4_vamp_b$b.err; say $O/4_vamp_b$b.json; done
