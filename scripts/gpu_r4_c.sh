#!/bin/bash
# folded-norm iteration: the split-plane test file, then per-kernel A/B under rocprofv3, then plain bench lines. $1 = tag
TAG=${1:-r4d}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -s -m gpu -x > $O/1_pytest_x3.log 2>&1; echo "x3 tests rc=$?"; tail -3 $O/1_pytest_x3.log; grep -E "^(FAILED|ERROR)|folded vs oracle|forced split" $O/1_pytest_x3.log | head -30
bash scripts/gpu_r4_prof_ab.sh $TAG > $O/2_prof.txt 2>&1; cat $O/2_prof.txt | grep -E "==|last step|gemm_x3_kernel<(2|3|5), [23]|rmsnorm|rowprep|attention"
for mode in fold nofold; do
  if [ $mode = nofold ]; then export VN_FOLD_NORM=0; else unset VN_FOLD_NORM; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt > $O/3_bench_n1_$mode.json 2> $O/3_bench_n1_$mode.err
  timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/3_bench_cfg1_$mode.json 2> $O/3_bench_cfg1_$mode.err
done
unset VN_FOLD_NORM
python - <<PY
import json
for f in ("3_bench_n1_fold", "3_bench_n1_nofold", "3_bench_cfg1_fold", "3_bench_cfg1_nofold"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, d["dtype"], round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms; frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
