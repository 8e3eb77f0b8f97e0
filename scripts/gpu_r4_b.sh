#!/bin/bash
# round 4: folded-norm validation pass — split-plane test files + bench. $1 = tag
TAG=${1:-r4b}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_saturation.py tests/test_gpu_exchange.py -q -s -m gpu -x > $O/1_pytest_x3.log 2>&1; echo "x3 tests rc=$?"; tail -4 $O/1_pytest_x3.log; grep -E "^(FAILED|ERROR)|folded|forced split" $O/1_pytest_x3.log | head -30
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/3_bench_n1.json 2> $O/3_bench_n1.err; echo "bench rc=$?"; tail -3 $O/3_bench_n1.err
VN_FOLD_NORM=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/3_bench_n1_nofold.json 2> $O/3_bench_n1_nofold.err; echo "bench nofold rc=$?"
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/3_bench_cfg1.json 2> $O/3_bench_cfg1.err; echo "cfg1 rc=$?"
VN_FOLD_NORM=0 timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/3_bench_cfg1_nofold.json 2> $O/3_bench_cfg1_nofold.err
python - <<PY
import json
for f in ("3_bench_n1", "3_bench_n1_nofold", "3_bench_cfg1", "3_bench_cfg1_nofold"):
    try:
        d = json.load(open("$O/%s.json" % f))
        a = d.get("alt") or {}
        print(f, d["dtype"], round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms; frac", round(d["roofline"]["frac"], 3), "| alt", round(a.get("value", 0)), round(a.get("ms_per_step", 0), 2))
    except Exception as e:
        print(f, "FAILED", e)
PY
