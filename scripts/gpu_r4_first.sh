#!/bin/bash
# round 4, first GPU pass: the new tests (saturation ledger, heavy-tailed models, RCCL exchange, cfg_guidance), smoke(), the headline
# bench line with its f16x2 `alt` block under the power / clock sampler.  $1 = tag
TAG=${1:-r4a}
O=gpurun_out/$TAG
mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket" >> $O/gpu_info.txt
ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $O/gpu_info.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_saturation.py tests/test_gpu_exchange.py -q -s -m gpu > $O/1_pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $O/1_pytest_new.log; grep -E "^(FAILED|ERROR)|agreement|RCCL" $O/1_pytest_new.log | head -30
timeout 600 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_model.py -q -m gpu -k "falls_back or generate_vs_oracle or error_behaviour" > $O/1_pytest_sel.log 2>&1; echo "selected tests rc=$?"; tail -3 $O/1_pytest_sel.log; grep -E "^(FAILED|ERROR)" $O/1_pytest_sel.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/2_smoke.log
timeout 600 python scripts/power_trace.py $O/4_power_bench -- python bench.py --steps 6 --warmup 2 > $O/3_bench_n1.out 2> $O/3_bench_n1.err; echo "bench rc=$?"
grep '^{' $O/3_bench_n1.out > $O/3_bench_n1.json; head -c 700 $O/3_bench_n1.json; echo; cat $O/4_power_bench.txt
python - <<PY
import json
d = json.load(open("$O/3_bench_n1.json"))
print("PRIMARY", d["dtype"], round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms; frac", round(d["roofline"]["frac"], 3), "attn us", round(d["roofline"]["attention"]["avg_launch_us"], 1))
a = d.get("alt")
if a: print("ALT", a["dtype"], round(a["value"]), "tok/s", round(a["ms_per_step"], 1), "ms; frac", round(a["roofline"]["frac"], 3), a["effective_precision"], "fallbacks", a["fallbacks"])
print("setup_s", d.get("setup_s"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
