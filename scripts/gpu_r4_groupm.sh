#!/bin/bash
# tile-walk group height of the split-plane GEMM (VN_X3_GROUPM: rows of tiles an XCD runs side by side; default 8) on the headline
# command: the traffic model in NOTES.md says 4 x 8 instead of 8 x 4 concurrent tiles cuts the fabric reads by 12 %.  $1 = tag
TAG=${1:-r4gm}
O=gpurun_out/$TAG
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_exchange.py -q -s -m gpu > $O/1_pytest_exchange.log 2>&1; echo "exchange tests rc=$?"; tail -2 $O/1_pytest_exchange.log
for g in 8 4 2 16 8; do
  VN_X3_GROUPM=$g timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_groupm_$g.json 2> $O/bench_groupm_$g.err
  python - <<PY
import json
d = json.load(open("$O/bench_groupm_$g.json")); print("VN_X3_GROUPM=$g", round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms; frac", round(d["roofline"]["frac"], 4))
PY
done
