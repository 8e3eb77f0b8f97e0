#!/bin/bash
# codec program validation: the codec test file, the fixed training test, the whole-request bench line.  $1 = tag
TAG=${1:-r4codec}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec.py -q -s -m gpu > $O/1_pytest_codec.log 2>&1; echo "codec tests rc=$?"; tail -3 $O/1_pytest_codec.log; grep -E "^(FAILED|ERROR)|launches behind" $O/1_pytest_codec.log | head -20
timeout 300 python -m pytest tests/test_gpu_train.py -q -m gpu -k "full_mode_resume" > $O/1_pytest_train_sel.log 2>&1; echo "train sel rc=$?"; tail -2 $O/1_pytest_train_sel.log
timeout 400 python bench.py --e2e --no-cpu-baseline > $O/3_bench_e2e.json 2> $O/3_bench_e2e.err; echo "e2e rc=$?"
VN_CODEC_EAGER=1 timeout 400 python bench.py --e2e --no-cpu-baseline > $O/3_bench_e2e_eager.json 2> $O/3_bench_e2e_eager.err
python - <<PY
import json
for f in ("3_bench_e2e", "3_bench_e2e_eager"):
    try:
        d = json.load(open("$O/%s.json" % f)); print(f, round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms", d["config"]["stages_ms"])
    except Exception as e: print(f, "FAILED", e)
PY
