#!/bin/bash
# Round 3, call 3: the DAC convolutions on the bf16x3 pipe — kernel tests, codec tests, codec throughput (both pipes), whole-request line
O=gpurun_out/r3c3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec.py -q -m gpu -x > $O/1_codec_tests.log 2>&1; echo "codec tests rc=$?"; tail -15 $O/1_codec_tests.log
timeout 300 python scripts/codec_bench.py > $O/2_codec_bench.txt 2>&1; cat $O/2_codec_bench.txt
timeout 400 python bench.py --e2e --no-cpu-baseline --steps 3 > $O/3_bench_e2e.json 2> $O/3_bench_e2e.err; echo "e2e rc=$?"; head -c 400 $O/3_bench_e2e.json; echo; tail -3 $O/3_bench_e2e.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c3/3_bench_e2e.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["stages_ms"])
PY
