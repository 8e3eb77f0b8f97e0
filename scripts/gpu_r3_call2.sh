#!/bin/bash
# Round 3, call 2: whole GPU suite on the restructured attention / per-context tuning build, attention kernel times, small-batch lines.
O=gpurun_out/r3c2
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/1_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -5 $O/1_pytest_gpu.log
grep -E "^(FAILED|ERROR)" $O/1_pytest_gpu.log | head -20
timeout 300 python scripts/attn_bench.py > $O/2_attn_bench.txt 2>&1; cat $O/2_attn_bench.txt
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/3_bench_config1.json 2> $O/3_bench_config1.err; head -c 300 $O/3_bench_config1.json; echo
for b in 1 2; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline > $O/3_bench_vamp_b$b.json 2> $O/3_bench_vamp_b$b.err; head -c 250 $O/3_bench_vamp_b$b.json; echo; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/4_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/4_smoke.log
