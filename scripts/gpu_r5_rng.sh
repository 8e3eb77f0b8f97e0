#!/bin/bash
# seed-exact mode (rng=torch_device) vs device RNG on one box: parity tests of the seeded paths, then the two bench lines at B = 8 and B = 2
TAG=${1:-r5rng}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "torch_device or batch8 or seeded or shard or interface" > $O/pytest_rng.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_rng.log
for b in 8 2; do
for r in device torch_device; do timeout 300 python bench.py --batch-per-gpu $b --rng $r --steps 4 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_b${b}_$r.json 2> $O/bench_b${b}_$r.err; python -c "import json;d=json.load(open('$O/bench_b${b}_$r.json'));print('B=$b rng=$r', round(d['ms_per_step'],2),'ms', round(d['value']), 'tok/s gemm_time_frac', round(d['roofline']['gemm_time_frac'],3))"; done; done
