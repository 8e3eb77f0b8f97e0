#!/bin/bash
# small-batch bench lines on the current build ($1 = tag): configs[1], vamp() B = 1 / 2 / 4, and the headline (B = 8) without the CPU leg
TAG=${1:-r5small}
O=gpurun_out/$TAG
mkdir -p $O
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_config1.json 2> $O/bench_config1.err; python -c "import json;d=json.load(open('$O/bench_config1.json'));print('config1', round(d['ms_per_step'],2),'ms', round(d['roofline']['frac'],3))"
for b in 1 2 4 8; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_vamp_b$b.json 2> $O/bench_vamp_b$b.err; python -c "import json;d=json.load(open('$O/bench_vamp_b$b.json'));print('vamp B=$b', round(d['ms_per_step'],2),'ms', round(d['value']), 'tok/s frac', round(d['roofline']['frac'],3), 'attn us', round(d['roofline']['attention']['avg_launch_us'],1))"; done
