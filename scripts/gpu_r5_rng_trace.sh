#!/bin/bash
# seed-exact mode: where the extra time goes.  Kernel trace of one vamp() in --rng torch_device and in the default mode on one box, plus
# the two bench lines; $1 = tag
TAG=${1:-rngtr}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in torch_device device; do
  timeout 300 python bench.py --rng $m --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/bench_$m.json 2> $O/bench_$m.err
  python -c "import json;d=json.load(open('$O/bench_$m.json'));print('$m', round(d['ms_per_step'],2),'ms', d.get('setup_s'))"
done
cd /tmp
for m in torch_device device; do
  rm -rf /tmp/pr_$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$m -o t -- python $R/bench.py --rng $m --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-kernel-events > $R/$O/trace_$m.json 2> $R/$O/trace_$m.err
  for f in $(find /tmp/pr_$m -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/last_vamp_$m.txt 2>&1; cp $f $R/$O/kernel_trace_$m.csv; done
  head -16 $R/$O/last_vamp_$m.txt | cut -c1-160
done
