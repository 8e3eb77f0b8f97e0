#!/bin/bash
# training-step bench (BASELINE configs[4]) + rocprofv3 kernel stats of the same command
mkdir -p gpurun_out/prof_train
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --workload train --steps 3 --warmup 1 ${BENCH_ARGS} > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
tail -3 gpurun_out/bench_train.err; cat gpurun_out/bench_train.json
cd /tmp; rm -rf /tmp/pt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_train/bench_trace.json 2> $R/gpurun_out/prof_train/trace.err
for f in $(find /tmp/pt -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/prof_train/; done
cd $R
for f in $(find /tmp/pt -name "*kernel_trace.csv"); do python scripts/kstats_last_step.py $f vn_embed_kernel 30 | tee gpurun_out/prof_train/last_step_stats.txt; done
