#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (no PMC passes); last-step per-kernel table
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/p1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o vamp -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_trace.json 2> $R/gpurun_out/prof/trace.err
for f in $(find /tmp/p1 -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/prof/; done
cd $R
for f in $(find /tmp/p1 -name "*kernel_trace.csv"); do python scripts/kstats_last_step.py $f vn_apply_mask 24 2 | tee gpurun_out/prof/last_step_stats.txt; done
