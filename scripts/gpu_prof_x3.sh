#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench (bf16x3) -> gpurun_out/prof_x3/
mkdir -p gpurun_out/prof_x3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/px3
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_x3/bench_trace.json 2> $R/gpurun_out/prof_x3/trace.err
for f in $(find /tmp/px3 -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/prof_x3/kernel_stats.csv; done
cd $R
for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python scripts/kstats_last_step.py $f > gpurun_out/prof_x3/last_step_kernel_stats.txt 2>&1; done
cat gpurun_out/prof_x3/last_step_kernel_stats.txt | head -14; cat gpurun_out/prof_x3/bench_trace.json | head -c 600
