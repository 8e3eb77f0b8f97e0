#!/bin/bash
# Round 2, GPU call 19: rocprofv3 kernel trace of the headline bench on the 192-row-tile build.
O=gpurun_out/r2c19
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/px3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2> $R/$O/trace.err
for f in $(find /tmp/px3 -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats.csv; done
for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 20 > $R/$O/last_vamp_kernel_stats.txt 2>&1; done
cd $R
head -16 $O/last_vamp_kernel_stats.txt
