#!/bin/bash
# round 6: per-kernel table of ONE vamp() at B = 1 (configs[1]: 12 coarse steps, and --batch 1 of the headline: coarse + c2f)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pb1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb1 -o b1 -- python $R/bench.py --config 1 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-sharded-check --no-kernel-events > $O/r06_b1_under_rocprof.json 2> $O/r06_b1_trace.err
for f in $(find /tmp/pb1 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 12 > $O/r06_config1_last_vamp_kernel_stats.txt 2>&1; done
head -30 $O/r06_config1_last_vamp_kernel_stats.txt | cut -c1-160
