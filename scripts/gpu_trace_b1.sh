#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/t1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t1 -o b1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch-per-gpu 1 --coarse-only --no-kernel-events > /tmp/b1.json 2>/dev/null
cat /tmp/b1.json | cut -c1-220
python $R/scripts/kstats.py $(find /tmp/t1 -name "*kernel_stats.csv") 12
