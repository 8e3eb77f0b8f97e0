#!/bin/bash
# bench + rocprofv3 kernel trace; logs under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "full_vamp_properties or api_surface" 2>&1 | tail -5
timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o vamp -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err
cd $GRAFT_REPO_ROOT; cat gpurun_out/prof_bench.json
find /tmp/prof -name "*stats*" | head; 
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/kernel_stats.csv; done
head -30 gpurun_out/kernel_stats.csv
