#!/bin/bash
# rocprofv3 kernel trace of the LoRA-only fine-tuning step; prints the last step's per-kernel stats
mkdir -p gpurun_out/prof_lora
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pl
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o lora -- python $R/bench.py --workload train --lora-only --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_lora/bench_trace.json 2> $R/gpurun_out/prof_lora/trace.err
cd $R
for f in $(find /tmp/pl -name "*kernel_trace.csv"); do python scripts/kstats_last_step.py $f vn_embed_kernel 30 | tee gpurun_out/prof_lora/last_step_stats.txt; done
