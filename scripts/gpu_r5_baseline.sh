#!/bin/bash
# Round-5 baseline evidence on the round-4 kernels ($1 = tag): the small-batch lines (configs[1], vamp() B = 1 / 2), rocprofv3 kernel
# traces of configs[1] and B = 2 with the per-forward timeline (scripts/ktrace_forward.py), the under-filled-launch A/B
# (scripts/gemm_underfill_ab.py), the per-shape plan sweep and the attention decompositions at the small shapes.
TAG=${1:-r5base}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_config1.json 2> $O/bench_config1.err; head -c 250 $O/bench_config1.json; echo
for b in 1 2; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_vamp_b$b.json 2> $O/bench_vamp_b$b.err; head -c 250 $O/bench_vamp_b$b.json; echo; done
timeout 300 python scripts/gemm_underfill_ab.py > $O/gemm_underfill_ab.txt 2>&1; cat $O/gemm_underfill_ab.txt
cd /tmp; rm -rf /tmp/pc1 /tmp/pb2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc1 -o c1 -- python $R/bench.py --config 1 --steps 2 --warmup 2 --no-cpu-baseline --no-alt --no-kernel-events > $R/$O/bench_config1_under_rocprof.json 2> $R/$O/trace_c1.err
for f in $(find /tmp/pc1 -name "*kernel_trace.csv"); do python $R/scripts/ktrace_forward.py $f vn_embed_kernel 3 --full > $R/$O/config1_forward_timeline.txt 2>&1; python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 12 > $R/$O/config1_last_call_kernel_stats.txt 2>&1; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb2 -o b2 -- python $R/bench.py --batch-per-gpu 2 --steps 1 --warmup 2 --no-cpu-baseline --no-alt --no-kernel-events > $R/$O/bench_b2_under_rocprof.json 2> $R/$O/trace_b2.err
for f in $(find /tmp/pb2 -name "*kernel_trace.csv"); do python $R/scripts/ktrace_forward.py $f vn_embed_kernel 12 --full > $R/$O/b2_forward_timeline.txt 2>&1; python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/b2_last_vamp_kernel_stats.txt 2>&1; done
cd $R
grep -v "^  +" $O/config1_forward_timeline.txt | head -30
grep -v "^  +" $O/b2_forward_timeline.txt | head -24
