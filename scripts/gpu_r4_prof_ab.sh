#!/bin/bash
# rocprofv3 kernel trace of the last vamp() call (B = 8, bf16x3), folded norms vs VN_FOLD_NORM=0.  $1 = tag
TAG=${1:-r4prof}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for mode in fold nofold; do
  rm -rf /tmp/pk_$mode
  if [ $mode = nofold ]; then export VN_FOLD_NORM=0; else unset VN_FOLD_NORM; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$mode -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-kernel-events > $R/$O/bench_under_rocprof_$mode.json 2> $R/$O/trace_$mode.err
  for f in $(find /tmp/pk_$mode -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 30 20 > $R/$O/last_vamp_kernel_stats_$mode.txt 2>&1; done
  for f in $(find /tmp/pk_$mode -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats_$mode.csv; done
done
unset VN_FOLD_NORM
cd $R
for mode in fold nofold; do echo "== $mode"; head -16 $O/last_vamp_kernel_stats_$mode.txt; done
