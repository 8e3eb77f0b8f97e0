#!/bin/bash
# per-launch rocprofv3 kernel trace of one warm DAC encode + decode at B = 8 (scripts/codec_trace.py); $1 = output tag
T=${1:-codec_trace}
export VN_CODEC_PRECISION=${2:-bf16x3}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pct
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pct -o c -- python $R/scripts/codec_trace.py > /dev/null 2> $R/gpurun_out/$T/err.txt
for f in $(find /tmp/pct -name "*kernel_trace.csv"); do python $R/scripts/codec_trace_report.py $f > $R/gpurun_out/$T/codec_kernel_trace.txt 2>&1; done
tail -5 $R/gpurun_out/$T/codec_kernel_trace.txt
