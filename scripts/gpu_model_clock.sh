#!/bin/bash
# In-model shader clock + matrix-pipe occupancy of the bf16x3 GEMM / attention launches of `python bench.py` (BASELINE configs[2]):
# one --pmc pass with the kernel trace (no other trace domain).  $1 = output tag, $2.. = extra bench.py flags.
T=${1:-clock}; shift
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pclk
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pclk -o c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events "$@" > $R/$O/bench_under_pmc.json 2> $R/$O/pmc.err
cd $R
python scripts/pmc_clock_summary.py /tmp/pclk > $O/model_clock.txt 2>&1
cat $O/model_clock.txt
