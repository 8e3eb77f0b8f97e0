#!/bin/bash
# rocprofv3 passes for bench.py (kernel trace + stats, then PMC passes one counter group at a time)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_trace.json 2> $R/gpurun_out/prof/trace.err
find /tmp/p1 -type f | head -20 > $R/gpurun_out/prof/files.txt
for f in $(find /tmp/p1 -name "*stats*.csv"); do cp $f $R/gpurun_out/prof/; done
# PMC: HBM read and write bytes in separate passes (TCC slots), 1 untimed step, no event bracketing
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/gpurun_out/prof/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/gpurun_out/prof/pmc_write.err
find /tmp/p2 /tmp/p3 -type f | head >> $R/gpurun_out/prof/files.txt
cd $R
python scripts/pmc_summary.py /tmp/p2 FETCH_SIZE > gpurun_out/prof/pmc_fetch_summary.txt 2>&1
python scripts/pmc_summary.py /tmp/p3 WRITE_SIZE > gpurun_out/prof/pmc_write_summary.txt 2>&1
cat gpurun_out/prof/files.txt; head -25 gpurun_out/prof/*kernel_stats.csv; cat gpurun_out/prof/pmc_fetch_summary.txt gpurun_out/prof/pmc_write_summary.txt
