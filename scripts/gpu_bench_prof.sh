#!/bin/bash
# rocprofv3 evidence for the headline command (python bench.py, BASELINE configs[2]): kernel trace + stats, then the
# fabric-traffic counters FETCH_SIZE / WRITE_SIZE in separate --pmc passes (no trace domains in those).  $1 = output tag.
T=${1:-x3}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/px3 /tmp/pf /tmp/pw
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2> $R/$O/trace.err
for f in $(find /tmp/px3 -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats.csv; done
for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/last_vamp_kernel_stats.txt 2>&1; done
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/pmc_write.err
cd $R
python scripts/pmc_summary.py /tmp/pf FETCH_SIZE > $O/pmc_fetch_size.txt 2>&1
python scripts/pmc_summary.py /tmp/pw WRITE_SIZE > $O/pmc_write_size.txt 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt vn_gemm_x3 $O/traffic_gemm_x3.json > /dev/null 2>&1
head -12 $O/last_vamp_kernel_stats.txt; head -6 $O/pmc_fetch_size.txt; head -6 $O/pmc_write_size.txt
