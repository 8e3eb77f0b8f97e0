#!/bin/bash
# Round-4 evidence pass on the current build ($1 = tag): the whole `-m gpu` suite, smoke(), the headline bench line (bf16x3 primary + f16x2
# `alt` + CPU leg) under the power / clock sampler, the other bench lines, rocprofv3 kernel stats of the headline command, its
# FETCH_SIZE / WRITE_SIZE passes, the in-model clock / MFMA-occupancy pass.  Everything lands under gpurun_out/$TAG/.
TAG=${1:-r4ev}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket" >> $O/gpu_info.txt
python -c "import torch; print('visible devices', torch.cuda.device_count())" >> $O/gpu_info.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu > $O/1_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -3 $O/1_pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/1_pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/2_smoke.log
timeout 600 python scripts/power_trace.py $O/4_power_bench -- python bench.py > $O/3_bench_n1.out 2> $O/3_bench_n1.err; echo "bench rc=$?"
grep '^{' $O/3_bench_n1.out > $O/3_bench_n1.json; head -c 300 $O/3_bench_n1.json; echo; tail -7 $O/4_power_bench.txt
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 > $O/3_bench_n1_config1.json 2> $O/3_bench_n1_config1.err; head -c 200 $O/3_bench_n1_config1.json; echo
for b in 1 2 4; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline > $O/3_bench_vamp_b$b.json 2> $O/3_bench_vamp_b$b.err; head -c 200 $O/3_bench_vamp_b$b.json; echo; done
timeout 300 python bench.py --dtype f32 --no-cpu-baseline > $O/3_bench_n1_f32.json 2> $O/3_bench_n1_f32.err; head -c 200 $O/3_bench_n1_f32.json; echo
timeout 300 python bench.py --rng torch_device --no-cpu-baseline --no-alt > $O/3_bench_rng_torch_device.json 2> $O/3_bench_rng.err; head -c 200 $O/3_bench_rng_torch_device.json; echo
timeout 400 python bench.py --e2e --no-cpu-baseline > $O/3_bench_e2e.json 2> $O/3_bench_e2e.err; head -c 200 $O/3_bench_e2e.json; echo
VN_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-alt > $O/3_bench_gpus2_one_gpu.json 2> $O/3_bench_gpus2.err; echo "gpus2 rc=$?"; head -c 200 $O/3_bench_gpus2_one_gpu.json; echo
# ---- rocprofv3: kernel trace + stats of the headline command (primary precision only), then the fabric-traffic counters
cd /tmp; rm -rf /tmp/px3 /tmp/pf /tmp/pw
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $R/$O/bench_under_rocprof.json 2> $R/$O/trace.err
for f in $(find /tmp/px3 -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats.csv; done
for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/last_vamp_kernel_stats.txt 2>&1; done
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-alt > /dev/null 2> $R/$O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-alt > /dev/null 2> $R/$O/pmc_write.err
cd $R
python scripts/pmc_summary.py /tmp/pf FETCH_SIZE > $O/pmc_fetch_size.txt 2>&1
python scripts/pmc_summary.py /tmp/pw WRITE_SIZE > $O/pmc_write_size.txt 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt vn_gemm_x3 $O/traffic_gemm_x3.json > /dev/null 2>&1
head -14 $O/last_vamp_kernel_stats.txt; cat $O/traffic_gemm_x3.json | head -12
bash scripts/gpu_model_clock.sh $TAG/clock --no-alt > $O/5_clock.txt 2>&1; tail -12 $O/5_clock.txt
