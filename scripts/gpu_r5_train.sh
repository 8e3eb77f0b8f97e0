#!/bin/bash
# training step (BASELINE configs[4]) on the split-plane pipe vs the fp32-input MFMA on one box ($1 = tag): parity tests of the step, the two
# bench lines, rocprofv3 last-step table of the default
TAG=${1:-r5train}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "train_step_vs_oracle or deterministic or full_size_train_step or trajectory" > $O/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_train.log
timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_x3.json 2> $O/bench_train_x3.err; tail -2 $O/bench_train_x3.err
VN_TRAIN_X3=0 timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_f32.json 2> $O/bench_train_f32.err
for k in x3 f32; do python -c "import json;d=json.load(open('$O/bench_train_$k.json'));r=d['roofline'];print('$k', round(d['ms_per_step'],2),'ms/step', round(d['value']),'tok/s  gemm', round(r['achieved'],1),'TF frac', round(r['frac'],3),'gemm_time_frac', round(r['gemm_time_frac'],3), 'loss', d['config']['final_loss'])"; done
cd /tmp; rm -rf /tmp/pt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/$O/bench_trace.json 2> $R/$O/trace.err
cd $R
for f in $(find /tmp/pt -name "*kernel_trace.csv"); do python scripts/kstats_last_step.py $f vn_embed_kernel 30 > $O/last_step_stats.txt; done; head -32 $O/last_step_stats.txt | cut -c1-150
