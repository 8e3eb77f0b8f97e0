#!/bin/bash
# training attention on the split-plane pipe ($1 = tag): single-op parity vs autograd (both entries), the step's parity tests, then the
# bench line with / without it on the same box and the rocprofv3 last-step table
TAG=${1:-atx3}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention_train" > $O/pytest_attn.log 2>&1; echo "pytest attn rc=$?"; tail -15 $O/pytest_attn.log
if [ "$2" == "attn-only" ]; then exit 0; fi
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x > $O/pytest_train.log 2>&1; echo "pytest train rc=$?"; tail -6 $O/pytest_train.log
timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_ax3.json 2> $O/bench_train_ax3.err; tail -2 $O/bench_train_ax3.err
VN_TRAIN_ATTN_X3=0 timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_af32.json 2> $O/bench_train_af32.err
for k in ax3 af32; do python -c "import json;d=json.load(open('$O/bench_train_$k.json'));r=d['roofline'];print('$k', round(d['ms_per_step'],2),'ms/step', round(d['value']),'tok/s  gemm', round(r['achieved'],1),'TF frac', round(r['frac'],3),'attn', r.get('attention'), 'loss', d['config']['final_loss'])"; done
cd /tmp; rm -rf /tmp/pt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/$O/bench_trace.json 2> $R/$O/trace.err
cd $R
for f in $(find /tmp/pt -name "*kernel_trace.csv"); do python scripts/kstats_last_step.py $f vn_embed_kernel 30 > $O/last_step_stats.txt; done; head -24 $O/last_step_stats.txt | cut -c1-150
