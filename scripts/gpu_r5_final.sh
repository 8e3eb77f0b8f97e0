#!/bin/bash
# Final evidence pass of round 5 on the last build ($1 = tag), ONE box: whole -m gpu suite, smoke(), the headline line (with the CPU leg) under
# the power sampler, the other bench lines (small batches, seed-exact, e2e, training with the attention on either pipe and on the fp32-input
# MFMA throughout), kernel tables of the last vamp() / the last training step, the codec trace.  The PMC / clock / probe files of
# scripts/gpu_r5_evidence.sh are not repeated: the inference kernels they describe did not change after that pass.
TAG=${1:-r5final}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
J() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d.get("dtype"), round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "frac", round(r.get("frac") or 0, 3), d.get("setup_s") or "")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket" >> $O/gpu_info.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -n 3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
timeout 600 python scripts/power_trace.py $O/power_bench -- python bench.py > $O/bench_n1.out 2> $O/bench_n1.err; grep '^{' $O/bench_n1.out > $O/bench_n1.json; J $O/bench_n1.json
timeout 400 python bench.py --config 1 --steps 10 --warmup 3 > $O/bench_n1_config1.json 2> $O/bench_n1_config1.err; J $O/bench_n1_config1.json
for b in 1 2 4; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_vamp_b$b.json 2> $O/bench_vamp_b$b.err; J $O/bench_vamp_b$b.json; done
timeout 300 python bench.py --rng torch_device --no-cpu-baseline --no-alt > $O/bench_rng_torch_device.json 2> $O/bench_rng.err; J $O/bench_rng_torch_device.json
timeout 300 python bench.py --no-cpu-baseline --no-alt > $O/bench_rng_device.json 2> /dev/null; J $O/bench_rng_device.json
timeout 400 python bench.py --e2e --no-cpu-baseline > $O/bench_e2e.json 2> $O/bench_e2e.err; J $O/bench_e2e.json
timeout 400 python bench.py --workload train --steps 4 --warmup 2 > $O/bench_train_n1.json 2> $O/bench_train_n1.err; J $O/bench_train_n1.json
VN_TRAIN_ATTN_X3=0 timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_n1_attn_f32.json 2> /dev/null; J $O/bench_train_n1_attn_f32.json
VN_TRAIN_X3=0 timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_n1_f32_mfma.json 2> /dev/null; J $O/bench_train_n1_f32_mfma.json
cd /tmp; rm -rf /tmp/px3 /tmp/pt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $R/$O/bench_under_rocprof.json 2> $R/$O/trace.err
for f in $(find /tmp/px3 -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats.csv; done
for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/last_vamp_kernel_stats.txt 2>&1; done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/trace_train.err
for f in $(find /tmp/pt -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 30 > $R/$O/train_last_step_kernel_stats.txt 2>&1; done
cd $R
head -8 $O/last_vamp_kernel_stats.txt | cut -c1-150; head -8 $O/train_last_step_kernel_stats.txt | cut -c1-150
bash scripts/gpu_codec_trace.sh $TAG/codec bf16x3 > $O/codec_trace.log 2>&1; tail -n 2 $O/codec_trace.log
