#!/bin/bash
# One-command evidence pass on the current build, ONE box ($1 = tag under gpurun_out/, default r06): the whole `-m gpu` suite (the guard-page
# children included), smoke(), the headline line (CPU leg, sharded-path check) under the power / clock sampler, the other bench lines
# (configs[1], B = 1 / 2 / 4, f32, seed-exact, whole request, training on both pipes, two ranks on one GPU), rocprofv3 kernel tables, the
# fabric-traffic counters of both split-plane precisions (stamped with the kernel source's hash: bench.py quotes them only for that
# source), the instruction mix per kernel, the in-model clock pass, the codec trace.  Copy what is to be judged from gpurun_out/$TAG to
# profiles/ (scripts/collect_evidence.sh).
TAG=${1:-r06}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
J() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(next(l for l in open(sys.argv[1]) if l.startswith("{")))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d.get("dtype"), round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "frac", round(r.get("frac") or 0, 3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket" >> $O/gpu_info.txt
timeout 2400 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -n 3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
# ---- fabric-traffic counters FIRST (their own passes, no trace domains): the bench lines below quote them (bench.py reads
# profiles/r06_traffic_*.json and checks the kernel source's hash), so the capture of THIS build is put in place before they run
cd /tmp; rm -rf /tmp/pf /tmp/pw /tmp/pfh /tmp/pwh
P="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-sharded-check"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o fetch -- python $R/bench.py $P --no-alt > /dev/null 2> $R/$O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o write -- python $R/bench.py $P --no-alt > /dev/null 2> $R/$O/pmc_write.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pfh -o fetch -- python $R/bench.py --dtype f16x2 $P > /dev/null 2> $R/$O/pmc_fetch_h2.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pwh -o write -- python $R/bench.py --dtype f16x2 $P > /dev/null 2> $R/$O/pmc_write_h2.err
cd $R
python scripts/pmc_summary.py /tmp/pf FETCH_SIZE > $O/pmc_fetch_size.txt 2>&1
python scripts/pmc_summary.py /tmp/pw WRITE_SIZE > $O/pmc_write_size.txt 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt vn_gemm_x3 $O/traffic_x3.json > /dev/null 2>&1
python scripts/pmc_summary.py /tmp/pfh FETCH_SIZE > $O/pmc_fetch_size_h2.txt 2>&1
python scripts/pmc_summary.py /tmp/pwh WRITE_SIZE > $O/pmc_write_size_h2.txt 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch_size_h2.txt $O/pmc_write_size_h2.txt vn_gemm_x3 $O/traffic_h2.json > /dev/null 2>&1
cp $O/traffic_x3.json profiles/r06_traffic_x3.json; cp $O/traffic_h2.json profiles/r06_traffic_h2.json; head -12 $O/traffic_x3.json
timeout 900 python scripts/power_trace.py $O/power_bench -- python bench.py --steps 20 --warmup 5 > $O/bench_n1.out 2> $O/bench_n1.err; grep '^{' $O/bench_n1.out > $O/bench_n1.json; J $O/bench_n1.json; tail -6 $O/power_bench.txt
timeout 400 python bench.py --config 1 --steps 10 --warmup 3 > $O/bench_n1_config1.json 2> $O/bench_n1_config1.err; J $O/bench_n1_config1.json
for b in 1 2 4; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-sharded-check > $O/bench_vamp_b$b.json 2> $O/bench_vamp_b$b.err; J $O/bench_vamp_b$b.json; done
timeout 300 python bench.py --dtype f32 --no-cpu-baseline --no-sharded-check > $O/bench_n1_f32.json 2> $O/bench_n1_f32.err; J $O/bench_n1_f32.json
timeout 300 python bench.py --rng torch_device --no-cpu-baseline --no-alt --no-sharded-check > $O/bench_rng_torch_device.json 2> $O/bench_rng.err; J $O/bench_rng_torch_device.json
timeout 400 python bench.py --e2e --no-cpu-baseline > $O/bench_e2e.json 2> $O/bench_e2e.err; J $O/bench_e2e.json
timeout 400 python bench.py --workload train --steps 4 --warmup 2 > $O/bench_train_n1.json 2> $O/bench_train_n1.err; J $O/bench_train_n1.json
VN_TRAIN_X3=0 timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_n1_f32_mfma.json 2> /dev/null; J $O/bench_train_n1_f32_mfma.json
VN_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-alt --no-cpu-baseline > $O/bench_gpus2_one_gpu.json 2> $O/bench_gpus2.err; J $O/bench_gpus2_one_gpu.json
# ---- rocprofv3: kernel trace + stats of the headline command and of the training step
cd /tmp; rm -rf /tmp/px3 /tmp/pt /tmp/psq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-sharded-check > $R/$O/bench_under_rocprof.json 2> $R/$O/trace.err
for f in $(find /tmp/px3 -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats.csv; done
for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/last_vamp_kernel_stats.txt 2>&1; done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/trace_train.err
for f in $(find /tmp/pt -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 30 > $R/$O/train_last_step_kernel_stats.txt 2>&1; done
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/psq -o sq -- python $R/bench.py $P --no-alt > /dev/null 2> $R/$O/pmc_sq.err
cd $R
python scripts/pmc_per_kernel.py /tmp/psq > $O/pmc_lds_mfma_per_kernel.txt 2>&1
head -12 $O/last_vamp_kernel_stats.txt | cut -c1-150; cat $O/traffic_x3.json | head -12; head -12 $O/pmc_lds_mfma_per_kernel.txt
bash scripts/gpu_model_clock.sh $TAG/clock --no-alt --no-sharded-check > $O/clock.txt 2>&1; tail -10 $O/clock.txt
bash scripts/gpu_codec_trace.sh $TAG/codec bf16x3 > $O/codec_trace.log 2>&1; tail -n 2 $O/codec_trace.log
