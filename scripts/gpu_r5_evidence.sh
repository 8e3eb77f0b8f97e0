#!/bin/bash
# Round-5 evidence pass on the current build ($1 = tag).  ONE box: the headline bench line under the power / clock sampler, the other
# bench lines (small batches with the round-5 tile / attention choices ON and OFF on the same box, f32, seed-exact, e2e, training on both
# pipes), rocprofv3 kernel stats + timelines, the fabric-traffic and LDS / MFMA instruction counters, the in-model clock pass and the
# matrix-pipe power probe (with the real-split-plane mode).  Everything lands under gpurun_out/$TAG/.
TAG=${1:-r5ev}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
J() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d.get("dtype"), round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "frac", round(r.get("frac") or 0, 3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket" >> $O/gpu_info.txt
timeout 600 python scripts/power_trace.py $O/power_bench -- python bench.py > $O/bench_n1.out 2> $O/bench_n1.err; grep '^{' $O/bench_n1.out > $O/bench_n1.json; J $O/bench_n1.json; tail -7 $O/power_bench.txt
timeout 400 python bench.py --config 1 --steps 10 --warmup 3 > $O/bench_n1_config1.json 2> $O/bench_n1_config1.err; J $O/bench_n1_config1.json
for b in 1 2 4; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_vamp_b$b.json 2> $O/bench_vamp_b$b.err; J $O/bench_vamp_b$b.json; done
# the same lines with the round-5 small-batch choices switched off (96-row k-split tile + 192-row GEGLU tile, pair-split attention): same box
export VN_X3_TILE96=0 VN_ATTN_X3_PAIR=0
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_r4choices_config1.json 2> /dev/null; J $O/bench_r4choices_config1.json
for b in 1 2 4 8; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_r4choices_vamp_b$b.json 2> /dev/null; J $O/bench_r4choices_vamp_b$b.json; done
unset VN_X3_TILE96 VN_ATTN_X3_PAIR
timeout 300 python bench.py --dtype f32 --no-cpu-baseline > $O/bench_n1_f32.json 2> $O/bench_n1_f32.err; J $O/bench_n1_f32.json
timeout 300 python bench.py --rng torch_device --no-cpu-baseline --no-alt > $O/bench_rng_torch_device.json 2> $O/bench_rng.err; J $O/bench_rng_torch_device.json
timeout 300 python bench.py --rng torch_device --batch-per-gpu 2 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_rng_torch_device_b2.json 2> /dev/null; J $O/bench_rng_torch_device_b2.json
timeout 400 python bench.py --e2e --no-cpu-baseline > $O/bench_e2e.json 2> $O/bench_e2e.err; J $O/bench_e2e.json
timeout 400 python bench.py --workload train --steps 4 --warmup 2 > $O/bench_train_n1.json 2> $O/bench_train_n1.err; J $O/bench_train_n1.json
VN_TRAIN_X3=0 timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_n1_f32_mfma.json 2> /dev/null; J $O/bench_train_n1_f32_mfma.json
VN_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-alt > $O/bench_gpus2_one_gpu.json 2> $O/bench_gpus2.err; J $O/bench_gpus2_one_gpu.json
# ---- rocprofv3: kernel trace + stats of the headline command, timelines of the small batches, the training step
cd /tmp; rm -rf /tmp/px3 /tmp/pc1 /tmp/pb2 /tmp/pt /tmp/pf /tmp/pw /tmp/psq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $R/$O/bench_under_rocprof.json 2> $R/$O/trace.err
for f in $(find /tmp/px3 -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats.csv; done
for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/last_vamp_kernel_stats.txt 2>&1; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc1 -o c1 -- python $R/bench.py --config 1 --steps 2 --warmup 2 --no-cpu-baseline --no-alt --no-kernel-events > /dev/null 2> $R/$O/trace_c1.err
for f in $(find /tmp/pc1 -name "*kernel_trace.csv"); do python $R/scripts/ktrace_forward.py $f vn_embed_kernel 3 --full > $R/$O/config1_forward_timeline.txt 2>&1; python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 12 > $R/$O/config1_last_call_kernel_stats.txt 2>&1; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb2 -o b2 -- python $R/bench.py --batch-per-gpu 2 --steps 1 --warmup 2 --no-cpu-baseline --no-alt --no-kernel-events > /dev/null 2> $R/$O/trace_b2.err
for f in $(find /tmp/pb2 -name "*kernel_trace.csv"); do python $R/scripts/ktrace_forward.py $f vn_embed_kernel 6 --full > $R/$O/b2_forward_timeline.txt 2>&1; python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 vamp > $R/$O/b2_last_vamp_kernel_stats.txt 2>&1; done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/trace_train.err
for f in $(find /tmp/pt -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 30 > $R/$O/train_last_step_kernel_stats.txt 2>&1; done
# ---- counters (their own passes, no trace domains): fabric traffic, then LDS / MFMA / VMEM instruction counts per kernel
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-alt > /dev/null 2> $R/$O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-alt > /dev/null 2> $R/$O/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/psq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-alt > /dev/null 2> $R/$O/pmc_sq.err
rm -rf /tmp/pfh /tmp/pwh
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pfh -o fetch -- python $R/bench.py --dtype f16x2 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/pmc_fetch_h2.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pwh -o write -- python $R/bench.py --dtype f16x2 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/pmc_write_h2.err
cd $R
python scripts/pmc_summary.py /tmp/pfh FETCH_SIZE > $O/pmc_fetch_size_h2.txt 2>&1
python scripts/pmc_summary.py /tmp/pwh WRITE_SIZE > $O/pmc_write_size_h2.txt 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch_size_h2.txt $O/pmc_write_size_h2.txt vn_gemm_x3 $O/traffic_h2.json > /dev/null 2>&1
python scripts/pmc_summary.py /tmp/pf FETCH_SIZE > $O/pmc_fetch_size.txt 2>&1
python scripts/pmc_summary.py /tmp/pw WRITE_SIZE > $O/pmc_write_size.txt 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt vn_gemm_x3 $O/traffic_x3.json > /dev/null 2>&1
python scripts/pmc_per_kernel.py /tmp/psq > $O/pmc_lds_mfma_per_kernel.txt 2>&1
head -14 $O/last_vamp_kernel_stats.txt; cat $O/traffic_x3.json | head -12; head -14 $O/pmc_lds_mfma_per_kernel.txt
bash scripts/gpu_model_clock.sh $TAG/clock --no-alt > $O/clock.txt 2>&1; tail -12 $O/clock.txt
timeout 200 python scripts/power_trace.py $O/power_probe -- scripts/ubench/mfma_power_probe > $O/mfma_power_probe.txt 2>&1; cat $O/mfma_power_probe.txt | grep -E "^mode|^--" ; tail -5 $O/power_probe.txt
grep -v "^  +" $O/config1_forward_timeline.txt | head -16
bash scripts/gpu_codec_trace.sh $TAG/codec bf16x3 > $O/codec_trace.log 2>&1; tail -3 $O/codec_trace.log
