#!/bin/bash
# copy what is to be judged from gpurun_out/$1 (scripts/gpu_evidence.sh) to profiles/$2_* (tracked)
S=gpurun_out/${1:-r06}; P=profiles/${2:-r06}
for f in pytest_gpu.log smoke.log gpu_info.txt bench_n1.json bench_n1_config1.json bench_vamp_b1.json bench_vamp_b2.json bench_vamp_b4.json bench_n1_f32.json \
         bench_rng_torch_device.json bench_e2e.json bench_train_n1.json bench_train_n1_f32_mfma.json bench_gpus2_one_gpu.json bench_under_rocprof.json \
         kernel_stats.csv last_vamp_kernel_stats.txt train_last_step_kernel_stats.txt pmc_fetch_size.txt pmc_write_size.txt traffic_x3.json \
         pmc_fetch_size_h2.txt pmc_write_size_h2.txt traffic_h2.json pmc_lds_mfma_per_kernel.txt power_bench.txt power_bench.csv; do
  [ -f $S/$f ] && cp $S/$f ${P}_$f
done
[ -f $S/clock/model_clock.txt ] && cp $S/clock/model_clock.txt ${P}_model_clock.txt
[ -f $S/codec/codec_kernel_trace.txt ] && cp $S/codec/codec_kernel_trace.txt ${P}_codec_kernel_trace_bf16x3.txt
# bench lines: keep the JSON line only (stderr chatter of a library that prints to stdout would break a parser)
for j in ${P}_bench_*.json; do python - "$j" <<'PY'
import sys
p = sys.argv[1]
lines = [l for l in open(p) if l.startswith("{")]
if lines:
    open(p, "w").write(lines[-1])
PY
done
ls ${P}_* | wc -l
