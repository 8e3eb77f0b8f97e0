#!/bin/bash
# One-command evidence pass on the current build ($1 = tag, default r3final): the whole `-m gpu` suite, smoke(), the headline bench line
# (with the CPU leg), the other bench lines (configs[1], B = 1 / 2 / 4, f32, training, self-launched 2 ranks, seed-exact RNG, whole
# request), rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the headline command, the in-model clock / MFMA-occupancy
# pass, attention and codec micro-benchmarks with the per-launch codec trace.  Everything lands under gpurun_out/$TAG/.
TAG=${1:-r3final}
O=gpurun_out/$TAG
mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket" >> $O/gpu_info.txt
python -c "import torch; print('visible devices', torch.cuda.device_count())" >> $O/gpu_info.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu > $O/1_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -3 $O/1_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/2_smoke.log
timeout 600 python bench.py > $O/3_bench_n1.json 2> $O/3_bench_n1.err; echo "bench rc=$?"; head -c 400 $O/3_bench_n1.json; echo
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 > $O/3_bench_n1_config1.json 2> $O/3_bench_n1_config1.err; head -c 300 $O/3_bench_n1_config1.json; echo
for b in 1 2 4; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline > $O/3_bench_vamp_b$b.json 2> $O/3_bench_vamp_b$b.err; head -c 250 $O/3_bench_vamp_b$b.json; echo; done
timeout 300 python bench.py --dtype f32 --no-cpu-baseline > $O/3_bench_n1_f32.json 2> $O/3_bench_n1_f32.err; head -c 300 $O/3_bench_n1_f32.json; echo
timeout 300 python bench.py --dtype bf16x3 --no-cpu-baseline > $O/3_bench_n1_bf16x3.json 2> $O/3_bench_n1_bf16x3.err; head -c 300 $O/3_bench_n1_bf16x3.json; echo
timeout 300 python bench.py --workload train --no-cpu-baseline > $O/3_bench_train_n1.json 2> $O/3_bench_train_n1.err; head -c 300 $O/3_bench_train_n1.json; echo
VN_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/3_bench_gpus2_one_gpu.json 2> $O/3_bench_gpus2.err; echo "gpus2 rc=$?"; head -c 300 $O/3_bench_gpus2_one_gpu.json; echo
timeout 300 python bench.py --rng torch_device --no-cpu-baseline > $O/3_bench_rng_torch_device.json 2> $O/3_bench_rng.err; head -c 300 $O/3_bench_rng_torch_device.json; echo
timeout 400 python bench.py --e2e --no-cpu-baseline > $O/3_bench_e2e.json 2> $O/3_bench_e2e.err; head -c 300 $O/3_bench_e2e.json; echo
bash scripts/gpu_bench_prof.sh $TAG/prof > $O/4_prof.txt 2>&1; tail -30 $O/4_prof.txt
bash scripts/gpu_model_clock.sh $TAG/clock > $O/5_clock.txt 2>&1; tail -14 $O/5_clock.txt
timeout 300 python scripts/attn_bench.py > $O/6_attn_bench.txt 2>&1; cat $O/6_attn_bench.txt
timeout 300 python scripts/codec_bench.py > $O/7_codec_bench.txt 2>&1; cat $O/7_codec_bench.txt
bash scripts/gpu_codec_trace.sh $TAG/codec_trace f16x2 > /dev/null 2>&1; tail -2 $O/codec_trace/codec_kernel_trace.txt
