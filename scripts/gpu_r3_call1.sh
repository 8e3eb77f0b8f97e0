#!/bin/bash
# Round 3, call 1: the restructured bf16x3 attention (bias-initialised accumulator, deferred max, key-split decomposition), per-context
# tuning state, and the new bench modes (self-launched ranks, seed-exact RNG line, whole-request line, in-model clock).
O=gpurun_out/r3c1
mkdir -p $O
python -c "import torch; print('devices', torch.cuda.device_count())" > $O/0_devices.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention or two_contexts or bf16x3 or splitk" > $O/1_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/1_kernels.log
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -q -m gpu -x > $O/1_bf16x3.log 2>&1; echo "bf16x3 rc=$?"; tail -3 $O/1_bf16x3.log
timeout 300 python scripts/attn_bench.py > $O/2_attn_bench.txt 2>&1; cat $O/2_attn_bench.txt
timeout 300 python bench.py --no-cpu-baseline > $O/3_bench_n1.json 2> $O/3_bench_n1.err; head -c 300 $O/3_bench_n1.json; echo
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/3_bench_config1.json 2> $O/3_bench_config1.err; head -c 300 $O/3_bench_config1.json; echo
for b in 1 2 4; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline > $O/3_bench_vamp_b$b.json 2> $O/3_bench_vamp_b$b.err; head -c 250 $O/3_bench_vamp_b$b.json; echo; done
VN_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/4_bench_gpus2_one_gpu.json 2> $O/4_bench_gpus2.err; echo "gpus2 rc=$?"; head -c 400 $O/4_bench_gpus2_one_gpu.json; echo; tail -3 $O/4_bench_gpus2.err
timeout 300 python bench.py --rng torch_device --no-cpu-baseline --steps 3 > $O/4_bench_rng_torch_device.json 2> $O/4_bench_rng.err; echo "rng rc=$?"; head -c 300 $O/4_bench_rng_torch_device.json; echo; tail -3 $O/4_bench_rng.err
timeout 400 python bench.py --e2e --no-cpu-baseline --steps 3 > $O/4_bench_e2e.json 2> $O/4_bench_e2e.err; echo "e2e rc=$?"; head -c 300 $O/4_bench_e2e.json; echo; tail -3 $O/4_bench_e2e.err
bash scripts/gpu_model_clock.sh r3c1/clock > $O/5_clock.txt 2>&1; tail -25 $O/5_clock.txt
