#!/bin/bash
# Round 2, evidence pass on the final build (192-row tile, tiled operand planes, device build_mask, fused reduce+norm): the whole `-m gpu` suite, smoke(),
# the headline bench line (with the CPU leg), bench lines for configs[1] / B = 1, 2, 4 / f32 / training, rocprofv3 kernel stats +
# FETCH_SIZE / WRITE_SIZE passes of the headline command, PMC groups of the GEMM kernel on the model's shapes.
O=gpurun_out/r2final3
mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket" >> $O/gpu_info.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/1_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -3 $O/1_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/2_smoke.log
timeout 600 python bench.py > $O/3_bench_n1.json 2> $O/3_bench_n1.err; echo "bench rc=$?"; head -c 400 $O/3_bench_n1.json; echo
timeout 300 python bench.py --config 1 --steps 10 --warmup 3 > $O/3_bench_n1_config1.json 2> $O/3_bench_n1_config1.err; head -c 300 $O/3_bench_n1_config1.json; echo
for b in 1 2 4; do timeout 300 python bench.py --batch-per-gpu $b --steps 5 --warmup 2 --no-cpu-baseline > $O/3_bench_vamp_b$b.json 2> $O/3_bench_vamp_b$b.err; head -c 250 $O/3_bench_vamp_b$b.json; echo; done
timeout 300 python bench.py --dtype f32 --no-cpu-baseline > $O/3_bench_n1_f32.json 2> $O/3_bench_n1_f32.err; head -c 300 $O/3_bench_n1_f32.json; echo
timeout 300 python bench.py --workload train --no-cpu-baseline > $O/3_bench_train_n1.json 2> $O/3_bench_train_n1.err; head -c 300 $O/3_bench_train_n1.json; echo
bash scripts/gpu_bench_prof.sh r2final3 > $O/4_prof.txt 2>&1; tail -30 $O/4_prof.txt
bash scripts/gpu_pmc_x3.sh 0 > $O/5_pmc_gemm_x3.txt 2>&1; tail -12 $O/5_pmc_gemm_x3.txt
