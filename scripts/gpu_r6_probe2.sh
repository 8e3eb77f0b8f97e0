#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 scripts/ubench/gemm_x3_tile_probe > $O/r06_gemm_tile_probe.txt 2>&1; echo rc=$?
grep -E "^==|arm 1|arm 2|arm 4|arm 5|WRONG|skipped" $O/r06_gemm_tile_probe.txt
echo "== bench default (sharded check on)"; timeout 900 python bench.py --steps 5 --warmup 2 > $O/r06_bench_n1_check.json 2> $O/r06_bench_n1_check.err; echo rc=$?; tail -3 $O/r06_bench_n1_check.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_bench_n1_check.json"))
print("value", round(d["value"]), "ms", round(d["ms_per_step"],2), "frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"], "|", d["roofline"]["traffic_source"][:90])
print("devices", json.dumps(d["devices"])[:900])
print("cpu", {k: (v if k!='sample' else v[:60]) for k,v in d["cpu_baseline"].items() if k in ("value","cores","kind","port_skips")}, "b8", d["cpu_baseline"].get("batch8",{}).get("value"), d["cpu_baseline"].get("batch8",{}).get("thread_sweep"))
PY
echo "== two ranks on one GPU (gloo exchange)"; VN_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $O/r06_bench_gpus2_one_gpu.json 2> $O/r06_bench_gpus2.err; echo rc=$?; tail -2 $O/r06_bench_gpus2.err; cut -c1-300 $O/r06_bench_gpus2_one_gpu.json
