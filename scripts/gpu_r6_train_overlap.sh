#!/bin/bash
# round 6: the layers' dW GEMMs on a side stream (VN_TRAIN_OVERLAP=1) vs in the caller's stream (=0): tests (incl. the guard-page children),
# then the step on ONE box, then the kernel trace of the last overlapped step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_guard.py -x -q -m gpu > $O/r06_train_overlap_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/r06_train_overlap_tests.log | tail -2
for ov in 1 0 1 0; do
  VN_TRAIN_OVERLAP=$ov timeout 300 python bench.py --workload train --no-cpu-baseline > $O/r06_train_ov$ov.json 2> $O/r06_train_ov$ov.err
  python - "$O/r06_train_ov$ov.json" <<'PY'
import json, sys
d = json.loads(next(l for l in open(sys.argv[1]) if l.startswith("{")))
r = d["roofline"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 2), round(d["value"]), "frac", round(r["frac"], 3), "overlapped brackets:", (r.get("overlapped") or {}).get("achieved"), (r.get("overlapped") or {}).get("serial_step_ms"))
PY
done
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/po
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/po -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/r06_trace_train_ov.err
for f in $(find /tmp/po -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 30 > $O/r06_train_last_step_kernel_stats_overlap.txt 2>&1; done
head -12 $O/r06_train_last_step_kernel_stats_overlap.txt | cut -c1-150
