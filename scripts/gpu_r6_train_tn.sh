#!/bin/bash
# round 6: the training step with and without the transposing passes of its dW GEMMs (VN_TRAIN_TN) on ONE box, + the tests that cover it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -x -q -m gpu > $O/r06_train_tn_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_train_tn_tests.log
for tn in 1 0 1 0; do
  VN_TRAIN_TN=$tn timeout 300 python bench.py --workload train --no-cpu-baseline > $O/r06_train_tn$tn.json 2> $O/r06_train_tn$tn.err
  python - "$O/r06_train_tn$tn.json" <<'PY'
import json, sys
d = json.loads(next(l for l in open(sys.argv[1]) if l.startswith("{")))
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["roofline"]["frac"])
PY
done
