#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_codec.py -q -x -s -p no:cacheprovider > $O/r06_convt_tests.log 2>&1; echo "codec tests rc=$?"; grep -E "passed|failed|^FAILED|Error|error" $O/r06_convt_tests.log | tail -8 | cut -c1-250
for f in 1 0; do
  VN_X3_CONVT=$f timeout 400 python bench.py --e2e --no-cpu-baseline --steps 3 --warmup 1 > $O/r06_convt_e2e_$f.json 2> $O/r06_convt_e2e_$f.err
  python - "$O/r06_convt_e2e_$f.json" <<'PY'
import json, sys
d = json.loads(next(l for l in open(sys.argv[1]) if l.startswith("{")))
g = d["codec_roofline"]["groups"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 2), d["config"]["stages_ms"], {k: (round(v["ms_per_step"], 2), round(v["frac"], 3)) for k, v in g.items() if v})
PY
done
bash scripts/gpu_codec_trace.sh r06_convt_codec bf16x3 > /dev/null 2>&1; tail -42 $O/r06_convt_codec/codec_kernel_trace.txt | cut -c1-110
