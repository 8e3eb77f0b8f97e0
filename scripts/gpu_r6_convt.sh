#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for w in 512 384 192 96; do
  VN_CODEC_CONVT_MIN_WORK=$w timeout 400 python bench.py --e2e --no-cpu-baseline --steps 3 --warmup 1 > $O/r06_convt_e2e_w$w.json 2> $O/r06_convt_e2e_w$w.err
  python - "$O/r06_convt_e2e_w$w.json" <<'PY'
import json, sys
d = json.loads(next(l for l in open(sys.argv[1]) if l.startswith("{")))
g = d["codec_roofline"]["groups"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 2), d["config"]["stages_ms"], {k: (round(v["ms_per_step"], 2), round(v["frac"], 3)) for k, v in g.items() if v})
PY
done
