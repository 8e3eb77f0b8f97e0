#!/bin/bash
# A/B of two builds of the library on ONE box: the in-tree libvampnet_hip.so vs vampnet_amd/libvampnet_hip_prev.so (VN_LIB), headline command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2 3; do
  for which in new prev; do
    if [ $which = prev ]; then export VN_LIB=$R/vampnet_amd/libvampnet_hip_prev.so; else unset VN_LIB; fi
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-sharded-check > $O/r06_ab_${which}_$i.json 2> $O/r06_ab_${which}_$i.err
    python - "$O/r06_ab_${which}_$i.json" <<'PY'
import json, sys
d = json.loads(next(l for l in open(sys.argv[1]) if l.startswith("{")))
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 2), round(d["value"]), round(d["roofline"]["frac"], 4))
PY
  done
done
