#!/bin/bash
# Round 2, GPU call 20: same-box A/B of the tile heights inside the model (rocprofv3 kernel stats of the headline bench, last vamp()).
O=gpurun_out/r2c20
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for bm in 0 128 192; do
  rm -rf /tmp/px3
  VN_X3_BM=$bm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px3 -o vamp -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/$O/bench_bm$bm.json 2> $R/$O/trace_bm$bm.err
  for f in $(find /tmp/px3 -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 24 20 > $R/$O/last_vamp_bm$bm.txt 2>&1; done
  echo "== VN_X3_BM=$bm"; head -11 $R/$O/last_vamp_bm$bm.txt | cut -c1-150
done
cd $R
for bm in 0 128 192; do VN_X3_BM=$bm timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/plain_bm$bm.json 2>/dev/null; python - <<PY
import json
b = json.load(open("$O/plain_bm$bm.json"))
print("VN_X3_BM=$bm", round(b["value"]), "tok/s", round(b["ms_per_step"], 1), "ms; GEMM frac", round(b["roofline"]["frac"], 3))
PY
done
