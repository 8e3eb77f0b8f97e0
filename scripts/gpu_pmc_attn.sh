#!/bin/bash
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/scripts/attn_bench.py
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES"; do
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pm -o g -- python $R/scripts/attn_bench.py > /dev/null 2>&1
  python - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attention" not in row["Kernel_Name"]: continue
        a = acc[row["Grid_Size"]][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for g, cs in acc.items():
    print("attention grid", g, " ".join("%s=%.4g" % (c, v[1] / v[0]) for c, v in sorted(cs.items())))
PY
done
