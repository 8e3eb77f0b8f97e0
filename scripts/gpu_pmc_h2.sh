#!/bin/bash
# PMC counter groups (one per pass, no trace domains) of the f16x2 GEMM on the model's B = 8 shapes (tile height by shape) and of the
# attention kernels in both operand formats (per kernel instantiation).  $1 = output tag.
T=${1:-pmc_h2}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp VN_PMC_FMT=f16x2
R=$GRAFT_REPO_ROOT
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmx
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmx -o g -- python $R/scripts/gemm_x3_pmc_driver.py > /dev/null 2> $R/$O/err_gemm_$tag.txt
  python $R/scripts/pmc_summary2.py /tmp/pmx > $R/$O/gemm_$tag.txt 2>&1
done
cat $R/$O/gemm_*.txt > $R/$O/pmc_gemm_f16x2.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES"; do
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pm -o g -- python $R/scripts/attn_bench.py > /dev/null 2>&1
  python - >> $R/$O/pmc_attention.txt <<PY
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attention_x3" not in row["Kernel_Name"]: continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        a = acc[(name, row["Grid_Size"])][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for (name, g), cs in sorted(acc.items()):
    print(name, "grid", g, " ".join("%s=%.4g" % (c, v[1] / v[0]) for c, v in sorted(cs.items())))
PY
done
cd $R
tail -12 $O/pmc_gemm_f16x2.txt; grep "kernel<4, false, 2> grid 204800" $O/pmc_attention.txt
