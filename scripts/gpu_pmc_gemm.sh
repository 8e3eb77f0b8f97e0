#!/bin/bash
# PMC counters for the GEMM kernel on the 4096^3 and QKV shapes (one counter group per pass)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(SQ_|GRBM_|TCC_|TCP_)" | awk '{print $1}' | sort -u | head -300 > $R/gpurun_out/pmc/counters.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pm -o g -- python $R/scripts/gemm_pmc_driver.py > /dev/null 2> $R/gpurun_out/pmc/err_$tag.txt
  python $R/scripts/pmc_summary2.py /tmp/pm > $R/gpurun_out/pmc/sum_$tag.txt 2>&1
  cat $R/gpurun_out/pmc/sum_$tag.txt
done
