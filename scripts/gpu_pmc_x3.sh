#!/bin/bash
# PMC counters + effective clock for the split-plane GEMM kernels (one counter group per pass; no trace domains besides
# --kernel-trace in the clock pass).  VN_PMC_KERNEL=h2 profiles the staged f16x2 kernel.  Output: gpurun_out/pmc_x3/
mkdir -p gpurun_out/pmc_x3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmx
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmx -o g -- python $R/scripts/gemm_x3_pmc_driver.py > /dev/null 2> $R/gpurun_out/pmc_x3/err_$tag.txt
  python $R/scripts/pmc_summary2.py /tmp/pmx > $R/gpurun_out/pmc_x3/sum_$tag.txt 2>&1
  cat $R/gpurun_out/pmc_x3/sum_$tag.txt
done
