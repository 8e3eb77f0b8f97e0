#!/bin/bash
# PMC counters for the bf16x3 GEMM kernel (one counter group per pass; no trace domains).  $1 = tile height (128 / 256),
# $2 = 1 stream-K / 0 data-parallel.  Output: gpurun_out/pmc_x3_bm$1_sk$2/
BM=${1:-128}

O=gpurun_out/pmc_x3_bm${BM}
mkdir -p $O
export TMPDIR=/tmp VN_X3_BM=$BM
R=$GRAFT_REPO_ROOT
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmx
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmx -o g -- python $R/scripts/gemm_x3_pmc_driver.py > /dev/null 2> $R/$O/err_$tag.txt
  python $R/scripts/pmc_summary2.py /tmp/pmx > $R/$O/sum_$tag.txt 2>&1
  cat $R/$O/sum_$tag.txt
done
