#!/bin/bash
# round 6: per-kernel times of the LAST training step with the dW GEMMs token-major (VN_TRAIN_TN=1) and transposed (=0), same box; then the
# LDS bank-conflict / instruction counters of the TN step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for tn in 1 0; do
  rm -rf /tmp/pt$tn
  VN_TRAIN_TN=$tn timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt$tn -o train -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/r06_trace_train_tn$tn.err
  for f in $(find /tmp/pt$tn -name "*kernel_trace.csv"); do python $R/scripts/kstats_last_step.py $f vn_embed_kernel 30 > $O/r06_train_last_step_kernel_stats_tn$tn.txt 2>&1; cp $f $O/r06_train_trace_tn$tn.csv; done
  head -14 $O/r06_train_last_step_kernel_stats_tn$tn.txt | cut -c1-150
done
rm -rf /tmp/pc
VN_TRAIN_TN=1 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d /tmp/pc -o c -- python $R/bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/r06_pmc_train_tn.err
python $R/scripts/pmc_per_kernel.py /tmp/pc > $O/r06_pmc_train_tn_per_kernel.txt 2>&1; head -30 $O/r06_pmc_train_tn_per_kernel.txt | cut -c1-220
