#!/bin/bash
# instruction mix (MFMA / LDS / VMEM / VALU) and LDS bank conflicts per kernel of one training step (its own --pmc pass, no trace domains); $1 = tag
TAG=${1:-trpmc}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/ptq
timeout 400 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/ptq -o sq -- python $R/bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $R/$O/pmc_sq.err
cd $R
python scripts/pmc_per_kernel.py /tmp/ptq > $O/pmc_train_per_kernel.txt 2>&1
head -30 $O/pmc_train_per_kernel.txt | cut -c1-200
