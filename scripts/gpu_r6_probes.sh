#!/bin/bash
# round 6: the three GEMM probes of VERDICT r5 item 2 on ONE box — (a) 256 x 256 block tile, (b) four waves / one per SIMD / 512 VGPRs,
# (c) product order in the pure matrix-pipe loop — with socket power next to (c) and the fabric bytes (FETCH_SIZE) of every arm of (a)/(b)
mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
cd $R
echo "== (c) product order, pure MFMA loop" 
python scripts/power_trace.py $O/r06_mfma_power_probe_power -- scripts/ubench/mfma_power_probe > $O/r06_mfma_power_probe.txt 2>&1
cat $O/r06_mfma_power_probe.txt | grep -E "^mode|^--"
cat $O/r06_mfma_power_probe_power.txt | tail -6
echo "== (a), (b) block tile A/B"
timeout 900 scripts/ubench/gemm_x3_tile_probe > $O/r06_gemm_tile_probe.txt 2>&1; echo rc=$?
cat $O/r06_gemm_tile_probe.txt
export TMPDIR=/tmp; cd /tmp
for arm in 0 1 2 3 4 5; do
  rm -rf /tmp/pf$arm
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf$arm -o f -- $R/scripts/ubench/gemm_x3_tile_probe $arm 5 > /dev/null 2> $O/r06_tile_probe_pmc_err_$arm.txt
  python $R/scripts/pmc_summary.py /tmp/pf$arm FETCH_SIZE > $O/r06_tile_probe_fetch_arm$arm.txt 2>&1
  head -5 $O/r06_tile_probe_fetch_arm$arm.txt
done
