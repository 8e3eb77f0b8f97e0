#!/bin/bash
# effective shader clock under the fp32 MFMA GEMM: GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md "DVFS give-back")
mkdir -p gpurun_out/clk
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pk
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pk -o g -- python $R/scripts/gemm_pmc_driver.py > /dev/null 2> $R/gpurun_out/clk/err.txt
cd $R
python - <<'PY'
import csv, glob, collections
ct = glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
kt = glob.glob("/tmp/pk/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
out = []
for r in csv.DictReader(open(ct[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "vn_gemm" not in r["Kernel_Name"]:
        continue
    d = dur.get(r["Dispatch_Id"])
    if d:
        out.append((r["Kernel_Name"][:40], r["Grid_Size"], float(r["Counter_Value"]), d[0]))
for name, grid, cyc, ns in out:
    print(f"{name:40s} grid={grid:>8s} GUI_ACTIVE={cyc:12.0f} dur_us={ns/1e3:9.1f} clock_GHz={cyc/ns:6.3f}")
PY
