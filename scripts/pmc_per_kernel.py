"""LDS / VMEM / VALU instructions per MFMA, per kernel name, from one rocprofv3 --pmc pass (counters: SQ_INSTS_VALU_MFMA_MOPS_BF16 — or
SQ_INSTS_MFMA —, SQ_INSTS_LDS, SQ_INSTS_VMEM, SQ_INSTS_VALU, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT).  Instruction counters count per
WAVE-instruction; a 32x32x16 bf16 MFMA is 512 "MOPS" units when the MOPS counter is the one collected (the script normalises both ways and
prints the raw sums, so the ratios can be recomputed).  A fragment read of gemm_x3.hip / attention_x3.hip is one ds_read_b128 (1 KiB per wave),
an LDS-DMA piece one global_load_lds_dwordx4 (1 KiB per wave): LDS KiB read per MFMA ~ SQ_INSTS_LDS / MFMA, DMA KiB per MFMA ~ SQ_INSTS_VMEM / MFMA
(the k-loop dominates both counts).  usage: pmc_per_kernel.py <rocprof dir> [name filter ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
filters = sys.argv[2:] or ["vn_gemm_x3", "vn_attention_x3", "vn_rowprep"]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]
        if not any(x in name for x in filters):
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name].add(r["Dispatch_Id"])
print(f"{'kernel':56s} {'launches':>8s} {'MFMA/launch':>12s} {'LDS inst/MFMA':>13s} {'VMEM inst/MFMA':>14s} {'VALU(non-MFMA)/MFMA':>19s} {'bank-conflict cyc/LDS-active':>28s}")
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    n = len(cnt[name])
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    mfma = c.get("SQ_INSTS_MFMA", 0.0) or mops / 512.0
    if mfma <= 0:
        print(f"{name:56s} {n:8d}  (no MFMA counter)  raw: " + " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
        continue
    valu = c.get("SQ_INSTS_VALU", 0.0)
    print(f"{name:56s} {n:8d} {mfma / n:12.0f} {c.get('SQ_INSTS_LDS', 0) / mfma:13.3f} {c.get('SQ_INSTS_VMEM', 0) / mfma:14.3f} "
          f"{max(valu - mfma, 0) / mfma:19.3f} {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1):28.4f}")
    print(f"{'':56s} raw sums: " + " ".join(f"{k}={v:.5g}" for k, v in sorted(c.items())))
