"""Summarise a rocprofv3 --pmc pass over prefill_attn_kernel: per-counter totals and the derived ratios the design
discussion uses (MFMA busy fraction of the CU-busy cycles, VALU instructions per MFMA instruction, LDS bank-conflict
share of LDS-active cycles). Usage: python tools/pmc_prefill_summary.py <counter_collection.csv> [out.json]"""
import csv
import json
import sys
from collections import defaultdict

tot = defaultdict(float)
disp = set()
with open(sys.argv[1], newline="") as fh:
    for row in csv.DictReader(fh):
        if "prefill_attn" not in row.get("Kernel_Name", ""):
            continue
        tot[row["Counter_Name"]] += float(row.get("Counter_Value", 0) or 0)
        disp.add(row.get("Dispatch_Id"))
g = lambda k: tot.get(k, 0.0)
out = {"dispatches": len(disp), "counters": dict(tot)}
if g("SQ_BUSY_CU_CYCLES"):
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD-issue; SQ_BUSY_CU_CYCLES per CU (MI355X_MICROARCH.md units note)
    out["mfma_busy_over_cu_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CU_CYCLES")
if g("SQ_INSTS_MFMA"):
    out["valu_insts_per_mfma_inst"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA")
if g("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
text = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
