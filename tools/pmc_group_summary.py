"""Per-launch-shape summary of a rocprofv3 --pmc pass: counter totals grouped by (kernel, grid, workgroup size), plus the
wave-cycle split the design discussion uses (parked at s_waitcnt / barrier, issue-stalled, issuing).
Usage: python tools/pmc_group_summary.py <counter_collection.csv> <kernel substring> [out.json]"""
import csv
import json
import sys
from collections import defaultdict

groups = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(sys.argv[1], newline="") as fh:
    for row in csv.DictReader(fh):
        if sys.argv[2] not in row.get("Kernel_Name", ""):
            continue
        key = f'{row.get("Kernel_Name", "")[:60]} grid={row.get("Grid_Size")} wg={row.get("Workgroup_Size")}'
        groups[key][row["Counter_Name"]] += float(row.get("Counter_Value", 0) or 0)
        disp[key].add(row.get("Dispatch_Id"))
out = {}
for key, c in groups.items():
    d = {"dispatches": len(disp[key]), "counters": dict(c)}
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                d[k + "_over_wave_cycles"] = round(c[k] / wc, 4)
    if c.get("SQ_INSTS_MFMA"):
        d["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"], 3)
    if c.get("SQ_BUSY_CU_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        d["mfma_busy_over_cu_busy_div4"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"] / 4, 4)
    out[key] = d
text = json.dumps(out, indent=1)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text)
print(text)
