"""Effective shader clock and wait split of the prefill attention kernel from one rocprofv3 pass that collects
`--pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA` together
with `--kernel-trace` (dispatch durations): per dispatch shape (grid), GRBM_GUI_ACTIVE / duration = cycles the chip was
active per second of the kernel = the clock it ran at (MI355X_MICROARCH.md, DVFS give-back); the SQ counters are in
quad-cycles. Usage: python tools/pmc_prefill_clock.py <dir with *counter_collection.csv and *kernel_trace.csv> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in kt:
    for row in csv.DictReader(open(f, newline="")):
        if "prefill_" in row.get("Kernel_Name", ""):
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row.get("Grid_Size", row.get("Grid_Size_X", "")))
by = defaultdict(lambda: defaultdict(float))
grid_of = {}
for f in cc:
    for row in csv.DictReader(open(f, newline="")):
        if "prefill_" not in row.get("Kernel_Name", ""):
            continue
        key = row.get("Grid_Size", "")
        by[(key, row["Dispatch_Id"])][row["Counter_Name"]] += float(row.get("Counter_Value", 0) or 0)
out = defaultdict(lambda: defaultdict(float))
for (grid, did), c in by.items():
    o = out[grid]
    o["dispatches"] += 1
    for k, v in c.items():
        o[k] += v
    if did in dur:
        o["duration_ns"] += dur[did][0]
res = {}
for grid, o in out.items():
    r = dict(o)
    if o.get("duration_ns") and o.get("GRBM_GUI_ACTIVE"):
        r["clock_GHz_if_counter_is_per_device"] = o["GRBM_GUI_ACTIVE"] / o["duration_ns"]
        r["clock_GHz_if_counter_sums_8_xcds"] = o["GRBM_GUI_ACTIVE"] / 8 / o["duration_ns"]
    w = o.get("SQ_WAVE_CYCLES", 0)
    if w:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            r[k + "_over_wave_cycles"] = round(o.get(k, 0) / w, 4)
    if o.get("SQ_INSTS_MFMA"):
        r["valu_per_mfma"] = round(o.get("SQ_INSTS_VALU", 0) / o["SQ_INSTS_MFMA"], 3)
    res["grid=" + str(grid)] = r
text = json.dumps(res, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
