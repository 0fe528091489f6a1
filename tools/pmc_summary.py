"""Aggregate a rocprofv3 --pmc counter_collection CSV by (kernel, counter): sum, dispatches, mean.
Usage: python tools/pmc_summary.py <dir-or-csv> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
agg = defaultdict(lambda: [0.0, 0])
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-80:]
            key = (name, row.get("Counter_Name", "?"))
            agg[key][0] += float(row.get("Counter_Value", 0) or 0)
            agg[key][1] += 1
out = [dict(kernel=k[0], counter=k[1], sum=v[0], dispatches=v[1], mean=v[0] / max(v[1], 1)) for k, v in sorted(agg.items())]
text = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
