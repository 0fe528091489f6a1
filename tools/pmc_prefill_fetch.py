"""HBM read traffic of prefill_attn_kernel from a rocprofv3 `--pmc FETCH_SIZE` pass over tools/prefill_bench.py:
FETCH_SIZE (KiB, x2 on gfx950 for 16 B/lane streaming reads — MI355X_MICROARCH.md, HBM section) per dispatch, against
the algorithmic reads of the same launch (Q + K + V once: 2 B x 128 x tokens x (Hq + 2 Hkv)).
usage: python tools/pmc_prefill_fetch.py <counter_collection.csv> [out.json]"""
import csv
import json
import sys

# the launches tools/prefill_bench.py makes, in order: one correctness call, then 13 calls (3 warm + 10 timed) per case
import random
random.seed(0)                      # the same ragged batches tools/prefill_bench.py draws
_ragged = []
while sum(_ragged) + 1024 <= 16384:
    _ragged.append(random.randint(100, 1024))
_short = [random.randint(100, 180) for _ in range(110)]
CASES = [("spot check 300+129+64, 8/2 heads", 493, 8, 2, 1), ("0.6B 16x1024", 16384, 16, 8, 13),
         ("bench-like 29x561", 29 * 561, 16, 8, 13), (f"bench ragged U[100,1024] x{len(_ragged)}", sum(_ragged), 16, 8, 13),
         ("110 short U[100,180]", sum(_short), 16, 8, 13), ("0.6B 4x4096", 16384, 16, 8, 13),
         ("32B/TP8 1x16384 (config 5)", 16384, 8, 1, 13), ("0.6B 1x16384", 16384, 16, 8, 13),
         ("32B 8x2048 G=8", 16384, 64, 8, 13)]
rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        if "prefill_attn" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE":
            rows.append((int(r.get("Dispatch_Id", 0)), float(r.get("Counter_Value", 0) or 0)))
rows.sort()
out, i = {"dispatches": len(rows), "cases": []}, 0
for name, tokens, hq, hkv, n in CASES:
    part = [v for _, v in rows[i:i + n]]
    i += n
    if not part:
        continue
    alg = 2.0 * 128 * tokens * (hq + 2 * hkv)
    hbm = sum(part) / len(part) * 1024 * 2
    out["cases"].append(dict(name=name, dispatches=len(part), fetch_bytes_per_launch=hbm, algorithmic_read_bytes=alg,
                             ratio=round(hbm / alg, 3)))
text = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
