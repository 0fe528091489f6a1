#!/bin/bash
set -u
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP_SHAPES=8b_qkv,8b_o,8b_down,8b_gate_up timeout 600 python tools/gemm_wide_sweep.py 256 > $OUT/sweep_8b_m256.jsonl 2> $OUT/sweep_8b_m256.err; echo "sweep 8b rc=$?"
SWEEP_SHAPES=32b_tp8_qkv,32b_tp8_gate_up,32b_tp8_down timeout 600 python tools/gemm_wide_sweep.py 131 > $OUT/sweep_tp8_m131.jsonl 2> $OUT/sweep_tp8_m131.err; echo "sweep tp8 rc=$?"
python - <<'P'
import json
from collections import defaultdict
for f in ('gpurun_out/r04g/sweep_8b_m256.jsonl','gpurun_out/r04g/sweep_tp8_m131.jsonl'):
    rows=[json.loads(l) for l in open(f) if l.startswith('{')]
    g=defaultdict(list); pick={}
    for r in rows:
        if 'us' in r: g[(r['shape'],r['m'])].append(r)
        else: pick[(r['shape'],r['m'])]=r.get('planner_split')
    for k,v in g.items():
        v.sort(key=lambda r:r['us'])
        print(k,'planner split',pick.get(k),'best',[(r['nt'],r['nw'],r['split'],r['us']) for r in v[:4]])
P
