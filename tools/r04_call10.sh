#!/bin/bash
set -u
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP_SHAPES=32b_qkv,32b_o,32b_down timeout 900 python tools/gemm_wide_sweep.py 256 > $OUT/sweep_32b_m256.jsonl 2> $OUT/sweep_32b_m256.err; echo "sweep 32b rc=$?"
SWEEP_SHAPES=8b_qkv,8b_o,8b_down,8b_gate_up timeout 600 python tools/gemm_wide_sweep.py 208 > $OUT/sweep_8b_m208.jsonl 2> $OUT/sweep_8b_m208.err; echo "sweep 8b 208 rc=$?"
SWEEP_SHAPES=32b_tp8_qkv,32b_tp8_gate_up,32b_tp8_down timeout 600 python tools/gemm_wide_sweep.py 256 > $OUT/sweep_tp8_m256.jsonl 2> $OUT/sweep_tp8_m256.err; echo "sweep tp8 256 rc=$?"
python - <<'P'
import json,glob
from collections import defaultdict
for f in sorted(glob.glob('gpurun_out/r04h/sweep_*.jsonl')):
    rows=[json.loads(l) for l in open(f) if l.startswith('{')]
    g=defaultdict(list); pick={}
    for r in rows:
        if 'us' in r: g[(r['shape'],r['m'])].append(r)
        else: pick[(r['shape'],r['m'])]=r.get('planner_split')
    for k,v in g.items():
        v.sort(key=lambda r:r['us'])
        print(k,'planner split',pick.get(k),'best',[(r['nt'],r['nw'],r['split'],r['wgs'],r['us']) for r in v[:5]])
P
