#!/bin/bash
set -u
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP_SHAPES=8b_qkv,8b_o,8b_down,8b_gate_up,32b_qkv,32b_o,32b_down,32b_gate_up timeout 900 python tools/gemm_wide_sweep.py 131 144 > $OUT/sweep_m131_m144.jsonl 2> $OUT/sweep.err; echo "sweep rc=$?"
SWEEP_SHAPES=8b_qkv,8b_o,8b_down,8b_gate_up,32b_qkv,32b_o,32b_down,32b_gate_up timeout 900 python tools/gemm_wide_sweep.py 64 > $OUT/sweep_m64.jsonl 2>> $OUT/sweep.err; echo "sweep64 rc=$?"
python - <<'P'
import json,glob
from collections import defaultdict
for f in sorted(glob.glob('gpurun_out/r04m/sweep_*.jsonl')):
    rows=[json.loads(l) for l in open(f) if l.startswith('{')]
    g=defaultdict(list); pick={}
    for r in rows:
        if 'us' in r: g[(r['shape'],r['m'])].append(r)
        else: pick[(r['shape'],r['m'])]=r.get('planner_split')
    for k,v in g.items():
        v.sort(key=lambda r:r['us'])
        print(k,'planner split',pick.get(k),'best',[(r['nt'],r['nw'],r['split'],r['us']) for r in v[:4]])
P
BENCH_M=64,131,144 timeout 600 python tools/gemm_wide_bench.py 8b 32b > $OUT/gemm_wide_now.json 2> $OUT/gemm_wide_now.err; python -c "
import json; d=json.load(open('$OUT/gemm_wide_now.json')); [print(k,v) for k,v in d['time_us'].items()]"
