#!/bin/bash
set -u
OUT=gpurun_out/r04n; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP_SHAPES=8b_down,32b_qkv,32b_down,32b_o timeout 900 python tools/gemm_wide_sweep.py 32 80 96 112 128 > $OUT/sweep_m32_m128.jsonl 2> $OUT/sweep.err; echo "sweep rc=$?"
python - <<'P'
import json,glob
from collections import defaultdict
rows=[json.loads(l) for l in open('gpurun_out/r04n/sweep_m32_m128.jsonl') if l.startswith('{')]
g=defaultdict(list)
for r in rows:
    if 'us' in r: g[(r['shape'],r['m'])].append(r)
for k,v in g.items():
    v.sort(key=lambda r:r['us'])
    print(k,'best',[(r['nt'],r['nw'],r['split'],r['us']) for r in v[:5]])
P
BENCH_M=32,80,96,112,128 timeout 600 python tools/gemm_wide_bench.py 8b 32b > $OUT/gemm_wide_now.json 2> $OUT/gemm_wide_now.err; python -c "
import json; d=json.load(open('$OUT/gemm_wide_now.json')); [print(k,v[:2]) for k,v in d['time_us'].items() if 'down' in k or '32b_qkv' in k or '32b_o' in k]"
