#!/bin/bash
# round 4, call 6: persistent prefill, group walk (NVL_PREFILL_PERSIST=2): tests + A/B against off and the snake walk
set -u
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
NVL_PREFILL_PERSIST=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -k "prefill" > $OUT/pytest_prefill_persist2.log 2>&1; echo "prefill tests (group walk) rc=$?"; tail -2 $OUT/pytest_prefill_persist2.log
NVL_PREFILL_PERSIST=2 timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -rf -k "tiny_model_greedy or chunked or other_head" > $OUT/pytest_e2e_persist2.log 2>&1; echo "e2e (group walk) rc=$?"; tail -2 $OUT/pytest_e2e_persist2.log
for p in 0 2 1 0 2 1; do
  NVL_PREFILL_PERSIST=$p timeout 300 python tools/prefill_bench.py > $OUT/prefill_persist${p}_$RANDOM.json 2> /dev/null
done
python - <<'P'
import json,glob
from collections import defaultdict
r=defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r04f/prefill_persist*_*.json')):
    p=f.split('persist')[1][0]
    for c in json.load(open(f))['cases']: r[c['name']][p].append(c['TFLOPs'])
for n,v in r.items(): print(n.ljust(40),'off',v['0'],'snake',v['1'],'group',v['2'])
P
