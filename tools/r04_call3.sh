#!/bin/bash
# round 4, call 3: skinny-GEMM decomposition sweep (0.6B shapes), planned decode attention with a smaller minimum share
# per wave on the one-kv-head shape (rocprofv3), the wide GEMM's shipped 64-column rule re-checked + the 8B lm_head at
# 208 / 256 rows, config 3 with the new rule
set -u
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python tools/gemm_skinny_sweep.py 64 131 208 > $OUT/skinny_sweep.jsonl 2> $OUT/skinny_sweep.err; echo "skinny sweep rc=$?"
python - <<'P'
import json
rows=[json.loads(l) for l in open('gpurun_out/r04c/skinny_sweep.jsonl') if l.startswith('{')]
from collections import defaultdict
g=defaultdict(list)
for r in rows: g[(r['shape'],r['m'])].append(r)
for k,v in g.items():
    rule=[r for r in v if r['plan']=='rule'][0]['us']
    best=sorted(v,key=lambda r:r['us'])[:3]
    print(k,'rule',rule,'best',[(b['plan'],b['splits'],b['us']) for b in best])
P
for mt in 4 2; do
  (cd /tmp && rm -rf /tmp/prof_g8_$mt && NVL_DECODE_MIN_TILES=$mt timeout 600 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_g8_$mt -o g8 -- python $REPO/tools/attn_replay.py --fused --hq 8 --hkv 1 --layers 64 --every 16 > $REPO/$OUT/replay_g8_min$mt.json 2> $REPO/$OUT/replay_g8_min$mt.err; echo "prof g8 min_tiles=$mt rc=$?")
  f=$(find /tmp/prof_g8_$mt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/replay_g8_min${mt}_kernel_stats.csv && head -3 $OUT/replay_g8_min${mt}_kernel_stats.csv | cut -c1-60,140-220
done
for mt in 4 2; do
  (cd /tmp && rm -rf /tmp/prof_g2_$mt && NVL_DECODE_MIN_TILES=$mt timeout 600 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_g2_$mt -o g2 -- python $REPO/tools/attn_replay.py --fused --every 16 > $REPO/$OUT/replay_g2_min$mt.json 2> $REPO/$OUT/replay_g2_min$mt.err; echo "prof g2 min_tiles=$mt rc=$?")
  f=$(find /tmp/prof_g2_$mt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/replay_g2_min${mt}_kernel_stats.csv && head -3 $OUT/replay_g2_min${mt}_kernel_stats.csv | cut -c1-60,140-220
done
NVL_DECODE_MIN_TILES=2 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "paged_attn_decode and not fp8" 2>&1 | tail -2
timeout 600 python tools/gemm_wide_m256.py 208 256 > $OUT/gemm_wide_m256_rule.json 2> $OUT/gemm_wide_m256_rule.err; echo "m256 rc=$?"; grep -v amdgpu.ids $OUT/gemm_wide_m256_rule.err | tail -24
timeout 600 python bench.py --model qwen3-8b --workload prefix --no-cpu-baseline --no-roofline --warmup 0 > $OUT/cfg3.json 2> $OUT/cfg3.err; echo "cfg3 rc=$?"; cut -c1-200 $OUT/cfg3.json; echo
