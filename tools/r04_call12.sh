#!/bin/bash
set -u
OUT=gpurun_out/r04j; mkdir -p $OUT; export TMPDIR=/tmp
for t in 1 0 1 0; do
  NVL_WIDE_TUNED=$t timeout 600 python bench.py --model qwen3-8b --workload prefix --no-cpu-baseline --no-roofline --warmup 1 > $OUT/cfg3_tuned${t}_$RANDOM.json 2> /dev/null; echo "cfg3 tuned=$t rc=$?"
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04j/cfg3_tuned*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), d['config']['host_seconds_in_last_step'] if 'host_seconds_in_last_step' in d['config'] else '')
P
