#!/bin/bash
set -u
OUT=gpurun_out/r04o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf -k "linear_wide or full_width" > $OUT/pytest_wide.log 2>&1; echo "wide tests rc=$?"; tail -2 $OUT/pytest_wide.log
BENCH_M=64,131,144 timeout 600 python tools/gemm_wide_bench.py 8b 32b > $OUT/gemm_wide_tuned.json 2> $OUT/gemm_wide_tuned.err; python -c "
import json; d=json.load(open('$OUT/gemm_wide_tuned.json')); print(d['relerr_max']); [print(k,v[:2]) for k,v in d['time_us'].items() if 'down' in k or '32b_qkv' in k]"
for t in 1 0; do
  NVL_WIDE_TUNED=$t timeout 600 python bench.py --model qwen3-8b --no-cpu-baseline --no-roofline > $OUT/bench_8b_tuned$t.json 2> /dev/null; echo "8b tuned=$t rc=$?"; python -c "import json; d=json.load(open('$OUT/bench_8b_tuned$t.json')); print(round(d['value']), d['ms_per_step'])"
done
for t in 1 0; do
  NVL_WIDE_TUNED=$t timeout 600 python bench.py --model qwen3-32b --tp 1 --no-cpu-baseline --no-roofline --warmup 0 > $OUT/bench_32b_tuned$t.json 2> /dev/null; echo "32b tuned=$t rc=$?"; python -c "import json; d=json.load(open('$OUT/bench_32b_tuned$t.json')); print(round(d['value']), d['ms_per_step'])"
done
