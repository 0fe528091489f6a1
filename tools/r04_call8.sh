#!/bin/bash
set -u
bash tools/gpu_round.sh r04final benchfull benchprof
OUT=gpurun_out/r04final
timeout 600 python bench.py --model qwen3-8b --no-cpu-baseline > $OUT/bench_8b.json 2> $OUT/bench_8b.err; echo "bench 8b rc=$?"; cut -c1-300 $OUT/bench_8b.json
