#!/bin/bash
# round 4, call 7: the measurement set — full bench line, rocprofv3 kernel trace of the bench, PMC traffic of the three
# decode-attention instantiations, Qwen3-8B on the bench workload
set -u
bash tools/gpu_round.sh r04final benchfull benchprof pmcg
OUT=gpurun_out/r04final
timeout 600 python bench.py --model qwen3-8b --no-cpu-baseline > $OUT/bench_8b.json 2> $OUT/bench_8b.err; echo "bench 8b rc=$?"; cut -c1-300 $OUT/bench_8b.json
