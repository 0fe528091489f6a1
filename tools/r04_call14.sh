#!/bin/bash
set -u
OUT=gpurun_out/r04l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/blas_tune_prefill.py $OUT/hipblaslt_prefill_gfx950.csv 16000 > $OUT/blas_tune_prefill.json 2> $OUT/blas_tune_prefill.err; echo "tune rc=$?"; grep -v amdgpu.ids $OUT/blas_tune_prefill.err | tail -10
ls -la $OUT/; head -5 $OUT/hipblaslt_prefill_gfx950.csv* | cut -c1-300
mkdir -p nano_vllm_amd/tuned; for f in $OUT/hipblaslt_prefill_gfx950.csv*; do cp $f nano_vllm_amd/tuned/hipblaslt_prefill_gfx950.csv; break; done
for t in 1 0; do
  NVL_BLAS_TUNED=$t timeout 600 python bench.py --model qwen3-32b --tp 1 --workload long --max-num-seqs 16 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/cfg5_blas$t.json 2> $OUT/cfg5_blas$t.err; echo "cfg5 blas_tuned=$t rc=$?"; cut -c1-160 $OUT/cfg5_blas$t.json; echo; tail -2 $OUT/cfg5_blas$t.err | grep -v amdgpu
done
