#!/bin/bash
set -u
OUT=gpurun_out/r04i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf -k "linear_wide or full_width" > $OUT/pytest_wide.log 2>&1; echo "wide tests rc=$?"; tail -2 $OUT/pytest_wide.log
SWEEP_SHAPES=8b_qkv,8b_o,8b_gate_up,8b_down,32b_qkv,32b_o,32b_down timeout 600 python tools/gemm_wide_m256.py 208 256 > $OUT/gemm_wide_m256_tuned.json 2> $OUT/gemm_wide_m256_tuned.err; echo "m256 rc=$?"; grep -v amdgpu.ids $OUT/gemm_wide_m256_tuned.err | tail -16
timeout 600 python bench.py --model qwen3-8b --workload prefix --no-cpu-baseline --no-roofline --warmup 0 > $OUT/cfg3.json 2> $OUT/cfg3.err; echo "cfg3 rc=$?"; cut -c1-200 $OUT/cfg3.json; echo
