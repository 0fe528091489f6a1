#!/bin/bash
set -u
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/blas_prefill_probe.py 16000 > $OUT/blas_prefill_probe.json 2> $OUT/blas_prefill_probe.err; echo "probe rc=$?"; grep -v amdgpu.ids $OUT/blas_prefill_probe.err | tail -10
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "sampler or lmhead or persistent" 2>&1 | tail -2
