#!/bin/bash
# round 4, call 4: the whole GPU suite + smoke on the current tree
set -u
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | cut -c1-300 | tail -30
tail -14 $OUT/pytest_gpu.log | head -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3
