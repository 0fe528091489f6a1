#!/bin/bash
# round 4: final validation of the tree — whole GPU suite, smoke, the bench line, rocprofv3 kernel stats of the bench
set -u
OUT=gpurun_out/r04final2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | cut -c1-300 | tail -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -2
bash tools/gpu_round.sh r04final2 benchfull benchprof
