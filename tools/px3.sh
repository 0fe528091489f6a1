# The shared-prefix attention on BASELINE config 3: kernel + engine parity, then bench.py --model qwen3-8b --workload prefix per arm
# (a number = NVL_PX_ITEMS_PER_WG: items per pack workgroup of the one-launch form; the record in
# profiles/r06_shared_prefix_one_launch.json also has a `sep` arm — NVL_PX_SEPARATE=1, the pass as its own launch — which existed
# at commit 6996907: the switch was dropped with the scalar-register work that followed)
set -u
OUT=gpurun_out/${TAG:-r06w}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
if [ "${TESTS:-1}" = 1 ]; then
timeout 400 python -m pytest tests/test_shared_prefix_gpu.py -q -rf -x > $OUT/pytest_px_kernel.log 2>&1; echo "px kernel rc=$?"; tail -3 $OUT/pytest_px_kernel.log | cut -c1-300
timeout 500 python -m pytest tests/test_e2e_gpu.py -q -rf -k "shared_system_prompt or two_shared or block_edges or g5" > $OUT/pytest_px_e2e.log 2>&1; echo "px e2e rc=$?"; tail -3 $OUT/pytest_px_e2e.log | cut -c1-300
fi
n=0
for x in ${ARMS:-3 4 3 4}; do
  n=$((n+1))
  export NVL_PX_ITEMS_PER_WG=$x
  OMP_NUM_THREADS=8 timeout 300 python bench.py --model qwen3-8b --workload prefix --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/cfg3_${x}_$n.json 2> $OUT/cfg3_${x}_$n.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/cfg3_${x}_$n.json') if l.startswith('{')][-1])
r=d['roofline']; print('arm $x:', round(d['value']), 'tok/s; attn', round(r['avg_launch_us'],1), 'us', 'step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'])
" || tail -5 $OUT/cfg3_${x}_$n.err; done
