set -u
OUT=gpurun_out/r06ao; mkdir -p $OUT
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_tp_gpu.py -q -rf -k "shared or prefix or config3 or block_edges or g5 or 14b" > $OUT/pytest_engine_px.log 2>&1; echo "engine px tests rc=$?"; tail -3 $OUT/pytest_engine_px.log | cut -c1-200
for cfg in "qwen3-8b 96" "qwen3-8b 128" "qwen3-8b 160" "qwen3-8b 256" "qwen3-32b-tp8rank 160" "qwen3-32b-tp8rank 256"; do set -- $cfg; for mode in off on; do
  unset NVL_SHARED_PREFIX NVL_SHARED_PREFIX_MIN_MB
  if [ $mode = off ]; then export NVL_SHARED_PREFIX=0; else export NVL_SHARED_PREFIX_MIN_MB=0; fi
  OMP_NUM_THREADS=8 timeout 200 python bench.py --model $1 --workload prefix --num-seqs $2 --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/e_$1_$2_$mode.json 2>$OUT/e_$1_$2_$mode.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/e_$1_$2_$mode.json') if l.startswith('{')][-1]); print('$1 B=$2 $mode:', round(d['value']), 'tok/s; step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'], 'px steps', d['config']['decode_step_fusions']['decode_steps_with_shared_prefix_pass'])" || tail -3 $OUT/e_$1_$2_$mode.err
done; done
