OUT=gpurun_out/r06aq; mkdir -p $OUT
for cfg in "qwen3-8b 80" "qwen3-8b 88" "qwen3-0.6b 72" "qwen3-0.6b 80" "qwen3-0.6b 64"; do set -- $cfg; for mode in off default; do
  unset NVL_SHARED_PREFIX NVL_SHARED_PREFIX_MIN_MB
  [ $mode = off ] && export NVL_SHARED_PREFIX=0
  OMP_NUM_THREADS=8 timeout 200 python bench.py --model $1 --workload prefix --num-seqs $2 --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/t_$1_$2_$mode.json 2>/dev/null
  python -c "
import json
d=json.loads([l for l in open('$OUT/t_$1_$2_$mode.json') if l.startswith('{')][-1]); print('$1 B=$2 $mode:', round(d['value']), 'tok/s; step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'], 'px steps', d['config']['decode_step_fusions']['decode_steps_with_shared_prefix_pass'])"
done; done
