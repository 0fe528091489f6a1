# Where the shared-prefix pass starts to pay (the engine's NVL_SHARED_PREFIX_MIN_MB threshold): BASELINE config 3's workload at
# 48 / 96 / 160 / 256 sequences on the 0.6B and 8B shapes, pass forced on (NVL_SHARED_PREFIX_MIN_MB=0) vs off, one box.
set -u
OUT=gpurun_out/${TAG:-r06ae}; mkdir -p $OUT
for m in ${MODELS:-qwen3-0.6b qwen3-8b}; do for n in ${SEQS:-48 96 160 256}; do for mode in off on; do
  unset NVL_SHARED_PREFIX NVL_SHARED_PREFIX_MIN_MB
  if [ $mode = off ]; then export NVL_SHARED_PREFIX=0; else export NVL_SHARED_PREFIX_MIN_MB=0; fi
  OMP_NUM_THREADS=8 timeout 200 python bench.py --model $m --workload prefix --num-seqs $n --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/cx_${m}_${n}_$mode.json 2> $OUT/cx_${m}_${n}_$mode.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/cx_${m}_${n}_$mode.json') if l.startswith('{')][-1])
r=d['roofline']; print('$m $n $mode:', round(d['value']), 'tok/s; attn', round(r['avg_launch_us'],1), 'us; step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'], 'px steps', d['config']['decode_step_fusions']['decode_steps_with_shared_prefix_pass'])
" || tail -3 $OUT/cx_${m}_${n}_$mode.err
done; done; done
# the opt-in fp8 KV cache (pass off by default there): forced on vs off, 256 sequences
if [ "${FP8:-1}" = 1 ]; then for m in qwen3-0.6b qwen3-8b; do for mode in off on; do
  unset NVL_SHARED_PREFIX NVL_SHARED_PREFIX_MIN_MB
  if [ $mode = off ]; then export NVL_SHARED_PREFIX=0; else export NVL_SHARED_PREFIX=1 NVL_SHARED_PREFIX_MIN_MB=0; fi
  OMP_NUM_THREADS=8 timeout 200 python bench.py --model $m --workload prefix --kv-cache-dtype fp8 --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/cx_fp8_${m}_$mode.json 2> $OUT/cx_fp8_${m}_$mode.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/cx_fp8_${m}_$mode.json') if l.startswith('{')][-1])
r=d['roofline']; print('fp8 KV $m 256 $mode:', round(d['value']), 'tok/s; attn', round(r['avg_launch_us'],1), 'us; step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'], 'px steps', d['config']['decode_step_fusions']['decode_steps_with_shared_prefix_pass'])
" || tail -3 $OUT/cx_fp8_${m}_$mode.err
done; done; fi
