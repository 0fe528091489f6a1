# items per pack workgroup (NVL_PX_ITEMS_PER_WG) against the batch size: config 3's workload at 96 / 160 / 256 sequences, pass forced on
# with n = 1 / 2 / 3 and off, 0.6B and 8B shapes, one box
set -u
OUT=gpurun_out/${TAG:-r06am}; mkdir -p $OUT
for m in qwen3-8b qwen3-0.6b; do for b in 96 160 256; do for n in off 1 2 3; do
  unset NVL_SHARED_PREFIX NVL_SHARED_PREFIX_MIN_MB NVL_PX_ITEMS_PER_WG
  if [ $n = off ]; then export NVL_SHARED_PREFIX=0; else export NVL_SHARED_PREFIX_MIN_MB=0 NVL_PX_ITEMS_PER_WG=$n; fi
  OMP_NUM_THREADS=8 timeout 200 python bench.py --model $m --workload prefix --num-seqs $b --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/s_${m}_${b}_$n.json 2>/dev/null
  python -c "
import json
d=json.loads([l for l in open('$OUT/s_${m}_${b}_$n.json') if l.startswith('{')][-1]); print('$m B=$b n=$n:', round(d['value']), 'tok/s; attn', round(d['roofline']['avg_launch_us'],1), 'us; step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'])"
done; done; done
