# The `value is never null` guarantee of a tensor-parallel bench run on the one-GPU stand-in (two ranks share the GPU, gloo): without
# CU masks the two processes' P2P kernels starve each other often enough that some attempts latch a spin timeout — the line must then
# carry the RCCL/gloo re-run's value and the first attempt's status. (a) torchrun --tp 2 (run_tp_external), (b) one process --tp 2
# (run_replica spawning its worker).
set -u
OUT=gpurun_out/${TAG:-r06aj}; mkdir -p $OUT
export NVL_BENCH_SHARE_GPU=1 NVL_BENCH_BACKEND=gloo
for i in 1 2 3; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 2 --tp 2 --steps 1 --warmup 0 --num-seqs 48 --num-kvcache-blocks 300 --no-cpu-baseline > $OUT/torchrun_tp2_$i.json 2> $OUT/torchrun_tp2_$i.err; echo "torchrun tp2 run $i rc=$?"
  python - <<P
import json
ls=[l for l in open('$OUT/torchrun_tp2_$i.json') if l.startswith('{')]
if ls:
    d=json.loads(ls[-1]); print('  value', d.get('value'), '|', d['config'].get('parallelism'), '|', (d.get('tp_p2p_attempt') or {}).get('p2p_status', 'no fallback')[:120])
else:
    print('  NO JSON LINE')
P
done
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --tp 2 --steps 1 --warmup 0 --num-seqs 48 --num-kvcache-blocks 300 --no-cpu-baseline --no-extra-configs > $OUT/single_tp2_$i.json 2> $OUT/single_tp2_$i.err; echo "single-process tp2 run $i rc=$?"
  python - <<P
import json
ls=[l for l in open('$OUT/single_tp2_$i.json') if l.startswith('{')]
if ls:
    d=json.loads(ls[-1]); print('  value', d.get('value'), '|', d['config'].get('parallelism'), '|', (d.get('tp_p2p_attempt') or {}).get('p2p_status', 'no fallback')[:120])
else:
    print('  NO JSON LINE'); import subprocess; print(subprocess.run("grep -v 'Gloo\|socket\|amdgpu' $OUT/single_tp2_$i.err | tail -5", shell=True, capture_output=True, text=True).stdout)
P
done
