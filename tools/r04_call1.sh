#!/bin/bash
# round 4, call 1: the new T>0 / pool / plan / budget tests, the skinny-GEMM A/B, the full bench line
set -u
OUT=gpurun_out/r04a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf -s -k "sampled or T06 or pow_31 or real_vocabulary or plan_is_refused or budget" --durations=6 > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|T > 0|T>0|T = 0.6|violations" $OUT/pytest_new.log | cut -c1-400 | tail -30
for g in 1 2; do
  NVL_SKINNY_SILU_MGROUPS=$g NVL_BENCH_MS=32,64,96,131,144,208,256 timeout 300 python tools/gemm_bench.py > $OUT/gemm_silu_groups$g.json 2> $OUT/gemm_silu_groups$g.err; echo "gemm groups=$g rc=$?"
  python -c "
import json; d=json.load(open('$OUT/gemm_silu_groups$g.json')); print(d['relerr_max']); [print(' ',k,v) for k,v in d['time_us'].items() if 'gate_up' in k or 'm131' in k]"
done
T0=$(date +%s); timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -c 600 $OUT/bench.err | grep -v "headline line"
python - <<'P'
import json
d=json.load(open('gpurun_out/r04a/bench.json'))
print(round(d['value']), d['roofline']['frac'], d['roofline'].get('decode_step_frac_of_8TBps'), d['config'].get('packed_weight_bytes'))
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:600])
print('parity', json.dumps(d.get('parity'))[:900])
for k,v in d.get('extra_configs',{}).items(): print(k, v.get('value'), v.get('error'), v.get('wall_s_incl_engine_start'), (v.get('roofline') or {}).get('frac'), (v.get('roofline_prefill') or {}).get('achieved'), (v.get('stderr_tail') or '')[-300:])
P
