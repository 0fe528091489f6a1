O=gpurun_out/r03p2; mkdir -p $O
timeout 700 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "prefill" > $O/pytest_prefill.log 2>&1; echo "prefill tests (defaults) rc=$?"; tail -2 $O/pytest_prefill.log
for cfg in "0 7" "1 7" "1 6" "1 4" "1 0"; do set -- $cfg; NVL_PREFILL_SCHED=$1 NVL_PREFILL_LEGACY=$2 timeout 300 python tools/prefill_bench.py > $O/prefill_s$1_l$2.json 2> $O/prefill_s$1_l$2.err; echo "sched=$1 legacy=$2 rc=$?"; python -c "
import json;d=json.load(open('$O/prefill_s$1_l$2.json'));print([(c['name'][:14],c['TFLOPs']) for c in d['cases']], d['relerr'])"; done
