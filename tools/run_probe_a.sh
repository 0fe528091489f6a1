O=gpurun_out/r03p7; mkdir -p $O
timeout 700 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "prefill" > $O/pytest_prefill.log 2>&1; echo "prefill tests rc=$?"; tail -1 $O/pytest_prefill.log
timeout 300 python tools/prefill_bench.py > $O/prefill.json 2> $O/prefill.err; echo "rc=$?"; python -c "
import json;d=json.load(open('$O/prefill.json'));print([(c['name'][:14],c['TFLOPs']) for c in d['cases']], d['relerr'])"
