O=gpurun_out/r03p9; mkdir -p $O
for a in 2 0 2 0; do NVL_DECODE_AHEAD=$a NVL_BENCH_MS=16,64,131,144,208,256 timeout 400 python tools/gemm_bench.py 0.6b > $O/gemm_ahead$a.json 2> $O/gemm_ahead$a.err; echo "ahead=$a rc=$?"; python -c "
import json;d=json.load(open('$O/gemm_ahead$a.json'));print({k:v[0] for k,v in d['time_us'].items()}, d['relerr_max'])"; done
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "linear_decode or splitk or packed" > $O/pytest_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -1 $O/pytest_gemm.log
