O=gpurun_out/r03p8; mkdir -p $O
NVL_BENCH_LIB=$(pwd)/tmp_ab/libnvl_hip_old.so timeout 300 python tools/norm_bench.py > $O/norm_old.json 2> $O/norm_old.err; echo "old rc=$?"; cat $O/norm_old.json
timeout 300 python tools/norm_bench.py > $O/norm_new.json 2> $O/norm_new.err; echo "new rc=$?"; cat $O/norm_new.json
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "norm or splitk or rms or silu" > $O/pytest_norm.log 2>&1; echo "norm tests rc=$?"; tail -1 $O/pytest_norm.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x > $O/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -1 $O/pytest_e2e.log
timeout 900 python -m pytest tests/test_tp_gpu.py -m gpu -q -x -k "p2p_collectives_between_processes or tp2" > $O/pytest_tp.log 2>&1; echo "tp rc=$?"; tail -1 $O/pytest_tp.log
