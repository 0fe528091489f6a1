set -u
OUT=gpurun_out/${TAG:-r06v}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
timeout 400 python -m pytest tests/test_shared_prefix_gpu.py -q -rf -x > $OUT/pytest_px_kernel.log 2>&1; echo "px kernel rc=$?"; tail -3 $OUT/pytest_px_kernel.log
for x in ${ARMS:-old new}; do
  if [ $x = old ]; then export NVL_LIBDIR=$REPO/nano_vllm_amd/lib_probes_b; else unset NVL_LIBDIR; fi
  (cd /tmp && OMP_NUM_THREADS=8 timeout 400 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_cfg3_$x -o cfg3 -- python $REPO/bench.py --model qwen3-8b --workload prefix --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $REPO/$OUT/cfg3_${x}_under_rocprof.json 2> $REPO/$OUT/cfg3_${x}_prof.err; echo "prof $x rc=$?")
  f=$(find /tmp/prof_cfg3_$x -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/cfg3_${x}_kernel_stats.csv && grep -E "decode_" $OUT/cfg3_${x}_kernel_stats.csv | cut -c1-20,110-190
done
