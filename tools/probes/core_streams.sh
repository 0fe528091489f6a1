mkdir -p gpurun_out/r06c
export NVL_PROBES=1 NVL_LIBDIR=$PWD/nano_vllm_amd/lib_probes SWEEP_SHAPES=8b_gate_up
for core in 1 0; do for k in 0 1 2 3 4 7 8 11; do
  if [ $core = 0 ] && [ $k -ge 4 ]; then continue; fi
  NVL_WIDE_CORE=$core NVL_WIDE_DBG=$k timeout 120 python tools/gemm_wide_streams.py 256 2>/dev/null | tail -1 | sed "s/^/core=$core /"
done; done | tee gpurun_out/r06c/streams_8b_gate_up_m256.txt
