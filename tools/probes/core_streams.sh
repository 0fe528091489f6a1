# Stream isolation of the deep-K decode GEMM kernels on a PROBE build (see tools/gemm_wide_streams.py): which of the x
# stream / the weight stream / the consumer loop sets the time. usage: bash tools/probes/core_streams.sh [tag]
mkdir -p gpurun_out/${1:-r06c}
export NVL_PROBES=1 NVL_LIBDIR=$PWD/nano_vllm_amd/lib_probes SWEEP_SHAPES=${SWEEP_SHAPES:-8b_gate_up}
for arm in tile4 core hipcc; do for k in 0 1 2 3 4 7 8 11; do
  if [ $arm != core ] && [ $k -ge 4 ]; then continue; fi
  case $arm in tile4) T4=1; CORE=1;; core) T4=0; CORE=1;; hipcc) T4=0; CORE=0;; esac
  NVL_WIDE_TILE4=$T4 NVL_WIDE_CORE=$CORE NVL_WIDE_DBG=$k timeout 120 python tools/gemm_wide_streams.py ${M:-256} 2>/dev/null | tail -1 | sed "s/^/$arm /"
done; done | tee gpurun_out/${1:-r06c}/streams_${SWEEP_SHAPES}_m${M:-256}.txt
