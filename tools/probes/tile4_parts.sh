# tile4 kernel, probe build: NVL_WIDE_DBG bits 1 no x stream, 2 no W stream, 4 no W parks, 8 no fragment reads, 16 no MFMAs,
# 32 x loaders issue every other piece. usage: bash tools/probes/tile4_parts.sh "0 32 2 34"
mkdir -p gpurun_out/r06d
export NVL_PROBES=1 NVL_LIBDIR=$PWD/nano_vllm_amd/lib_probes SWEEP_SHAPES=${SWEEP_SHAPES:-8b_gate_up}
for k in ${1:-3 7 11 15 19 23}; do
  NVL_WIDE_DBG=$k timeout 120 python tools/gemm_wide_streams.py ${M:-256} 2>/dev/null | tail -1
done | tee -a gpurun_out/r06d/tile4_skeleton_parts.txt
