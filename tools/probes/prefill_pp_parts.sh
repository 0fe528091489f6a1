# Deletion probes of the ping-pong prefill kernel on a PROBE build (NVL_PREFILL_VAR bits: 1 no softmax, 2 no LDS fragment
# reads, 4 no staging, 8 no MFMAs): which part sets the time. usage: bash tools/probes/prefill_pp_parts.sh [tag]
mkdir -p gpurun_out/${1:-r06g}
export NVL_PROBES=1 NVL_LIBDIR=$PWD/nano_vllm_amd/lib_probes
for v in 0 1 2 4 3 5 6 7 8 9 12 13 14 15; do
  NVL_PREFILL_VAR=$v timeout 120 python tools/probes/prefill_time.py 2>/dev/null | tail -1 | sed "s/^/var=$v /"
done | tee gpurun_out/${1:-r06g}/prefill_pp_parts.txt
NVL_PREFILL_PP=0 timeout 120 python tools/probes/prefill_time.py 2>/dev/null | tail -1 | sed "s/^/lockstep /" | tee -a gpurun_out/${1:-r06g}/prefill_pp_parts.txt
