#!/bin/bash
# One gpurun call: GPU tests, bench line, rocprofv3 kernel stats + PMC traffic of the attention replay.
# Usage (from repo root on the GPU box): bash tools/gpu_round.sh <tag> [tests] [bench] [prof] [pmc] [probe]
set -u
TAG=${1:-r01}; shift || true
WHAT="${*:-tests bench prof pmc}"
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -rf --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -40;;
tptests)
  timeout 900 python -m pytest tests/test_tp_gpu.py -m gpu -q -rf -s > $OUT/pytest_tp.log 2>&1; echo "tp pytest rc=$?" | tee -a $OUT/pytest_tp.log; tail -40 $OUT/pytest_tp.log;;
bench)
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json;;
prof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_replay -o replay -- python $REPO/tools/attn_replay.py --fused > $OUT/replay_under_rocprof.json 2> $OUT/replay_prof.err; echo "prof rc=$?")
  f=$(find /tmp/prof_replay -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/replay_kernel_stats.csv && head -5 $OUT/replay_kernel_stats.csv;;
pmc)
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'decode_' -f csv -d /tmp/pmc_replay -o replay -- python $REPO/tools/attn_replay.py --fused --reps 1 > $OUT/replay_under_pmc.json 2> $OUT/replay_pmc.err; echo "pmc rc=$?")
  python tools/pmc_summary.py /tmp/pmc_replay $OUT/pmc_fetch_summary.json | tail -30;;
benchprof)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_bench -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.json 2> $OUT/bench_prof.err; echo "benchprof rc=$?")
  f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_kernel_stats.csv && head -12 $OUT/bench_kernel_stats.csv | cut -c1-200;;
probe)
  timeout 600 python tools/gpu_probe.py > $OUT/probe.log 2>&1; cp gpurun_out/probe.json $OUT/ 2>/dev/null; tail -3 $OUT/probe.log;;
gemm)
  timeout 600 python tools/gemm_bench.py > $OUT/gemm_bench.json 2> $OUT/gemm_bench.err; echo "gemm rc=$?"; tail -c 1500 $OUT/gemm_bench.err; cat $OUT/gemm_bench.json;;
retest)
  timeout 1200 python -m pytest tests/test_tp_gpu.py tests/test_kernels_gpu.py -m gpu -q -rf -k "tp2 or long or continuation or shards or linear_decode or p2p" > $OUT/pytest_retest.log 2>&1; echo "retest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_retest.log | tail -20;;
prefillx)
  for x in 0 1; do NVL_PREFILL_XCD=$x timeout 600 python tools/prefill_bench.py > $OUT/prefill_xcd$x.json 2> $OUT/prefill_xcd$x.err; echo "prefill xcd=$x rc=$?"; cat $OUT/prefill_xcd$x.json; echo; done;;
newtests)
  timeout 1200 python -m pytest tests -m gpu -q -rf -k "lmhead or prefill or logits or tiny_model_greedy or tp2 or sampler or 06b_shape_greedy" > $OUT/pytest_new.log 2>&1; echo "newtests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_new.log | tail -20;;
fp8)
  timeout 900 python -m pytest tests -m gpu -q -rf -k "fp8" > $OUT/pytest_fp8.log 2>&1; echo "fp8 tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|fp8 KV vs" $OUT/pytest_fp8.log | tail -12
  timeout 600 python bench.py --kv-cache-dtype fp8 --no-cpu-baseline > $OUT/bench_fp8kv.json 2> $OUT/bench_fp8kv.err; echo "bench fp8 rc=$?"; tail -c 400 $OUT/bench_fp8kv.err; cut -c1-1500 $OUT/bench_fp8kv.json;;
px)
  # shared-prefix attention pass: kernel parity, engine parity, config 3 A/B (pass off / on, alternating), headline sanity.
  # "old" = another commit's tree built in a worktree next to this one (git worktree add -f _ab_old <commit>; build() there):
  # skipped when it is not there
  timeout 400 python -m pytest tests/test_shared_prefix_gpu.py -q -rf > $OUT/pytest_px_kernel.log 2>&1; echo "px kernel rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_px_kernel.log | tail -30
  timeout 500 python -m pytest tests/test_e2e_gpu.py -q -rf -s -k "shared_system_prompt or block_edges" > $OUT/pytest_px_e2e.log 2>&1; echo "px e2e rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|shared system|block /" $OUT/pytest_px_e2e.log | cut -c1-400 | tail -20
  for x in old 0 1 old 0 1; do
    [ $x = old ] && [ ! -d _ab_old ] && continue
    if [ $x = old ]; then (cd _ab_old && timeout 300 python bench.py --model qwen3-8b --workload prefix --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/cfg3_px$x.json 2> $OUT/cfg3_px$x.err)
    else NVL_SHARED_PREFIX=$x timeout 300 python bench.py --model qwen3-8b --workload prefix --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/cfg3_px$x.json 2> $OUT/cfg3_px$x.err; fi
    echo "cfg3 px=$x rc=$?"; tail -c 300 $OUT/cfg3_px$x.err | grep -v amdgpu.ids; python -c "
import json,sys
d=json.loads([l for l in open('$OUT/cfg3_px$x.json') if l.startswith('{')][-1])
r=d['roofline']; print('px=$x', round(d['value']), 'tok/s', round(d['ms_per_step'],1), 'ms; attn', round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), 'step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'], 'px steps', d['config']['decode_step_fusions'].get('decode_steps_with_shared_prefix_pass'))
"; cp $OUT/cfg3_px$x.json $OUT/cfg3_px${x}_run_$(date +%s).json; done
  for x in old new old new; do
    [ $x = old ] && [ ! -d _ab_old ] && continue
    if [ $x = old ]; then (cd _ab_old && timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 2 --warmup 1 > $OUT/headline_$x.json 2> $OUT/headline_$x.err)
    else timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 2 --warmup 1 > $OUT/headline_$x.json 2> $OUT/headline_$x.err; fi
    echo "headline $x rc=$?"; python -c "
import json
d=json.loads([l for l in open('$OUT/headline_$x.json') if l.startswith('{')][-1])
print('$x', round(d['value']), 'tok/s attn', round(d['roofline']['avg_launch_us'],2), d['config']['decode_ms_per_step_by_batch']['ms_per_step'])
"; cp $OUT/headline_$x.json $OUT/headline_${x}_run_$(date +%s).json; done;;
px2)
  # after the prefix kernel's load reorder: kernel parity again, config 3 with the pass (child-style env), and the headline
  # with / without a small OpenMP pool (the host loop's torch ops)
  timeout 400 python -m pytest tests/test_shared_prefix_gpu.py -q -rf > $OUT/pytest_px_kernel.log 2>&1; echo "px kernel rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_px_kernel.log | tail -30
  for x in 1 0 1; do NVL_SHARED_PREFIX=$x OMP_NUM_THREADS=8 timeout 300 python bench.py --model qwen3-8b --workload prefix --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/cfg3_px$x.json 2> $OUT/cfg3_px$x.err; python -c "
import json
d=json.loads([l for l in open('$OUT/cfg3_px$x.json') if l.startswith('{')][-1])
r=d['roofline']; print('px=$x omp8', round(d['value']), 'tok/s; attn', round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), 'step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'])
"; cp $OUT/cfg3_px$x.json $OUT/cfg3_px${x}_run_$(date +%s).json; done
  for x in unset 8 unset 8; do
    if [ $x = unset ]; then timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 2 --warmup 1 > $OUT/headline_omp$x.json 2> $OUT/headline_omp$x.err
    else OMP_NUM_THREADS=$x timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 2 --warmup 1 > $OUT/headline_omp$x.json 2> $OUT/headline_omp$x.err; fi
    python -c "
import json
d=json.loads([l for l in open('$OUT/headline_omp$x.json') if l.startswith('{')][-1])
print('omp=$x', round(d['value']), 'tok/s attn', round(d['roofline']['avg_launch_us'],2), d['config']['decode_ms_per_step_by_batch']['ms_per_step'])
"; cp $OUT/headline_omp$x.json $OUT/headline_omp${x}_run_$(date +%s).json; done;;
pxprof)
  # rocprofv3 kernel stats of config 3 without / with the shared-prefix pass (same box)
  for x in 0 1; do (cd /tmp && NVL_SHARED_PREFIX=$x OMP_NUM_THREADS=8 timeout 400 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_cfg3_px$x -o cfg3 -- python $REPO/bench.py --model qwen3-8b --workload prefix --warmup 1 --steps 1 --no-cpu-baseline --no-extra-configs > $OUT/cfg3_px${x}_under_rocprof.json 2> $OUT/cfg3_px${x}_prof.err; echo "pxprof $x rc=$?")
    f=$(find /tmp/prof_cfg3_px$x -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/cfg3_px${x}_kernel_stats.csv && head -14 $OUT/cfg3_px${x}_kernel_stats.csv | cut -c1-160
    python -c "
import json
d=json.loads([l for l in open('$OUT/cfg3_px${x}_under_rocprof.json') if l.startswith('{')][-1])
print('px=$x (under rocprof, OMP 8)', round(d['value']), 'tok/s step', d['config']['decode_ms_per_step_by_batch']['ms_per_step'])
"; done;;
cfg3)
  timeout 900 python bench.py --model qwen3-8b --workload prefix --no-cpu-baseline --warmup 0 > $OUT/bench_cfg3_8b_prefix.json 2> $OUT/bench_cfg3.err; echo "cfg3 rc=$?"; tail -c 300 $OUT/bench_cfg3.err; cut -c1-1200 $OUT/bench_cfg3_8b_prefix.json;;
tpfunc)
  timeout 600 python tools/tp_functional.py qwen3-32b 2 8 48 > $OUT/tp_functional_32b_tp2.json 2> $OUT/tp_functional.err; echo "tpfunc rc=$?"; tail -c 800 $OUT/tp_functional.err; cat $OUT/tp_functional_32b_tp2.json;;
pmcprefill)
  (cd /tmp && rocprofv3 -L > $OUT/rocprof_counters.txt 2>&1; grep -c . $OUT/rocprof_counters.txt; grep -o -E "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_INSTS_VALU\b|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_BUSY_CU_CYCLES" $OUT/rocprof_counters.txt | sort -u | tr '\n' ' '; echo
   timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-include-regex prefill_attn -f csv -d /tmp/pmc_prefill -o pf -- python $REPO/tools/prefill_bench.py > $OUT/prefill_under_pmc.json 2> $OUT/prefill_pmc.err; echo "pmcprefill rc=$?"; tail -c 500 $OUT/prefill_pmc.err)
  f=$(find /tmp/pmc_prefill -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_prefill_summary.py $f $OUT/prefill_pmc_summary.json && cat $OUT/prefill_pmc_summary.json;;
p2pbench)
  for w in 2 4; do timeout 300 python tools/p2p_bench.py $w > $OUT/p2p_bench_w$w.json 2> $OUT/p2p_bench_w$w.err; echo "p2pbench w=$w rc=$?"; tail -c 300 $OUT/p2p_bench_w$w.err; cat $OUT/p2p_bench_w$w.json; done;;
tprun)
  # what the driver runs for N = 2 (python -m torch.distributed.run ... bench.py --gpus 2), with both ranks on the ONE GPU
  # (functional: NVL_BENCH_SHARE_GPU=1 + gloo). (a) default model: 2 data-parallel replicas + the Qwen3-32B TP=2 extra;
  # (b) --tp 2 as the primary metric on Qwen3-0.6B shapes.
  export NVL_BENCH_SHARE_GPU=1 NVL_BENCH_BACKEND=gloo
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --num-seqs 48 --gpu-memory-utilization 0.3 --num-kvcache-blocks 300 --no-cpu-baseline --no-roofline > $OUT/torchrun_dp2_plus_tp_extra.json 2> $OUT/torchrun_dp2.err; echo "torchrun dp2+extra rc=$?"; grep -v "socket.cpp\|Gloo\|amdgpu.ids" $OUT/torchrun_dp2.err | tail -15; cat $OUT/torchrun_dp2_plus_tp_extra.json
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --tp 2 --steps 1 --warmup 0 --num-seqs 48 --num-kvcache-blocks 300 --no-cpu-baseline > $OUT/torchrun_tp2_qwen3-0.6b.json 2> $OUT/torchrun_tp2.err; echo "torchrun tp2 rc=$?"; grep -v "socket.cpp\|Gloo\|amdgpu.ids" $OUT/torchrun_tp2.err | tail -15; cat $OUT/torchrun_tp2_qwen3-0.6b.json
  unset NVL_BENCH_SHARE_GPU NVL_BENCH_BACKEND;;
tprun2)
  export NVL_BENCH_SHARE_GPU=1 NVL_BENCH_BACKEND=gloo
  for v in "lean" "fenced"; do
    export NVL_TP_P2P_HANDOFF=$v
    T0=$(date +%s); timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --tp 2 --steps 1 --warmup 0 --num-seqs 48 --num-kvcache-blocks 300 --no-cpu-baseline > $OUT/torchrun_tp2_$v.json 2> $OUT/torchrun_tp2_$v.err; echo "torchrun tp2 $v rc=$? wall=$(( $(date +%s) - T0 )) s"; grep -v "socket.cpp\|Gloo\|amdgpu.ids\|OMP_NUM\|\*\*\*" $OUT/torchrun_tp2_$v.err | tail -6; cut -c1-700 $OUT/torchrun_tp2_$v.json; echo
  done
  unset NVL_BENCH_SHARE_GPU NVL_BENCH_BACKEND NVL_TP_P2P_HANDOFF;;
prefillw)
  for w in 4 8; do NVL_PREFILL_WAVES=$w timeout 600 python tools/prefill_bench.py > $OUT/prefill_waves$w.json 2> $OUT/prefill_waves$w.err; echo "prefill waves=$w rc=$?"; cat $OUT/prefill_waves$w.json; echo; done
  NVL_PREFILL_WAVES=8 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "prefill" 2>&1 | tail -3;;
replay)
  timeout 600 python tools/attn_replay.py --fused > $OUT/replay.json 2> $OUT/replay.err; cat $OUT/replay.json;;
fp8g8)
  timeout 900 python -m pytest tests -m gpu -q -rf -k "fp8" > $OUT/pytest_fp8.log 2>&1; echo "fp8 tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|fp8 KV vs" $OUT/pytest_fp8.log | tail -12
  for f in "" "--fp8"; do
    timeout 600 python tools/attn_replay.py --fused --hq 8 --hkv 1 --layers 64 --every 16 $f > $OUT/replay_g8$f.json 2> $OUT/replay_g8$f.err; cat $OUT/replay_g8$f.json
    timeout 600 python tools/attn_replay.py --fused --hq 16 --hkv 2 --layers 64 --every 16 $f > $OUT/replay_g8x2$f.json 2> $OUT/replay_g8x2$f.err; cat $OUT/replay_g8x2$f.json
  done;;
wide)
  timeout 900 python tools/gemm_wide_bench.py ${WIDE_MODELS:-8b 32b_tp8} > $OUT/gemm_wide${WIDE_TAG:-}.json 2> $OUT/gemm_wide${WIDE_TAG:-}.err; echo "wide rc=$?"; tail -c 800 $OUT/gemm_wide${WIDE_TAG:-}.err; cat $OUT/gemm_wide${WIDE_TAG:-}.json;;
widex)
  for cfg in "1 0" "2 2" "1 2" "1 4" "2 4" "2 8"; do
    set -- $cfg
    NVL_WIDE_NT=$1 NVL_WIDE_SPLIT=$2 BENCH_M=16,144 timeout 600 python tools/gemm_wide_bench.py ${WIDE_MODELS:-8b} > $OUT/gemm_wide_nt$1_s$2.json 2> $OUT/gemm_wide_nt$1_s$2.err; echo "nt=$1 split=$2 rc=$?"
    python -c "import json,sys; d=json.load(open('$OUT/gemm_wide_nt$1_s$2.json')); print(d['relerr_max']); [print(' ',k,v) for k,v in d['time_us'].items()]"
  done;;
widetests)
  timeout 900 python -m pytest tests -m gpu -q -rf -x -k "linear_wide or full_width" -s > $OUT/pytest_wide.log 2>&1; echo "wide tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|exact argmax" $OUT/pytest_wide.log | tail -12;;
bench8b)
  for wmode in ${BENCH8B_MODES:-0 auto}; do
    NVL_GEMM_WIDE=$wmode timeout 900 python bench.py --model qwen3-8b --no-cpu-baseline --no-roofline > $OUT/bench_8b_wide_$wmode.json 2> $OUT/bench_8b_wide_$wmode.err; echo "bench 8b wide=$wmode rc=$?"; tail -c 300 $OUT/bench_8b_wide_$wmode.err; cut -c1-400 $OUT/bench_8b_wide_$wmode.json; echo
  done;;
wideall)
  WIDE_MODELS="8b 32b 32b_tp4 32b_tp8 lm_head" BENCH_M=16,64,144,256 timeout 900 python tools/gemm_wide_bench.py 8b 32b 32b_tp4 32b_tp8 lm_head > $OUT/gemm_wide_all.json 2> $OUT/gemm_wide_all.err; echo "wideall rc=$?"; tail -c 300 $OUT/gemm_wide_all.err
  (cd /tmp && BENCH_M=144 timeout 600 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_wide -o wide -- python $REPO/tools/gemm_wide_bench.py 8b 32b_tp8 > $OUT/gemm_wide_under_rocprof.json 2> $OUT/gemm_wide_prof.err; echo "wide prof rc=$?")
  f=$(find /tmp/prof_wide -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/gemm_wide_kernel_stats.csv && head -8 $OUT/gemm_wide_kernel_stats.csv | cut -c1-160;;
mfmag2)
  NVL_DECODE_MFMA=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "decode and not fp8 and not linear" > $OUT/pytest_mfmag2.log 2>&1; echo "mfma small-G tests rc=$?"; tail -3 $OUT/pytest_mfmag2.log
  for v in 0 1; do
    NVL_DECODE_MFMA=$v timeout 600 python tools/attn_replay.py --fused > $OUT/replay_mfma$v.json 2> $OUT/replay_mfma$v.err; cat $OUT/replay_mfma$v.json
    NVL_DECODE_MFMA=$v timeout 600 python tools/attn_replay.py --fused --hq 32 --hkv 8 --layers 36 > $OUT/replay_g4_mfma$v.json 2> $OUT/replay_g4_mfma$v.err; cat $OUT/replay_g4_mfma$v.json
  done;;
mfmafp8)
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "decode or fp8" > $OUT/pytest_mfmafp8.log 2>&1; echo "decode+fp8 tests rc=$?"; tail -3 $OUT/pytest_mfmafp8.log
  for v in 0 1; do
    NVL_DECODE_MFMA=$v timeout 600 python tools/attn_replay.py --fused --fp8 > $OUT/replay_fp8_mfma$v.json 2> $OUT/replay_fp8_mfma$v.err; cat $OUT/replay_fp8_mfma$v.json
  done;;
tprun1)
  # only the data-parallel line + the Qwen3-32B TP extra (a separate child job per rank), both ranks on the ONE GPU
  export NVL_BENCH_SHARE_GPU=1 NVL_BENCH_BACKEND=gloo NVL_BENCH_TP_EXTRA_TIMEOUT=400
  timeout 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --num-seqs 48 --gpu-memory-utilization 0.3 --num-kvcache-blocks 300 --no-cpu-baseline --no-roofline > $OUT/torchrun_dp2_plus_tp_extra.json 2> $OUT/torchrun_dp2.err; echo "torchrun dp2+extra rc=$?"; grep -v "socket.cpp\|Gloo\|amdgpu.ids\|OMP_NUM\|\*\*\*" $OUT/torchrun_dp2.err | tail -8; cut -c1-2500 $OUT/torchrun_dp2_plus_tp_extra.json
  unset NVL_BENCH_SHARE_GPU NVL_BENCH_BACKEND NVL_BENCH_TP_EXTRA_TIMEOUT;;
pmcg)
  # HBM traffic (FETCH_SIZE) of the decode-attention kernel at every group size the BASELINE configs run:
  # G = 2 (Qwen3-0.6B), G = 4 (Qwen3-8B: config 3), G = 8 (Qwen3-32B: configs 4 / 5); bench schedule, one rep
  for cfg in "16 8 28 8 qwen3-0.6b 2" "32 8 36 8 qwen3-8b 4" "64 8 64 16 qwen3-32b 8"; do
    set -- $cfg
    rm -rf /tmp/pmc_g$6
    (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'decode_' -f csv -d /tmp/pmc_g$6 -o replay -- python $REPO/tools/attn_replay.py --fused --reps 1 --hq $1 --hkv $2 --layers $3 --every $4 > $OUT/replay_under_pmc_g$6.json 2> $OUT/replay_pmc_g$6.err; echo "pmc G=$6 rc=$?")
    python tools/pmc_summary.py /tmp/pmc_g$6 $OUT/pmc_fetch_summary_g$6.json > /dev/null
    python tools/pmc_traffic_update.py $OUT/pmc_fetch_summary_g$6.json $OUT/replay_under_pmc_g$6.json $5 "decode_mfma8_kernel<fused, bf16 KV, G=$6>"
  done
  cp profiles/pmc_traffic.json $OUT/pmc_traffic.json;;
pmcpx)
  # HBM traffic of BASELINE config 3's own schedule (shared system prompt), G = 4: with the shared-prefix pass, and
  # the same schedule without it
  cp gpurun_out/r05r/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
  rm -rf /tmp/pmc_g4px /tmp/pmc_g4nopx
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'decode_' -f csv -d /tmp/pmc_g4px -o replay -- python $REPO/tools/attn_replay.py --fused --reps 1 --hq 32 --hkv 8 --layers 36 --every 8 --workload prefix --pool-blocks 6432 > $OUT/replay_under_pmc_g4_prefix.json 2> $OUT/replay_pmc_g4_prefix.err; echo "pmc G=4 prefix rc=$?")
  python tools/pmc_summary.py /tmp/pmc_g4px $OUT/pmc_fetch_summary_g4_prefix.json > /dev/null
  python tools/pmc_traffic_update.py $OUT/pmc_fetch_summary_g4_prefix.json $OUT/replay_under_pmc_g4_prefix.json qwen3-8b "decode_mfma8_shared_kernel<fused, bf16 KV, G=4>"
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'decode_' -f csv -d /tmp/pmc_g4nopx -o replay -- python $REPO/tools/attn_replay.py --fused --reps 1 --hq 32 --hkv 8 --layers 36 --every 8 --workload prefix --pool-blocks 6432 --no-shared-prefix > $OUT/replay_under_pmc_g4_prefix_pass_off.json 2> $OUT/replay_pmc_g4_prefix_pass_off.err; echo "pmc G=4 prefix, pass off rc=$?")
  python tools/pmc_summary.py /tmp/pmc_g4nopx $OUT/pmc_fetch_summary_g4_prefix_pass_off.json | tail -8
  cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
  # and the kernel durations of the two forms of the same replay
  for v in "" "--no-shared-prefix"; do (cd /tmp && rm -rf /tmp/prof_px && timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_px -o replay -- python $REPO/tools/attn_replay.py --fused --hq 32 --hkv 8 --layers 36 --every 8 --workload prefix --pool-blocks 6432 $v > $OUT/replay_prefix_under_rocprof$v.json 2>/dev/null; f=$(find /tmp/prof_px -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/replay_prefix_kernel_stats$v.csv && grep -i "decode_" $OUT/replay_prefix_kernel_stats$v.csv | cut -c1-140); done;;
tp3)
  timeout 1500 python -m pytest tests -m gpu -q -rf -s -k "tp or cpu_oracle or fp8_kv_store or rccl or p2p" --durations=8 > $OUT/pytest_tp3.log 2>&1; echo "tp3 pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|argmax|device-resident" $OUT/pytest_tp3.log | tail -30;;
benchfull)
  T0=$(date +%s); timeout 1500 python bench.py ${BENCH_ARGS:-} > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -c 400 $OUT/bench_full.err; python -c "
import json; d=json.load(open('$OUT/bench_full.json')); print(round(d['value']), d['roofline']['frac'], d['roofline']['decode_step']['frac_of_8TBps'], d.get('cpu_baseline')); [print(k, v.get('value'), v.get('error'), v.get('wall_s_incl_engine_start'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'), (v.get('roofline_prefill') or {}).get('achieved')) for k, v in d.get('extra_configs', {}).items()]";;
pmcprefillfetch)
  (cd /tmp && rm -rf /tmp/pmc_pf_fetch && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex prefill_attn -f csv -d /tmp/pmc_pf_fetch -o pf -- python $REPO/tools/prefill_bench.py > $OUT/prefill_under_pmc_fetch.json 2> $OUT/prefill_pmc_fetch.err; echo "pmcprefillfetch rc=$?")
  f=$(find /tmp/pmc_pf_fetch -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_prefill_fetch.py $f $OUT/prefill_pmc_fetch_summary.json | tail -40;;
final)
  # the whole validation of a tree: GPU suite, smoke, the bench line, rocprofv3 kernel stats of the bench
  timeout 1500 python -m pytest tests -m gpu -q -rf --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | cut -c1-300 | tail -20
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -2
  bash tools/gpu_round.sh $TAG benchfull benchprof;;
widesweep)
  # every (NT, NW, split) plan of the wide decode GEMM at the row counts in SWEEP_M on the shapes in SWEEP_SHAPES
  timeout 900 python tools/gemm_wide_sweep.py ${SWEEP_M:-131 144 256} > $OUT/gemm_wide_sweep.jsonl 2> $OUT/gemm_wide_sweep.err; echo "sweep rc=$?";;
skinnysweep)
  timeout 600 python tools/gemm_skinny_sweep.py ${SWEEP_M:-64 131 208} > $OUT/skinny_sweep.jsonl 2> $OUT/skinny_sweep.err; echo "skinny sweep rc=$?";;
prefillbench)
  timeout 600 python tools/prefill_bench.py > $OUT/prefill_bench.json 2> $OUT/prefill_bench.err; echo "prefill rc=$?"; cat $OUT/prefill_bench.json;;
cumask)
  # item (e) of the round-4 review: the externally launched TP = 2 stand-in (both ranks on the ONE GPU) latched a spin
  # timeout in 4 of 8 runs. Same command, alternating: ranks on DISJOINT halves of the compute units (HSA_CU_MASK per
  # rank, NVL_BENCH_CU_SPLIT=1) vs sharing all of them. One line per run: split, wall, p2p_status / value.
  export NVL_BENCH_SHARE_GPU=1 NVL_BENCH_BACKEND=gloo
  for i in ${CUMASK_RUNS:-1 0 1 0 1 0 1 1}; do
    T0=$(date +%s); NVL_BENCH_CU_SPLIT=$i timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29540 + RANDOM % 200)) bench.py --gpus 2 --tp 2 --steps 1 --warmup 0 --num-seqs 48 --num-kvcache-blocks 300 --no-cpu-baseline --no-roofline > $OUT/cumask_run.json 2> $OUT/cumask_run.err; rc=$?
    python - "$i" "$rc" "$(( $(date +%s) - T0 ))" $OUT/cumask_run.json $OUT/cumask_run.err >> $OUT/cumask_runs.jsonl <<'PY'
import json, sys
split, rc, wall, out, err = sys.argv[1:6]
rec = {"cu_split": int(split), "rc": int(rc), "wall_s": int(wall)}
try:
    d = json.loads([ln for ln in open(out) if ln.startswith("{")][-1])
    rec.update(value=d.get("value"), p2p_status=d["config"].get("p2p_status"), handoff=d["config"].get("p2p_handoff"))
except Exception as ex:
    rec["parse_error"] = repr(ex)
e = open(err).read()
rec["spin_limit_in_stderr"] = "spin limit" in e
print(json.dumps(rec))
PY
    tail -1 $OUT/cumask_runs.jsonl
  done
  unset NVL_BENCH_SHARE_GPU NVL_BENCH_BACKEND;;
lmhead256)
  SWEEP_SHAPES=lm_head_8b timeout 600 python tools/gemm_wide_m256.py 160 208 256 > $OUT/lm_head_m160_m256.json 2> $OUT/lm_head_m160_m256.err; echo "lmhead256 rc=$?"; cat $OUT/lm_head_m160_m256.json;;
pmcwaits)
  # wave-cycle split of the prefill-attention kernel (same counters as profiles/r03_prefill_pmc_waits.json)
  (cd /tmp && rm -rf /tmp/pmc_pf_waits && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --kernel-include-regex prefill_attn -f csv -d /tmp/pmc_pf_waits -o pf -- python $REPO/tools/prefill_bench.py > $OUT/prefill_under_pmc_waits.json 2> $OUT/prefill_pmc_waits.err; echo "pmcwaits rc=$?"; tail -c 300 $OUT/prefill_pmc_waits.err)
  f=$(find /tmp/pmc_pf_waits -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_group_summary.py $f prefill_attn $OUT/prefill_pmc_waits.json | tail -5;;
prefill64)
  # the generated one-wave-per-SIMD prefill loop (attn_prefill64.hip): random packed shapes against the oracle with the shape
  # forced, then the long-prompt shapes alternating with the 8-wave loop, then tools/prefill_bench.py's eight shapes
  NVL_PREFILL_W64=2 timeout 900 python tools/probes/prefill_fuzz.py 60 > $OUT/prefill_fuzz_w64.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/prefill_fuzz_w64.log
  NVL_PREFILL_W64=2 timeout 900 python tools/probes/prefill_fuzz.py 40 --paged > $OUT/prefill_fuzz_w64_paged.log 2>&1; echo "paged fuzz rc=$?"; tail -1 $OUT/prefill_fuzz_w64_paged.log
  for w in 1 0; do NVL_PREFILL_W64=$w timeout 200 python tools/probes/prefill_time_paged.py 2>/dev/null | tail -1 | sed "s/^/paged w64=$w /"; done | tee $OUT/prefill_w64_paged_ab.txt
  for w in 1 0 1 0; do NVL_PREFILL_W64=$w timeout 200 python tools/probes/prefill_time.py 2>/dev/null | tail -1 | sed "s/^/w64=$w /"; done | tee $OUT/prefill_w64_ab.txt
  timeout 300 python tools/prefill_bench.py > $OUT/prefill_bench.json 2> $OUT/prefill_bench.err; echo "prefill bench rc=$?"; cut -c1-1600 $OUT/prefill_bench.json;;
first8)
  # FIRST CONTACT with a multi-GPU node, unattended, most valuable evidence first (every sub-step has its own timeout; a
  # failure never stops the sequence). NGPUS=n (default: every visible GPU). FIRST8_DRY=1 = the one-GPU stand-in: 2 ranks
  # on GPU 0 with disjoint CU halves, gloo, Qwen3-0.6B shapes — checks that the sequence itself runs.
  #   1. tp.init_p2p's hand-off decision on the real links + P2P all-reduce us vs RCCL at the decode sizes (131 x 5120 ...)
  #   2. bench.py --model qwen3-32b --tp 2 / 4 / 8 (BASELINE's second metric; falls back to RCCL by itself if P2P latches)
  #   3. one rocprofv3 kernel trace of a TP = N decode-heavy pass (timeline of a decode step: where the collectives sit)
  N=${NGPUS:-$(python -c "import torch; print(torch.cuda.device_count())")}
  if [ "${FIRST8_DRY:-0}" = 1 ]; then
    export NVL_BENCH_SHARE_GPU=1 NVL_BENCH_BACKEND=gloo NVL_BENCH_CU_SPLIT=1; N=2; MODEL=qwen3-0.6b; SEQS="--num-seqs 32 --num-kvcache-blocks 200"; P2PMODE=""
  else
    MODEL=qwen3-32b; SEQS=""; P2PMODE="--multi-gpu"
  fi
  echo "first8: $N ranks, model $MODEL, dry=${FIRST8_DRY:-0}"
  timeout 600 python tools/p2p_bench.py $N $P2PMODE > $OUT/first8_p2p_vs_rccl_w$N.json 2> $OUT/first8_p2p.err; echo "first8 p2p rc=$?"; tail -c 300 $OUT/first8_p2p.err | grep -v amdgpu.ids; cut -c1-1500 $OUT/first8_p2p_vs_rccl_w$N.json
  for t in 2 4 8; do
    [ $t -gt $N ] && continue
    T0=$(date +%s); timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $t --master-addr 127.0.0.1 --master-port $((29600 + t)) bench.py --gpus $t --tp $t --model $MODEL --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs $SEQS > $OUT/first8_tp$t.json 2> $OUT/first8_tp$t.err; echo "first8 tp=$t rc=$? wall=$(( $(date +%s) - T0 )) s"
    grep -v "socket.cpp\|Gloo\|amdgpu.ids\|OMP_NUM\|\*\*\*" $OUT/first8_tp$t.err | tail -4; python - $OUT/first8_tp$t.json <<'PY'
import json, sys
try:
    d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
    c = d["config"]
    print("  value", d.get("value"), d.get("unit"), "| p2p", c.get("p2p_collectives"), c.get("p2p_handoff"), c.get("p2p_status"), "| attempt", d.get("tp_p2p_attempt"))
except Exception as ex:
    print("  no line:", repr(ex))
PY
  done
  (cd /tmp && rm -rf /tmp/prof_first8 && timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_first8 -o tp -- python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29650 $REPO/bench.py --gpus $N --tp $N --model $MODEL --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs --no-roofline --num-seqs 32 ${SEQS#--num-seqs 32} > $OUT/first8_under_rocprof.json 2> $OUT/first8_prof.err; echo "first8 prof rc=$?")
  for f in $(find /tmp/prof_first8 -name '*kernel_stats.csv' | head -$N); do cp $f $OUT/first8_kernel_stats_$(basename $(dirname $f)).csv; done
  f=$(find /tmp/prof_first8 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -14 $f | cut -c1-160
  unset NVL_BENCH_SHARE_GPU NVL_BENCH_BACKEND NVL_BENCH_CU_SPLIT;;
*) echo "unknown step $w";;
esac
done
