#!/bin/bash
# round 4, call 2: 64-column k step of the wide GEMM at 193-256 rows (tests, A/B vs 128 columns vs hipBLASLt, config 3 both
# ways), batched combine loop (tests), where the G = 8 / one-kv-head decode attention spends its time (rocprofv3)
set -u
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests -m gpu -q -rf -k "linear_wide or full_width or fused_lm_head_sampled or paged_attn_decode or fp8_kv_decode" > $OUT/pytest_sel.log 2>&1; echo "selected tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_sel.log | cut -c1-300 | tail -20
timeout 600 python tools/gemm_wide_m256.py 208 256 > $OUT/gemm_wide_m256.json 2> $OUT/gemm_wide_m256.err; echo "m256 rc=$?"; cat $OUT/gemm_wide_m256.err | grep -v amdgpu.ids | tail -24
for w in auto 1; do
  NVL_GEMM_WIDE=$w timeout 600 python bench.py --model qwen3-8b --workload prefix --no-cpu-baseline --no-roofline --warmup 0 > $OUT/cfg3_wide_$w.json 2> $OUT/cfg3_wide_$w.err; echo "cfg3 wide=$w rc=$?"; cut -c1-260 $OUT/cfg3_wide_$w.json; echo
done
(cd /tmp && rm -rf /tmp/prof_g8 && timeout 600 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_g8 -o g8 -- python $REPO/tools/attn_replay.py --fused --hq 8 --hkv 1 --layers 64 --every 16 > $REPO/$OUT/replay_g8_under_rocprof.json 2> $REPO/$OUT/replay_g8_prof.err; echo "prof g8 rc=$?")
f=$(find /tmp/prof_g8 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/replay_g8_kernel_stats.csv && head -6 $OUT/replay_g8_kernel_stats.csv | cut -c1-200
cat $OUT/replay_g8_under_rocprof.json
