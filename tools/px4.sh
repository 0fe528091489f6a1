# kernel durations of the attention replay on config 3's own schedule, per arm (old = nano_vllm_amd/lib_probes_b; n = NVL_PX_ITEMS_PER_WG; nf = n with the pack workgroups first)
set -u
OUT=gpurun_out/${TAG:-r06x}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
for x in ${ARMS:-old 4 6}; do
  unset NVL_PX_FIRST
  if [ $x = old ]; then export NVL_LIBDIR=$REPO/nano_vllm_amd/lib_probes_b; unset NVL_PX_ITEMS_PER_WG; else unset NVL_LIBDIR; export NVL_PX_ITEMS_PER_WG=${x%f}; [ $x != ${x%f} ] && export NVL_PX_FIRST=1; fi
  (cd /tmp && rm -rf /tmp/prof_px && timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /tmp/prof_px -o replay -- python $REPO/tools/attn_replay.py --fused --hq 32 --hkv 8 --layers 36 --every 8 --workload prefix --pool-blocks 6432 > $REPO/$OUT/replay_$x.json 2>$REPO/$OUT/replay_$x.err; f=$(find /tmp/prof_px -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$OUT/replay_kernel_stats_$x.csv)
  echo "arm $x: $(grep -i "decode_mfma8\|decode_prefix" $OUT/replay_kernel_stats_$x.csv | cut -d, -f4 | tr '\n' ' ') $(python -c "
import json
d=json.loads([l for l in open('$OUT/replay_$x.json') if l.startswith('{')][-1]); print(round(d['avg_launch_us'],1))")"
done
