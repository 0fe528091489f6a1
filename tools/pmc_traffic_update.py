"""Fold one rocprofv3 PMC pass of the decode-attention replay into profiles/pmc_traffic.json.

    python tools/pmc_traffic_update.py <pmc_summary.json> <replay_under_pmc.json> <model> "<kernel label>"

<pmc_summary.json>   : tools/pmc_summary.py output of `rocprofv3 --pmc FETCH_SIZE --kernel-include-regex decode_ --
                       python tools/attn_replay.py --fused --reps 1 [--hq .. --hkv .. --layers ..]`
<replay_under_pmc.json>: the JSON line that replay printed (algorithmic_bytes_per_launch of the same launches)
Entry written: kernels[<kernel label>] = {hbm_read_bytes_over_algorithmic, ...}. FETCH_SIZE is in KiB and, on gfx950,
counts HALF the bytes of a 16 B/lane streaming read (MI355X_MICROARCH.md, HBM section): doubled here.
bench.py multiplies a run's algorithmic bytes per launch by this ratio for `roofline.traffic`.
"""
import hashlib
import json
import os
import sys

summary, replay, model, label = sys.argv[1:5]
rows = json.load(open(summary))
rep = json.loads([ln for ln in open(replay).read().splitlines() if ln.startswith("{")][-1])
fetch = {r["kernel"]: r for r in rows if r["counter"] == "FETCH_SIZE"}
# (template instantiations show up with truncated mangled names: the attention kernel is the row that is neither the
# split merge nor the per-step plan — and by far the largest)
main = max((v for k, v in fetch.items() if "combine" not in k and "plan" not in k and "prefix" not in k), key=lambda v: v["mean"])
comb = next((v for k, v in fetch.items() if "combine" in k), None)
# the shared-prefix pass when it is a launch of its own (NVL_PX_SEPARATE=1 runs; label names it): its share of a call's bytes. In
# the shipped form the packs are workgroups of the main launch (decode_mfma8_shared_kernel: the `main` row)
pref = next((v for k, v in fetch.items() if "prefix" in k), None) if "prefix" in label else None
per_launch_kib = main["mean"] + (comb["mean"] if comb else 0.0) + (pref["mean"] * pref["dispatches"] / main["dispatches"] if pref else 0.0)
hbm = per_launch_kib * 1024 * 2
alg = rep["algorithmic_bytes_per_launch"]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
try:
    db = json.load(open(path))
except OSError:
    db = {}
if "kernels" not in db:                         # round-2 layout: one entry at the top level
    old = dict(db)
    db = {"kernels": {}}
    if "kernel_name" in old:
        db["kernels"][old["kernel_name"]] = old
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(root, "nano_vllm_amd", "csrc", "attn_decode.hip"), "rb") as fh:
    src_sha = hashlib.sha256(fh.read()).hexdigest()[:16]
db["kernels"][label] = {
    # the kernel source this pass measured: bench.py applies the ratio only while attn_decode.hip still hashes to this
    "attn_decode_hip_sha16": src_sha,
    "source": f"{os.path.basename(summary)} (rocprofv3 --pmc FETCH_SIZE --kernel-include-regex decode_ -- python "
              f"tools/attn_replay.py --fused --reps 1 ...)",
    "model": model, "dispatches": main["dispatches"],
    "FETCH_SIZE_mean_KiB": {"main": main["mean"], "decode_stream_combine_kernel": comb["mean"] if comb else None,
                            **({"decode_prefix_kernel": pref["mean"]} if pref else {})},
    "correction": "x2: on gfx950 FETCH_SIZE reports half the bytes of a 16 B/lane streaming read (MI355X_MICROARCH.md, HBM section)",
    "hbm_read_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "hbm_read_bytes_over_algorithmic": hbm / alg,
}
json.dump(db, open(path, "w"), indent=1)
print(label, "traffic / algorithmic =", round(hbm / alg, 4))
