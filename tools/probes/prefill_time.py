"""Time one prefill-attention shape (HIP events, TFLOP/s): python tools/probes/prefill_time.py [lens hq hkv]; used by the
deletion probes of the ping-pong kernel (tools/probes/prefill_pp_parts.sh) — results of probe variants are NOT checked."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nano_vllm_amd import ops
ops.load_library()
shapes = [([16384], 16, 8), ([4096] * 4, 16, 8), ([2048] * 8, 64, 8)]
out = {}
for lens, hq, hkv in shapes:
    n = sum(lens)
    q = torch.randn(n, hq, 128, device="cuda").to(torch.bfloat16); k = torch.randn(n, hkv, 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(n, hkv, 128, device="cuda").to(torch.bfloat16)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
    o = torch.empty_like(q)
    fn = lambda: ops.attn_prefill_varlen(q, k, v, cu, cu, max(lens), 128 ** -0.5, out=o)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    pairs = sum(l * (l + 1) // 2 for l in lens)
    out[f"{len(lens)}x{lens[0]}_{hq}/{hkv}"] = [round(t * 1e6, 1), round(4 * hq * 128 * pairs / t / 1e12, 1)]
print(json.dumps(out))
