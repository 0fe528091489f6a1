"""Time the paged (prefix-cache / chunk-continuation) prefill attention: Lq new tokens behind Lk - Lq cached ones, K / V from the
paged cache. python tools/probes/prefill_time_paged.py ; NVL_PREFILL_W64=0|1 selects the loop."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nano_vllm_amd import ops
ops.load_library()
out = {}
for lq, lk, hq, hkv in [(8192, 16384, 16, 8), (4096, 20480, 16, 8), (8192, 8192 + 512, 64, 8)]:
    bs = 256
    nb = (lk + bs - 1) // bs
    kc = torch.randn(nb + 2, hkv, bs, 128, device="cuda").to(torch.bfloat16)
    vc = torch.randn(nb + 2, hkv, bs, 128, device="cuda").to(torch.bfloat16)
    bt = (torch.randperm(nb, device="cuda").to(torch.int32) + 1).view(1, nb)
    q = torch.randn(lq, hq, 128, device="cuda").to(torch.bfloat16)
    cuq = torch.tensor([0, lq], dtype=torch.int32, device="cuda"); cuk = torch.tensor([0, lk], dtype=torch.int32, device="cuda")
    o = torch.empty_like(q)
    fn = lambda: ops.attn_prefill_varlen(q, kc, vc, cuq, cuk, lq, 128 ** -0.5, block_tables=bt, out=o)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    pairs = lq * (lk - lq) + lq * (lq + 1) // 2
    out[f"lq{lq}_lk{lk}_{hq}/{hkv}"] = [round(t * 1e6, 1), round(4 * hq * 128 * pairs / t / 1e12, 1)]
print(json.dumps(out))
