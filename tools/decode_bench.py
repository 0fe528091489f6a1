"""Decode-attention micro-benchmark + correctness check of the shipped kernel.
Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
from oracle import ops as ref
BF16 = torch.bfloat16
ops.load_library()
res = {}

def check(lens, hq, hkv):
    bs = 256
    gen = torch.Generator().manual_seed(1)
    nb = [(n + bs - 1)//bs for n in lens]
    total = sum(nb) + 2
    kc = torch.randn(total, bs, hkv, 128, generator=gen).to(BF16)
    vc = torch.randn(total, bs, hkv, 128, generator=gen).to(BF16)
    perm = torch.randperm(total, generator=gen).tolist()
    bt = torch.full((len(lens), 16), -1, dtype=torch.int32)
    c = 0
    for s, n in enumerate(nb):
        for j in range(n):
            bt[s, j] = perm[c]; c += 1
    q = torch.randn(len(lens), hq, 128, generator=gen).to(BF16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    o_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, 128 ** -0.5).squeeze(1)
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(len(lens), hq, 4096), dtype=torch.uint8, device="cuda")
    o = ops.paged_attn_decode(q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda(), bt.cuda(), ctx.cuda(), 128 ** -0.5, 4096, ws)
    return float((o.cpu().float() - o_ref.float()).abs().max() / o_ref.float().abs().max())

res["relerr"] = [check([1, 31, 32, 33, 255, 256, 257, 1000, 4096, 0, 77], 16, 8), check([100, 2048, 5], 8, 8),
                 check([300, 700], 32, 8), check([129, 1025], 8, 1)]

def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def case(batch, lo, hi, hq, hkv, seed=0, layers=8):
    gen = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (batch,), generator=gen)
    bs = 256
    nb = (lens + bs - 1) // bs
    total = int(nb.sum()) + 8
    caches = [(torch.randn(total, hkv, bs, 128, device="cuda").to(BF16), torch.randn(total, hkv, bs, 128, device="cuda").to(BF16)) for _ in range(layers)]
    perm = torch.randperm(total, generator=gen)
    bt = torch.full((batch, 16), -1, dtype=torch.int32)
    c = 0
    for i in range(batch):
        bt[i, : nb[i]] = perm[c: c + nb[i]].to(torch.int32); c += int(nb[i])
    q = torch.randn(batch, hq, 128, device="cuda").to(BF16)
    ctx = lens.to(torch.int32).cuda(); btd = bt.cuda()
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(batch, hq, 4096), dtype=torch.uint8, device="cuda")
    o = torch.empty_like(q)
    def fn():
        for kc, vc in caches:   # rotate over distinct caches: cold K/V like consecutive layers
            ops.paged_attn_decode(q, kc, vc, btd, ctx, 128 ** -0.5, 4096, ws, out=o)
    t = timeit(fn, iters=10) / layers
    return round(int(lens.sum()) * 2 * hkv * 128 * 2 / t / 1e9, 1), round(t * 1e6, 1)

res["b131"] = case(131, 100, 2048, 16, 8)
res["b256"] = case(256, 100, 2048, 16, 8)
res["b64"] = case(64, 100, 2048, 16, 8)
res["b16"] = case(16, 100, 2048, 16, 8)
res["b256_g4"] = case(256, 100, 2048, 32, 8)
res["b256_g8"] = case(256, 100, 2048, 8, 1)
res["b256_g1"] = case(256, 100, 2048, 8, 8)
res["b256_g8_tp4_long"] = case(256, 1024, 4096, 16, 2)     # Qwen3-32B TP=4 per-rank heads, long contexts
print(json.dumps(res))
