"""Prefill attention micro-benchmark (MFMA utilisation) + correctness spot check; prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
from oracle import ops as ref
ops.load_library()
BF16 = torch.bfloat16
res = {"cases": []}


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def case(lens, hq, hkv, name):
    n = sum(lens)
    q = torch.randn(n, hq, 128, device="cuda").to(BF16)
    k = torch.randn(n, hkv, 128, device="cuda").to(BF16)
    v = torch.randn(n, hkv, 128, device="cuda").to(BF16)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
    o = torch.empty_like(q)
    fn = lambda: ops.attn_prefill_varlen(q, k, v, cu, cu, max(lens), 128 ** -0.5, out=o)
    t = timeit(fn)
    pairs = sum(l * (l + 1) // 2 for l in lens)
    tf = 4 * hq * 128 * pairs / t / 1e12
    res["cases"].append(dict(name=name, tokens=n, seqs=len(lens), hq=hq, hkv=hkv, us=round(t * 1e6, 1), TFLOPs=round(tf, 1),
                             frac_of_2500=round(tf / 2500, 4)))


# correctness spot check vs the CPU oracle (small)
lens = [300, 129, 64]
n = sum(lens)
g = torch.Generator().manual_seed(0)
q = torch.randn(n, 8, 128, generator=g).to(BF16); k = torch.randn(n, 2, 128, generator=g).to(BF16); v = torch.randn(n, 2, 128, generator=g).to(BF16)
cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
o = ops.attn_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), cu.cuda(), cu.cuda(), max(lens), 128 ** -0.5).cpu()
o_ref = ref.flash_attn_varlen_func(q, k, v, max(lens), cu, max(lens), cu, 128 ** -0.5)
res["relerr"] = float((o.float() - o_ref.float()).abs().max() / o_ref.float().abs().max())

case([1024] * 16, 16, 8, "0.6B 16x1024")
case([561] * 29, 16, 8, "bench-like 29x561")
import random
random.seed(0)
ragged = []
while sum(ragged) + 1024 <= 16384:
    ragged.append(random.randint(100, 1024))
case(ragged, 16, 8, f"bench ragged U[100,1024] x{len(ragged)}")           # what a prefill step of bench.py looks like
case([random.randint(100, 180) for _ in range(110)], 16, 8, "110 short U[100,180]")  # > 64 sequences: LDS tile list
case([4096] * 4, 16, 8, "0.6B 4x4096")
case([16384], 8, 1, "32B/TP8 1x16384 (config 5)")
case([16384], 16, 8, "0.6B 1x16384")
case([2048] * 8, 64, 8, "32B 8x2048 G=8")
print(json.dumps(res))
