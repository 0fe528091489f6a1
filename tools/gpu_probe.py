"""One-shot MI355X probe: environment facts + micro-benchmarks of the HIP kernels.
Writes gpurun_out/probe.json. Run: python tools/gpu_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nano_vllm_amd import ops

out = {}
os.makedirs("gpurun_out", exist_ok=True)
p = torch.cuda.get_device_properties(0)
free, total = torch.cuda.mem_get_info()
out["device"] = dict(name=p.name, cus=p.multi_processor_count, total_mem=total, free_mem=free,
                     warp=getattr(p, "warp_size", None), gcn=getattr(p, "gcnArchName", None))
out["host"] = dict(cpus=os.cpu_count())
out["torch"] = dict(version=torch.__version__, hip=torch.version.hip)
ops.load_library()
out["nvl_cu_count"] = ops.lib().nvl_device_cu_count()
BF16 = torch.bfloat16


def timeit(fn, iters=20, warm=3):
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


# --- graph capture of a ctypes launch ------------------------------------------------------
try:
    x = torch.randn(64, 1024, device="cuda").to(BF16)
    w = torch.ones(1024, device="cuda", dtype=BF16)
    y = torch.empty_like(x)
    ops.rmsnorm(x, w, 1e-6, out=y)
    torch.cuda.synchronize()
    y_eager = y.clone()
    y.zero_()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ops.rmsnorm(x, w, 1e-6, out=y)
    y.zero_()
    gr.replay()
    torch.cuda.synchronize()
    out["graph_capture_ok"] = bool(torch.equal(y, y_eager))
except Exception as ex:  # noqa
    out["graph_capture_ok"] = f"FAILED: {ex!r}"

# --- yardsticks ------------------------------------------------------------------------------
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a), iters=10)
out["copy_1GiB_TBps_rw"] = 2 * (1 << 30) / t / 1e12
del a, b
A = torch.randn(8192, 8192, device="cuda").to(BF16)
B = torch.randn(8192, 8192, device="cuda").to(BF16)
t = timeit(lambda: A @ B, iters=10)
out["gemm_8192_TFLOPs"] = 2 * 8192 ** 3 / t / 1e12
del A, B
# decode-shaped GEMMs (0.6B): M=131
for (m, n, k) in [(131, 4096, 1024), (131, 1024, 2048), (131, 6144, 1024), (131, 1024, 3072), (131, 151936, 1024),
                  (256, 4096, 1024), (256, 151936, 1024)]:
    X = torch.randn(m, k, device="cuda").to(BF16)
    W = torch.randn(n, k, device="cuda").to(BF16)
    t = timeit(lambda: torch.nn.functional.linear(X, W), iters=30)
    out[f"linear_{m}x{n}x{k}_us"] = t * 1e6
    out[f"linear_{m}x{n}x{k}_GBps"] = (n * k * 2) / t / 1e9

# --- decode attention: bench-like batch ------------------------------------------------------
def decode_case(batch, lo, hi, hq, hkv, seed=0):
    gen = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (batch,), generator=gen)
    bs = 256
    nb = ((lens + bs - 1) // bs)
    total = int(nb.sum()) + 8
    kc = torch.randn(total, hkv, bs, 128, device="cuda").to(BF16)
    vc = torch.randn(total, hkv, bs, 128, device="cuda").to(BF16)
    perm = torch.randperm(total, generator=gen)
    max_ctx = 4096
    bt = torch.full((batch, max_ctx // bs), -1, dtype=torch.int32)
    c = 0
    for i in range(batch):
        bt[i, : nb[i]] = perm[c: c + nb[i]].to(torch.int32)
        c += int(nb[i])
    q = torch.randn(batch, hq, 128, device="cuda").to(BF16)
    ctx = lens.to(torch.int32).cuda()
    btd = bt.cuda()
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(batch, hq, max_ctx), dtype=torch.uint8, device="cuda")
    o = torch.empty_like(q)
    fn = lambda: ops.paged_attn_decode(q, kc, vc, btd, ctx, 128 ** -0.5, max_ctx, ws, out=o)
    t = timeit(fn, iters=30)
    bytes_ = int(lens.sum()) * 2 * hkv * 128 * 2
    return dict(batch=batch, tokens=int(lens.sum()), us=t * 1e6, GBps=bytes_ / t / 1e9)


out["decode_attn"] = []
for (bsz, lo, hi, hq, hkv) in [(131, 100, 2048, 16, 8), (256, 100, 2048, 16, 8), (256, 1024, 2048, 16, 8),
                               (32, 100, 2048, 16, 8), (256, 100, 2048, 32, 8), (256, 100, 2048, 8, 1),
                               (1, 4096, 4096, 16, 8)]:
    try:
        r = decode_case(bsz, lo, hi, hq, hkv)
        r.update(hq=hq, hkv=hkv, lo=lo, hi=hi)
        out["decode_attn"].append(r)
    except Exception as ex:  # noqa
        out["decode_attn"].append(dict(error=repr(ex), batch=bsz, hq=hq, hkv=hkv))

# --- prefill attention ---------------------------------------------------------------------
def prefill_case(lens, hq, hkv):
    n = sum(lens)
    q = torch.randn(n, hq, 128, device="cuda").to(BF16)
    k = torch.randn(n, hkv, 128, device="cuda").to(BF16)
    v = torch.randn(n, hkv, 128, device="cuda").to(BF16)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
    o = torch.empty_like(q)
    fn = lambda: ops.attn_prefill_varlen(q, k, v, cu, cu, max(lens), 128 ** -0.5, out=o)
    t = timeit(fn, iters=5, warm=2)
    pairs = sum(l * (l + 1) // 2 for l in lens)
    return dict(tokens=n, us=t * 1e6, TFLOPs=4 * hq * 128 * pairs / t / 1e12)


out["prefill_attn"] = []
for lens, hq, hkv in [([1024] * 16, 16, 8), ([4096] * 4, 16, 8), ([16384], 8, 1), ([561] * 29, 16, 8)]:
    try:
        r = prefill_case(lens, hq, hkv)
        r.update(hq=hq, hkv=hkv, seqs=len(lens), len=lens[0])
        out["prefill_attn"].append(r)
    except Exception as ex:  # noqa
        out["prefill_attn"].append(dict(error=repr(ex)))

# --- small kernels ---------------------------------------------------------------------------
x = torch.randn(16384, 1024, device="cuda").to(BF16)
r = torch.randn(16384, 1024, device="cuda").to(BF16)
w = torch.ones(1024, device="cuda", dtype=BF16)
y = torch.empty_like(x)
t = timeit(lambda: ops.add_rmsnorm(x, r, w, 1e-6, out=y))
out["add_rmsnorm_16384x1024_GBps"] = x.numel() * 2 * 4 / t / 1e9
x2 = torch.randn(16384, 6144, device="cuda").to(BF16)
y2 = torch.empty(16384, 3072, device="cuda", dtype=BF16)
t = timeit(lambda: ops.silu_mul(x2, out=y2))
out["silu_mul_16384x3072_GBps"] = (x2.numel() + y2.numel()) * 2 / t / 1e9
lg = torch.randn(256, 151936, device="cuda").to(BF16)
tt = torch.full((256,), 0.6, device="cuda")
ws = torch.empty(ops.sample_workspace_bytes(256), dtype=torch.uint8, device="cuda")
so = torch.empty(256, dtype=torch.int64, device="cuda")
t = timeit(lambda: ops.sample(lg, tt, 1, 0, ws, out=so))
out["sample_256x151936_us"] = t * 1e6
out["sample_256x151936_GBps"] = lg.numel() * 2 / t / 1e9
xs = torch.randn(131, 1024, device="cuda").to(BF16)
rs = torch.randn(131, 1024, device="cuda").to(BF16)
ys = torch.empty_like(xs)
t = timeit(lambda: ops.add_rmsnorm(xs, rs, w, 1e-6, out=ys), iters=200)
out["add_rmsnorm_131x1024_us_eager_call"] = t * 1e6

json.dump(out, open("gpurun_out/probe.json", "w"), indent=1)
print(json.dumps(out, indent=1))
