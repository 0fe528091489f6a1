"""lm_head + sampler: fused kernel (nvl_lmhead_sample) vs hipBLASLt GEMM + nvl_sample; per-call us inside a hipGraph,
weights rotated through several copies so they come from HBM as in the real step. Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nano_vllm_amd import ops
ops.load_library()
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
    return s.elapsed_time(e) / iters * 1e3


def graph_time(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return timeit(g.replay)


res = {"cases": []}
for name, v, k, copies in (("qwen3-0.6b", 151936, 1024, 4), ("qwen3-8b", 151936, 4096, 2), ("qwen3-32b/tp8", 18992, 5120, 8)):
    ws_list = [(torch.randn(v, k, device="cuda") * 0.05).to(BF16) for _ in range(copies)]
    for b in (16, 64, 131, 144, 192):
        x = (torch.randn(b, k, device="cuda") * 0.5).to(BF16)
        temps = torch.full((b,), 0.6, device="cuda")
        out = torch.empty(b, dtype=torch.int64, device="cuda")
        wsf = torch.empty(max(ops.lmhead_sample_workspace_bytes(144, v, k), ops.lmhead_sample_workspace_bytes(192, v, k)),
                          dtype=torch.uint8, device="cuda")
        wss = torch.empty(ops.sample_workspace_bytes(512), dtype=torch.uint8, device="cuda")

        def fused():
            for w in ws_list:
                ops.lmhead_sample(x, w, temps, 1, 2, wsf, out=out)

        def split():
            for w in ws_list:
                ops.sample(F.linear(x, w), temps, 1, 2, wss, out=out)

        def gemm_only():
            for w in ws_list:
                F.linear(x, w)

        tf, ts, tg = (graph_time(f) / copies for f in (fused, split, gemm_only))
        res["cases"].append(dict(model=name, batch=b, fused_us=round(tf, 1), gemm_plus_sample_us=round(ts, 1),
                                 gemm_only_us=round(tg, 1), fused_weight_GBps=round(v * k * 2 / tf / 1e3, 1)))
print(json.dumps(res))
