"""Skinny decode-linear micro-benchmark + correctness vs an fp32 reference; prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nano_vllm_amd import ops
ops.load_library()
BF16 = torch.bfloat16
res = {}


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def graph_time(fn, reps=20, iters=20):
    """Per-call time inside a hipGraph of `reps` back-to-back calls (what the decode step sees)."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return timeit(g.replay, iters=iters, warm=3) / reps


def check(m, n, k, mode):
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n + k + mode)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(BF16)
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(BF16)
    ref = x.float() @ w.float().t()
    if mode == 0:
        out = ops.linear_decode(x, w, 0).float()
        want = ref.to(BF16).float()
    elif mode == 1:
        out = ops.linear_decode(x, w, 1).float()
        gte, up = ref[:, : n // 2].to(BF16).float(), ref[:, n // 2:].to(BF16).float()
        want = (F.silu(gte) * up).to(BF16).float()
    else:
        out = ops.linear_decode(x, w, 2).sum(0)
        want = ref
    err = (out - want).abs().max().item()
    return err / want.abs().max().item()


MODELS = {
    "0.6b": [("qkv", 4096, 1024, 0), ("o", 1024, 2048, 2), ("gate_up", 6144, 1024, 1), ("down", 1024, 3072, 2)],
    "lm_head": [("lm_head_0.6b", 151936, 1024, 0)],
    "32b_tp8": [("qkv", 1280, 5120, 0), ("o", 5120, 1024, 2), ("gate_up", 6400, 5120, 1)],
    "32b_tp4": [("qkv", 2560, 5120, 0), ("o", 5120, 2048, 2), ("gate_up", 12800, 5120, 1), ("down", 5120, 6400, 2)],
    "8b": [("qkv", 6144, 4096, 0), ("o", 4096, 4096, 2), ("gate_up", 24576, 4096, 1), ("down", 4096, 12288, 2)],
    "32b": [("qkv", 10240, 5120, 0), ("o", 5120, 8192, 2), ("gate_up", 51200, 5120, 1), ("down", 5120, 25600, 2)],
}
model = sys.argv[1] if len(sys.argv) > 1 else "0.6b"
shapes = MODELS[model]
res["model"] = model
res["relerr"] = {}
for name, n, k, mode in shapes:
    for m in ((1, 16, 131, 144, 256, 300, 512) if model == "0.6b" else (16, 131, 144, 256)):
        if ops.linear_decode_splits(m, n, k, mode):
            res["relerr"][f"{name}_m{m}"] = check(m, n, k, mode)
res["relerr_max"] = max(res["relerr"].values()) if res["relerr"] else None

res["time_us"] = {}
MS = os.environ.get("NVL_BENCH_MS")   # e.g. "64,144": time these row counts only
for m in (tuple(int(v) for v in MS.split(",")) if MS else
          ((16, 32, 64, 96, 144, 208, 256, 512) if model == "0.6b" else (16, 64, 144, 256))):
    for name, n, k, mode in shapes:
        # rotate over several weight copies so the weights come from HBM like in the real step
        ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF16) for _ in range(24 if model == "0.6b" else 6)]
        x = torch.randn(m, k, device="cuda").to(BF16)
        covered = bool(ops.linear_decode_splits(m, n, k, mode))
        outs = ops.linear_decode(x, ws[0], mode) if covered else None
        def ours():
            for w in ws:
                ops.linear_decode(x, w, mode, out=outs)
        pk = [ops.pack_weight_tiles(w) for w in ws] if covered else []
        def ours_packed():
            for w in pk:
                ops.linear_decode(x, w, mode, out=outs, packed=True)
        def blas():
            for w in ws:
                F.linear(x, w)
        t_rowmajor = graph_time(ours, reps=1) / len(ws) if covered else float("nan")
        t_ours = graph_time(ours_packed, reps=1) / len(ws) if covered else float("nan")
        t_blas = graph_time(blas, reps=1) / len(ws)
        t_warm = None
        if covered and os.environ.get("NVL_BENCH_WARM") == "1":
            # upper bound of what a perfect L2 prefetch of the weights could buy: every weight twice back to back,
            # the second call finds its tiles in the XCD's L2 (same workgroup -> XCD mapping)
            def twice():
                for w in pk:
                    ops.linear_decode(x, w, mode, out=outs, packed=True)
                    ops.linear_decode(x, w, mode, out=outs, packed=True)
            t_warm = graph_time(twice, reps=1) / len(ws) - t_ours
        # [ours with tile-packed weights (what the engine runs), hipBLASLt, GB/s of ours, ours with row-major weights]
        res["time_us"][f"{name}_m{m}"] = [round(t_ours, 2), round(t_blas, 2), round(n * k * 2 / t_ours / 1e3, 1),
                                          round(t_rowmajor, 2)] + ([round(t_warm, 2)] if t_warm is not None else [])
print(json.dumps(res))
