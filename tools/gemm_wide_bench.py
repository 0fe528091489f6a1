"""Wide-tile deep-K decode linear (nvl_linear_wide): correctness vs an fp32 reference + time vs hipBLASLt.
usage: python tools/gemm_wide_bench.py [model ...]   (8b 32b 32b_tp4 32b_tp8 lm_head; default 8b 32b_tp8)
Prints one JSON line: time_us[shape] = [ours_us (tile-packed weights), blas_us, ours_GBps, splits, ours_us with the
row-major weight stream]."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nano_vllm_amd import ops
ops.load_library()
BF16 = torch.bfloat16

MODELS = {
    "8b": [("qkv", 6144, 4096, 0), ("o", 4096, 4096, 2), ("gate_up", 24576, 4096, 1), ("down", 4096, 12288, 2)],
    "32b": [("qkv", 10240, 5120, 0), ("o", 5120, 8192, 2), ("gate_up", 51200, 5120, 1), ("down", 5120, 25600, 2)],
    "32b_tp4": [("qkv", 2560, 5120, 0), ("o", 5120, 2048, 2), ("gate_up", 12800, 5120, 1), ("down", 5120, 6400, 2)],
    "32b_tp8": [("qkv", 1280, 5120, 0), ("o", 5120, 1024, 2), ("gate_up", 6400, 5120, 1), ("down", 5120, 3200, 2)],
    "lm_head": [("lm_head", 151936, 1024, 0)],
    "lm_head_deep": [("lm_head_8b", 151936, 4096, 0), ("lm_head_32b", 151936, 5120, 0), ("lm_head_32b_tp8", 18992, 5120, 0)],
}


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


def check(m, n, k, mode):
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n + k + mode)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(BF16)
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(BF16)
    ref = x.float() @ w.float().t()
    if mode == 0:
        out = ops.linear_wide(x, w, 0).float()
        want = ref.to(BF16).float()
    elif mode == 1:
        out = ops.linear_wide(x, w, 1).float()
        gte, up = ref[:, : n // 2].to(BF16).float(), ref[:, n // 2:].to(BF16).float()
        want = (F.silu(gte) * up).to(BF16).float()
    else:
        out = ops.linear_wide(x, w, 2).sum(0)
        want = ref
    return ((out - want).abs().max() / want.abs().max()).item()


def main():
    models = sys.argv[1:] or ["8b", "32b_tp8"]
    res = {"relerr": {}, "time_us": {}, "env": {k: v for k, v in os.environ.items() if k.startswith("NVL_WIDE")}}
    ms = [int(v) for v in os.environ.get("BENCH_M", "16,64,144,256").split(",")]
    for model in models:
        for name, n, k, mode in MODELS[model]:
            for m in (1, 16, 131, 144, 200, 256, 300):
                if ops.linear_wide_plan(m, n, k, mode) and not (model in ("32b", "lm_head", "lm_head_deep") and m not in (131, 256)):
                    res["relerr"][f"{model}_{name}_m{m}"] = round(check(m, n, k, mode), 5)
            for m in ms:
                plan = ops.linear_wide_plan(m, n, k, mode)
                ncopy = max(2, min(12, int(0.8e9 // (n * k * 2))))
                ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF16) for _ in range(ncopy)]
                x = torch.randn(m, k, device="cuda").to(BF16)
                t_ours = t_rowmajor = float("nan")
                if plan:
                    out = ops.linear_wide(x, ws[0], mode)
                    scratch = torch.empty(max(plan[1], 16), dtype=torch.uint8, device="cuda")
                    def ours():
                        for w in ws:
                            ops.linear_wide(x, w, mode, out=out, workspace=scratch)
                    t_rowmajor = graph_time(ours) / len(ws)
                    pk = [ops.pack_weight_tiles(w) for w in ws]
                    def ours_packed():
                        for w in pk:
                            ops.linear_wide(x, w, mode, out=out, workspace=scratch, packed=True)
                    t_ours = graph_time(ours_packed) / len(ws)
                    del pk
                def blas():
                    for w in ws:
                        F.linear(x, w)
                t_blas = graph_time(blas) / len(ws)
                res["time_us"][f"{model}_{name}_m{m}"] = [round(t_ours, 2), round(t_blas, 2),
                                                          round(n * k * 2 / t_ours / 1e3, 1), plan[0] if plan else 0,
                                                          round(t_rowmajor, 2)]
                del ws
    res["relerr_max"] = max(res["relerr"].values()) if res["relerr"] else None
    print(json.dumps(res))


if __name__ == "__main__":
    main()
