"""nvl_linear_wide at 193-256 rows (ONE row group of 16 row tiles): the 64-column k step (three x stages, round 4) against
the 128-column step (two stages, round 3: NVL_WIDE_BK=128) and against hipBLASLt (+ the separate SiLU launch for gate_up),
on the full-width Qwen3-8B / 32B projections and the per-rank TP = 4 gate_up, tile-packed weights; plus the relative
error of the 64-column form against an fp32 reference.
usage: python tools/gemm_wide_m256.py [m ...]    (default 208 256)
Prints one JSON line: time_us[shape_m] = [bk64_us, bk128_us, blas_us (incl. SiLU where it applies), bk64 GB/s]."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nano_vllm_amd import ops
ops.load_library()
BF16 = torch.bfloat16
SHAPES = {"8b_qkv": (6144, 4096, 0), "8b_o": (4096, 4096, 2), "8b_gate_up": (24576, 4096, 1), "8b_down": (4096, 12288, 2),
          "32b_qkv": (10240, 5120, 0), "32b_o": (5120, 8192, 2), "32b_gate_up": (51200, 5120, 1), "32b_down": (5120, 25600, 2),
          "32b_tp4_gate_up": (12800, 5120, 1), "32b_tp8_gate_up": (6400, 5120, 1), "lm_head_8b": (151936, 4096, 0)}


def timeit(fn, iters=12):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def relerr(m, n, k, mode):
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n + k + mode)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(BF16)
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(BF16)
    ref = x.float() @ w.float().t()
    pk = ops.pack_weight_tiles(w)
    if mode == 0:
        out, want = ops.linear_wide(x, pk, 0, packed=True).float(), ref.to(BF16).float()
    elif mode == 1:
        out = ops.linear_wide(x, pk, 1, packed=True).float()
        want = (F.silu(ref[:, : n // 2].to(BF16).float()) * ref[:, n // 2:].to(BF16).float()).to(BF16).float()
    else:
        out, want = ops.linear_wide(x, pk, 2, packed=True).sum(0), ref
    return ((out - want).abs().max() / want.abs().max()).item()


def main():
    ms = [int(a) for a in sys.argv[1:]] or [208, 256]
    only = os.environ.get("SWEEP_SHAPES")
    res = {"time_us": {}, "relerr_bk64": {}}
    for name, (n, k, mode) in SHAPES.items():
        if only and name not in only.split(","):
            continue
        ncopy = max(2, min(6, int(0.8e9 // (n * k * 2))))
        ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF16) for _ in range(ncopy)]
        pk = [ops.pack_weight_tiles(w) for w in ws]
        for m in ms:
            x = torch.randn(m, k, device="cuda").to(BF16)
            t = {}
            for bk in (64, 128):
                os.environ.pop("NVL_WIDE_BK", None)
                if bk == 128:
                    os.environ["NVL_WIDE_BK"] = "128"
                ops._wide_cache.clear()
                plan = ops.linear_wide_plan(m, n, k, mode)
                if not plan:
                    t[bk] = float("nan")
                    continue
                if bk == 64:
                    res["relerr_bk64"][f"{name}_m{m}"] = round(relerr(m, n, k, mode), 5)
                out = ops.linear_wide(x, pk[0], mode, packed=True)
                scratch = torch.empty(max(plan[1], 16), dtype=torch.uint8, device="cuda")

                def ours():
                    for w in pk:
                        ops.linear_wide(x, w, mode, out=out, workspace=scratch, packed=True)
                t[bk] = timeit(ours) / len(pk)
            os.environ.pop("NVL_WIDE_BK", None)
            ops._wide_cache.clear()

            def blas():
                for w in ws:
                    y = F.linear(x, w)
                    if mode == 1:
                        ops.silu_mul(y)
            t_blas = timeit(blas) / len(ws)
            res["time_us"][f"{name}_m{m}"] = [round(t[64], 2), round(t[128], 2), round(t_blas, 2),
                                              round(n * k * 2 / t[64] / 1e3)]
            print(f"{name}_m{m}", res["time_us"][f"{name}_m{m}"], res["relerr_bk64"].get(f"{name}_m{m}"), file=sys.stderr, flush=True)
        del ws, pk
    print(json.dumps(res))


if __name__ == "__main__":
    main()
