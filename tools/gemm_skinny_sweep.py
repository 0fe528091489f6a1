"""Sweep the (column tiles per workgroup NT, row groups, K split) decompositions of nvl_linear_decode on the four
Qwen3-0.6B decode projections (tile-packed weights, weights rotated so that they come from HBM), timing what the decode
step pays for each: the GEMM alone for qkv / gate_up, the GEMM + the add-RMSNorm that sums its split-K slabs for o / down.
usage: python tools/gemm_skinny_sweep.py [m ...]        (default 64 131 208)
Prints JSON lines {"shape", "m", "plan": "nt,groups,split" | "rule", "us"}; "rule" = what make_plan picks by itself."""
import itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
ops.load_library()
BF16 = torch.bfloat16
SHAPES = {"qkv": (4096, 1024, 0), "o": (1024, 2048, 2), "gate_up": (6144, 1024, 1), "down": (1024, 3072, 2)}


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ms = [int(a) for a in sys.argv[1:]] or [64, 131, 208]
    for name, (n, k, mode) in SHAPES.items():
        ws = [ops.pack_weight_tiles((torch.randn(n, k, device="cuda") * 0.05).to(BF16)) for _ in range(24)]
        for m in ms:
            x = torch.randn(m, k, device="cuda").to(BF16)
            res = torch.randn(m, n, device="cuda").to(BF16) if mode == 2 else None
            nw = torch.ones(n, device="cuda", dtype=BF16)
            plans = ["rule"] + [f"{nt},{mg},{sp}" for nt, mg, sp in itertools.product((1, 2), (1, 2, 3), (1, 2, 4, 8) if mode == 2 else (0,))]
            for plan in plans:
                os.environ.pop("NVL_SKINNY_PLAN", None)
                if plan != "rule":
                    os.environ["NVL_SKINNY_PLAN"] = plan
                ops._splits_cache.clear()
                splits = ops.linear_decode_splits(m, n, k, mode)
                if not splits:
                    continue
                try:
                    out = ops.linear_decode(x, ws[0], mode, packed=True)
                except ops.NvlError:
                    continue
                y = torch.empty(m, n, device="cuda", dtype=BF16) if mode == 2 else None

                def run():
                    for w in ws:
                        ops.linear_decode(x, w, mode, out=out, packed=True)
                        if mode == 2:
                            ops.add_rmsnorm_splitk(out, res, nw, 1e-6, out=y)
                us = timeit(run) / len(ws)
                print(json.dumps(dict(shape=name, m=m, plan=plan, splits=splits, us=round(us, 2))), flush=True)
        os.environ.pop("NVL_SKINNY_PLAN", None)
        ops._splits_cache.clear()
        del ws


if __name__ == "__main__":
    main()
