"""Sweep the (columns per wave NT, consumer waves NW, K split) plans of nvl_linear_wide on the deep-K decode shapes, with
tile-packed weights, to (re)fit the planner's cost model (csrc/gemm_wide.hip::wide_cost).
usage: python tools/gemm_wide_sweep.py [m ...]     (default 144)
Prints JSON lines {"shape", "m", "nt", "nw", "split", "wgs", "us", "GBps"} and, per shape, the planner's own pick."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
ops.load_library()
BF16 = torch.bfloat16
SHAPES = {"8b_qkv": (6144, 4096, 0), "8b_o": (4096, 4096, 2), "8b_down": (4096, 12288, 2), "8b_gate_up": (24576, 4096, 1),
          "32b_qkv": (10240, 5120, 0), "32b_o": (5120, 8192, 2), "32b_down": (5120, 25600, 2), "32b_gate_up": (51200, 5120, 1),
          "32b_tp8_qkv": (1280, 5120, 0), "32b_tp8_gate_up": (6400, 5120, 1), "32b_tp8_down": (5120, 3200, 2),
          # per-rank shapes of Qwen3-32B at TP = 4 (16 / 2 heads, intermediate 6400); the TP = 8 o_proj (K = 1024) runs on
          # the skinny kernel
          "32b_tp4_qkv": (2560, 5120, 0), "32b_tp4_o": (5120, 2048, 2), "32b_tp4_gate_up": (12800, 5120, 1),
          "32b_tp4_down": (5120, 6400, 2)}


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


def main():  # noqa: C901
    ms = [int(a) for a in sys.argv[1:]] or [144]
    only = os.environ.get("SWEEP_SHAPES")
    for name, (n, k, mode) in SHAPES.items():
        if only and name not in only.split(","):
            continue
        ncopy = max(2, min(8, int(0.6e9 // (n * k * 2))))
        ws = [ops.pack_weight_tiles((torch.randn(n, k, device="cuda") * 0.05).to(BF16)) for _ in range(ncopy)]
        for m in ms:
            x = torch.randn(m, k, device="cuda").to(BF16)
            for key in ("NVL_WIDE_NT", "NVL_WIDE_NW", "NVL_WIDE_SPLIT"):
                os.environ.pop(key, None)
            ops._wide_cache.clear()
            auto = ops.linear_wide_plan(m, n, k, mode)
            ksteps = k // 128
            for nt in (1, 2):
                if mode == 1 and nt == 1:
                    continue
                for nw in (3, 4):
                    cols = nw * (1 if mode == 1 else nt) * 16
                    tiles = -(-(n // 2 if mode == 1 else n) // cols)
                    for split in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
                        if ksteps % split or (split > 1 and ksteps // split < 2):
                            continue
                        wgs = tiles * split * (1 if m <= 144 else 2)
                        if wgs < 96 or wgs > 640:
                            continue
                        os.environ.update(NVL_WIDE_NT=str(nt), NVL_WIDE_NW=str(nw), NVL_WIDE_SPLIT=str(split))
                        ops._wide_cache.clear()
                        plan = ops.linear_wide_plan(m, n, k, mode)
                        if not plan or plan[0] != split:
                            continue
                        out = ops.linear_wide(x, ws[0], mode, packed=True)
                        scratch = torch.empty(max(plan[1], 16), dtype=torch.uint8, device="cuda")

                        def run():
                            for w in ws:
                                ops.linear_wide(x, w, mode, out=out, workspace=scratch, packed=True)
                        us = timeit(run) / len(ws)
                        print(json.dumps(dict(shape=name, m=m, nt=nt, nw=nw, split=split, wgs=wgs, us=round(us, 2),
                                              GBps=round(n * k * 2 / us / 1e3))), flush=True)
            print(json.dumps(dict(shape=name, m=m, planner_split=auto[0] if auto else None)), flush=True)
        del ws


if __name__ == "__main__":
    main()
