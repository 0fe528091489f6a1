"""Per-call time of the decode-sized add-RMSNorm kernels inside a hipGraph (what the decode step sees): a chain of 64
dependent launches, split-K prologue form (4 fp32 slabs) and plain form; prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
ops.load_library(os.environ.get("NVL_BENCH_LIB"))   # A/B against another build of the library
BF16 = torch.bfloat16
res = {}
for rows in (16, 131, 256):
    for hidden in (1024, 4096, 5120):
        part = torch.randn(4, rows, hidden, device="cuda") * 0.1
        r = torch.randn(rows, hidden, device="cuda").to(BF16)
        w = torch.randn(hidden, device="cuda").to(BF16)
        y = torch.empty(rows, hidden, device="cuda", dtype=BF16)
        x = torch.randn(rows, hidden, device="cuda").to(BF16)

        def chain_split():
            for _ in range(64):
                ops.add_rmsnorm_splitk(part, r, w, 1e-6, out=y)

        def chain_plain():
            for _ in range(64):
                ops.add_rmsnorm(x, r, w, 1e-6, out=y)

        for name, fn in (("splitk4", chain_split), ("plain", chain_plain)):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                g.replay()
            e.record(); torch.cuda.synchronize()
            res[f"{name}_{rows}x{hidden}_us"] = round(s.elapsed_time(e) / 20 / 64 * 1e3, 3)
print(json.dumps(res))
