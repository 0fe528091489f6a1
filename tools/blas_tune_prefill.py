"""Search hipBLASLt's solutions (torch TunableOp) for the PREFILL-sized library GEMMs of Qwen3-32B at 16,000 rows — BASELINE
config 5's steps are exactly one 16,000-token prompt each — full width (TP = 1) and per rank at TP = 8, and write the result
table the engine loads at start (nano_vllm_amd/tuned/hipblaslt_prefill_gfx950.csv; tuning itself never runs in the engine).
The library's heuristic pick is 20-25 % off its best solution on three of the four projections at this size
(profiles/r04_blas_prefill_probe.json).
usage (GPU box): python tools/blas_tune_prefill.py <out.csv> [rows ...]      (default rows: 16000)"""
import json, os, sys, time
import torch
import torch.nn.functional as F
import torch.cuda.tunable as tunable
BF16 = torch.bfloat16
SHAPES = {"32b_qkv": (10240, 5120), "32b_o": (5120, 8192), "32b_gate_up": (51200, 5120), "32b_down": (5120, 25600),
          "32b_tp8_qkv": (1280, 5120), "32b_tp8_o": (5120, 1024), "32b_tp8_gate_up": (6400, 5120), "32b_tp8_down": (5120, 3200)}


def timeit(fn, iters=5, warm=2):
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


out = sys.argv[1]
rows = [int(a) for a in sys.argv[2:]] or [16000]
res = {}
cases = []
for m in rows:
    for name, (n, k) in SHAPES.items():
        w = (torch.randn(n, k, device="cuda") * 0.05).to(BF16)
        x = torch.randn(m, k, device="cuda").to(BF16)
        cases.append((f"{name}_m{m}", x, w))
        res[f"{name}_m{m}"] = [round(timeit(lambda: F.linear(x, w)), 1)]
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(150)
tunable.set_max_tuning_iterations(8)
tunable.set_filename(out)
for key, x, w in cases:
    t0 = time.time()
    F.linear(x, w)
    torch.cuda.synchronize()
    res[key] += [round(timeit(lambda: F.linear(x, w)), 1), round(time.time() - t0, 1)]
    print(key, res[key], file=sys.stderr, flush=True)
print(json.dumps(res))
