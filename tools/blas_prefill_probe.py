"""Prefill-sized library GEMMs (hipBLASLt through F.linear) on the Qwen3-32B / 8B / 0.6B projections at 16,000 rows: default
heuristic pick vs torch TunableOp's search. Prints one JSON line {shape: [default_us, tuned_us, default TFLOP/s, tuned TFLOP/s]}."""
import json, os, sys, time
import torch
import torch.nn.functional as F
BF16 = torch.bfloat16
SHAPES = {"32b_qkv": (10240, 5120), "32b_o": (5120, 8192), "32b_gate_up": (51200, 5120), "32b_down": (5120, 25600),
          "8b_gate_up": (24576, 4096), "8b_down": (4096, 12288), "06b_gate_up": (6144, 1024), "06b_qkv": (4096, 1024)}


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


m = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
res, cases = {}, []
for name, (n, k) in SHAPES.items():
    w = (torch.randn(n, k, device="cuda") * 0.05).to(BF16)
    x = torch.randn(m, k, device="cuda").to(BF16)
    cases.append((name, x, w, n, k))
    res[name] = [round(timeit(lambda: F.linear(x, w)), 1)]
import torch.cuda.tunable as tunable
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(200)
tunable.set_max_tuning_iterations(10)
tunable.set_filename("/tmp/tunableop_prefill.csv")
for name, x, w, n, k in cases:
    t0 = time.time()
    F.linear(x, w)
    torch.cuda.synchronize()
    ts = time.time() - t0
    t = timeit(lambda: F.linear(x, w))
    fl = 2.0 * m * n * k
    res[name] += [round(t, 1), round(fl / res[name][0] / 1e6, 1), round(fl / t / 1e6, 1), round(ts, 1)]
    print(name, res[name], file=sys.stderr, flush=True)
print(json.dumps(res))
