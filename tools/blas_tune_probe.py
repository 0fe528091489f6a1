"""How much of the library GEMM's gap to the HBM roofline on decode shapes is heuristic algo choice?
Times F.linear (hipBLASLt heuristic default) vs the same call after torch TunableOp has searched the library's
algorithms for the shape. Prints one JSON line: {shape: [default_us, tuned_us, tuned_GBps]}."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

BF16 = torch.bfloat16
SHAPES = {
    "8b": [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 24576, 4096), ("down", 4096, 12288)],
    "32b_tp8": [("qkv", 1280, 5120), ("o", 5120, 1024), ("gate_up", 6400, 5120), ("down", 5120, 3200)],
    "lm_head": [("lm_head", 151936, 1024)],
    "32b": [("qkv", 10240, 5120), ("o", 5120, 8192), ("gate_up", 51200, 5120), ("down", 5120, 25600)],
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


def main():
    models = sys.argv[1:] or ["8b", "32b_tp8", "lm_head"]
    ms = (144,)
    res = {}
    cases = []
    for model in models:
        for name, n, k in SHAPES[model]:
            for m in ms:
                ncopy = max(2, min(12, int(1.5e9 // (n * k * 2))))
                ws = [(torch.randn(n, k, device="cuda") * 0.05).to(BF16) for _ in range(ncopy)]
                x = torch.randn(m, k, device="cuda").to(BF16)
                cases.append((f"{model}_{name}_m{m}", x, ws, n, k))

    def run(x, ws):
        for w in ws:
            F.linear(x, w)

    for key, x, ws, n, k in cases:
        res[key] = [round(timeit(lambda: run(x, ws)) / len(ws), 2)]
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30)      # ms per candidate
    tunable.set_max_tuning_iterations(20)
    tunable.set_filename("/tmp/tunableop.csv")
    for key, x, ws, n, k in cases:
        t0 = time.time()
        F.linear(x, ws[0])                    # triggers the search for this (m, n, k)
        torch.cuda.synchronize()
        tune_s = time.time() - t0
        t = timeit(lambda: run(x, ws)) / len(ws)
        res[key] += [round(t, 2), round(n * k * 2 / t / 1e3, 1), round(tune_s, 1)]
        print(key, res[key], file=sys.stderr, flush=True)
    try:
        res["csv"] = open("/tmp/tunableop.csv").read().splitlines()
    except OSError:
        pass
    print(json.dumps(res))


if __name__ == "__main__":
    main()
