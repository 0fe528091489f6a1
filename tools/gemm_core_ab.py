"""The hand-scheduled consumer core of nvl_linear_wide (csrc/gemm_wide_core.inc, NVL_WIDE_CORE=1, the default) against
hipcc's schedule of the same decomposition (NVL_WIDE_CORE=0) and hipBLASLt: time per call, bit-equality of the outputs
between the two schedules (same MFMA order => same bits), and the relative error against an fp32 reference.
usage: python tools/gemm_core_ab.py [m ...]      (default 208 256; SWEEP_SHAPES=a,b limits the shapes)
The switch is read once per process, so each arm runs in a child process of its own."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = {"8b_qkv": (6144, 4096, 0), "8b_o": (4096, 4096, 2), "8b_gate_up": (24576, 4096, 1), "8b_down": (4096, 12288, 2),
          "32b_qkv": (10240, 5120, 0), "32b_o": (5120, 8192, 2), "32b_gate_up": (51200, 5120, 1), "32b_down": (5120, 25600, 2),
          "14b_gate_up": (34816, 5120, 1), "06b_lm_head": (151936, 1024, 0), "32b_tp4_gate_up": (12800, 5120, 1), "32b_tp8_gate_up": (6400, 5120, 1)}


def child(ms):
    import torch
    import torch.nn.functional as F
    from nano_vllm_amd import ops
    from tools.gemm_wide_m256 import timeit
    ops.load_library()
    BF16 = torch.bfloat16
    only = os.environ.get("SWEEP_SHAPES")
    force = os.environ.get("CORE_AB_FORCE", "0") == "1"
    if force:                                    # the decomposition the core exists for, on every shape
        os.environ["NVL_WIDE_NT"], os.environ["NVL_WIDE_NW"] = "2", "3"
    res = {}
    for name, (n, k, mode) in SHAPES.items():
        if only and name not in only.split(","):
            continue
        g = torch.Generator(device="cuda").manual_seed(n + k + mode)
        ncopy = max(2, min(6, int(0.8e9 // (n * k * 2))))
        ws = [(torch.randn(n, k, device="cuda", generator=g) * 0.05).to(BF16) for _ in range(ncopy)]
        pk = [ops.pack_weight_tiles(w) for w in ws]
        for m in ms:
            x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(BF16)
            ops._wide_cache.clear()
            plan = ops.linear_wide_plan(m, n, k, mode)
            if not plan:
                continue
            out = ops.linear_wide(x, pk[0], mode, packed=True)
            torch.cuda.synchronize()
            ref = x.float() @ ws[0].float().t()
            if mode == 0:
                got, want = out.float(), ref.to(BF16).float()
            elif mode == 1:
                got = out.float()
                want = (F.silu(ref[:, : n // 2].to(BF16).float()) * ref[:, n // 2:].to(BF16).float()).to(BF16).float()
            else:
                got, want = out.sum(0), ref
            err = ((got - want).abs().max() / want.abs().max()).item()
            digest = hashlib.sha256(out.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
            scratch = torch.empty(max(plan[1], 16), dtype=torch.uint8, device="cuda")

            def ours():
                for w in pk:
                    ops.linear_wide(x, w, mode, out=out, workspace=scratch, packed=True)
            t = timeit(ours) / len(pk)
            t_blas = None
            if os.environ.get("CORE_AB_BLAS") == "1":
                def blas():
                    for w in ws:
                        y = F.linear(x, w)
                        if mode == 1:
                            ops.silu_mul(y)
                t_blas = round(timeit(blas) / len(ws), 2)
            res[f"{name}_m{m}"] = dict(us=round(t, 2), relerr=round(err, 5), sha=digest, split=plan[0], blas_us=t_blas)
            print(name, m, res[f"{name}_m{m}"], file=sys.stderr, flush=True)
        del ws, pk
    print(json.dumps(res))


ARMS = {"tile4": {"CORE_AB_BLAS": "1"},                                   # the default: four-consumer tile kernel where it applies
        "core": {"NVL_WIDE_TILE4": "0"},                                  # one-wave-per-SIMD kernel, hand-scheduled consumer loop
        "hipcc": {"NVL_WIDE_TILE4": "0", "NVL_WIDE_CORE": "0"}}           # ... hipcc's schedule of it (the round-5 kernel)


def main():
    ms = [a for a in sys.argv[1:] if a.isdigit()] or ["208", "256"]
    arms = {}
    for name, extra in ARMS.items():
        env = dict(os.environ, CORE_AB_CHILD="1", **extra)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), *ms], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        arms[name] = json.loads(line[-1]) if line else {"error": p.stderr[-800:]}
    out = {}
    for key, a in arms["tile4"].items():
        if not isinstance(a, dict):
            continue
        row = dict(tile4_us=a["us"], blas_us=a["blas_us"], relerr=a["relerr"], split=a["split"])
        for other in ("core", "hipcc"):
            b = arms[other].get(key)
            if isinstance(b, dict):
                row[other + "_us"] = b["us"]
                row[other + "_relerr"] = b["relerr"]
        row["core_same_bits_as_hipcc"] = (arms["core"].get(key) or {}).get("sha") == (arms["hipcc"].get(key) or {}).get("sha")
        out[key] = row
    print(json.dumps(dict(arms_error={k: v.get("error") for k, v in arms.items() if "error" in v}, shapes=out)))


if __name__ == "__main__":
    if os.environ.get("CORE_AB_CHILD") == "1":
        child([int(a) for a in sys.argv[1:]])
    else:
        main()
