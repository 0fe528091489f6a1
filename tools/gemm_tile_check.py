"""Correctness of the TILED form of nvl_linear_wide (csrc/gemm_tile.hip) against an fp32 GEMM, in a process of its own:
the form is selected by NVL_WIDE_TILE (read once per process), so the test suite runs this script with NVL_WIDE_TILE=1
(tests/test_kernels_gpu.py::test_linear_wide_tiled_form_in_its_own_process). Prints one JSON line:
{"cases": n, "tiled": how many of them the planner gave to the tiled form, "worst": {...}, "bad": [...]}.
Oracle and bars as in the streaming form's tests: the bf16 output is the fp32 product rounded once (within one bf16
spacing of it), SiLU outputs within 2e-2 x absmax and <= 2 % of elements off by more than 1 ulp, fp32 slabs sum to the
product within 1e-4 x absmax."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nano_vllm_amd import ops
ops.load_library()
BF16 = torch.bfloat16
# (n, k, mode): whole and ragged last workgroups (n not a multiple of 128 / 64 output columns), K splits, few stages
SHAPES = [(6144, 4096, 0), (1280, 5120, 0), (1296, 640, 0), (24576, 4096, 1), (6400, 5120, 1), (1632, 384, 1), (160, 256, 1),
          (4096, 12288, 2), (5120, 3200, 2), (5120, 1024, 2), (272, 512, 2)]
ROWS = [33, 48, 64, 100, 131, 144, 160, 200, 255, 256]


def main():
    bad, worst, n_tiled = [], {}, 0
    for n, k, mode in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(n + k + mode)
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(BF16)
        pk = ops.pack_weight_tiles(w)
        for m in ROWS:
            x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(BF16)
            acc = x.float() @ w.float().t()
            plan = ops.linear_wide_plan(m, n, k, mode)
            if plan is None:
                continue
            ws = torch.empty(max(plan[1], 16), dtype=torch.uint8, device="cuda")
            y = ops.linear_wide(x, pk, mode, workspace=ws, packed=True)
            y_rm = ops.linear_wide(x, w, mode, workspace=ws)          # row-major weights: the streaming form, same split
            tiled = not torch.equal(y, y_rm) or os.environ.get("NVL_WIDE_TILE") == "1"
            n_tiled += int(os.environ.get("NVL_WIDE_TILE") == "1")
            key = f"n{n}_k{k}_mode{mode}_m{m}"
            if mode == 0:
                err = ((y.float() - acc).abs() - acc.abs() * 2.0 ** -8).max().item()
                ok = err <= 1e-4 and y.shape == (m, n)
            elif mode == 1:
                want = (F.silu(acc[:, : n // 2].to(BF16).float()) * acc[:, n // 2:].to(BF16).float()).to(BF16)
                d = (y.float() - want.float()).abs()
                err = (d.max() / want.float().abs().max()).item()
                ulp_off = ((y.view(torch.int16).int() - want.view(torch.int16).int()).abs() > 1).float().mean().item()
                ok = err <= 2e-2 and ulp_off < 0.02 and y.shape == (m, n // 2)
            else:
                assert y.shape == (plan[0], m, n) and y_rm.shape == y.shape
                err = ((y.sum(0) - acc).abs().max() / acc.abs().max()).item()
                ok = err <= 1e-4
            # the two forms agree to the same bars (different fp32 summation orders)
            agree = ((y.float() - y_rm.float()).abs().max() / (acc.abs().max() + 1e-9)).item()
            ok = ok and agree <= (2e-2 if mode != 2 else 1e-4)
            if err > worst.get("err", -1):
                worst = {"case": key, "err": err}
            if not ok:
                bad.append({"case": key, "err": err, "agree": agree, "tiled": tiled})
    print(json.dumps({"cases": len(SHAPES) * len(ROWS), "tiled_env": os.environ.get("NVL_WIDE_TILE"), "worst": worst, "bad": bad[:20]}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
