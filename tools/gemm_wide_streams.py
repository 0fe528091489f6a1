"""Which stream sets the step time of nvl_linear_wide? Times the shipped plans with NVL_WIDE_DBG (csrc/gemm_wide.hip):
0 = normal, 1 = the loader stages only step 0 (no x stream), 2 = every weight load re-reads step 0's lines (no HBM weight
stream), 3 = both (the MFMA / LDS / barrier skeleton); on the hand-scheduled core (256 rows) also + 4 = the loop without
its x fragment reads, + 8 = without its MFMAs. Run once per value (the switch is read once per process); needs a PROBE
build of the library: NVL_PROBES=1 NVL_LIBDIR=$PWD/nano_vllm_amd/lib_probes python -m nano_vllm_amd.build, then the same
two variables on this command.
usage: NVL_WIDE_DBG=k python tools/gemm_wide_streams.py [m ...]     (SWEEP_SHAPES=a,b limits the shapes)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
from tools.gemm_wide_sweep import timeit
ops.load_library()
BF16 = torch.bfloat16
SHAPES = {"8b_gate_up": (24576, 4096, 1), "8b_down": (4096, 12288, 2), "32b_gate_up": (51200, 5120, 1),
          "32b_tp8_gate_up": (6400, 5120, 1), "32b_tp8_qkv": (1280, 5120, 0)}


def main():
    ms = [int(a) for a in sys.argv[1:]] or [16, 144, 256]
    out = {"dbg": os.environ.get("NVL_WIDE_DBG", "0"), "us": {}}
    only = os.environ.get("SWEEP_SHAPES")
    for name, (n, k, mode) in SHAPES.items():
        if only and name not in only.split(","):
            continue
        ncopy = max(2, min(8, int(0.6e9 // (n * k * 2))))
        ws = [ops.pack_weight_tiles((torch.randn(n, k, device="cuda") * 0.05).to(BF16)) for _ in range(ncopy)]
        for m in ms:
            x = torch.randn(m, k, device="cuda").to(BF16)
            plan = ops.linear_wide_plan(m, n, k, mode)
            o = ops.linear_wide(x, ws[0], mode, packed=True)
            scratch = torch.empty(max(plan[1], 16), dtype=torch.uint8, device="cuda")

            def run():
                for w in ws:
                    ops.linear_wide(x, w, mode, out=o, workspace=scratch, packed=True)
            out["us"][f"{name}_m{m}"] = round(timeit(run) / len(ws), 2)
        del ws
    print(json.dumps(out))


if __name__ == "__main__":
    main()
