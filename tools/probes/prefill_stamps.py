"""Cycle stamps of the ping-pong prefill kernel's intervals (probe build, NVL_PREFILL_VAR & 16): mean cycles per loop
iteration of the vector interval (softmax + staging), the wait at its barrier, the matrix interval, the wait at its barrier,
for the two halves of the workgroups (waves 0-3 / 4-7). One 1 x 16384 launch, 16 / 8 heads."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nano_vllm_amd import ops
ops.load_library()
n, hq, hkv = 16384, 16, 8
q = torch.randn(n, hq, 128, device="cuda").to(torch.bfloat16); k = torch.randn(n, hkv, 128, device="cuda").to(torch.bfloat16)
v = torch.randn(n, hkv, 128, device="cuda").to(torch.bfloat16)
cu = torch.tensor([0, n], dtype=torch.int32, device="cuda")
lse = torch.zeros(n, hq, dtype=torch.float32, device="cuda")
for _ in range(2):
    ops.attn_prefill_varlen(q, k, v, cu, cu, n, 128 ** -0.5, lse=lse)
torch.cuda.synchronize()
d = lse.flatten()[: 1024 * 64].view(1024, 8, 8).cpu()
out = {}
for name, sl in (("half0", slice(0, 4)), ("half1", slice(4, 8))):
    x = d[:, sl, :]
    it = x[..., 4].sum().item()
    out[name] = dict(iters=it, sm=round(x[..., 0].sum().item() / it, 1), wait_after_sm=round(x[..., 1].sum().item() / it, 1),
                     m=round(x[..., 2].sum().item() / it, 1), wait_after_m=round(x[..., 3].sum().item() / it, 1))
# per wave index
for w in range(8):
    x = d[:, w, :]; it = x[:, 4].sum().item()
    out[f"w{w}"] = [round(x[:, i].sum().item() / it, 1) for i in range(4)]
print(json.dumps(out))
