import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nano_vllm_amd import ops
ops.load_library()
lens = [4096] * 4; hq, hkv = 16, 8; n = sum(lens)
q = torch.randn(n, hq, 128, device="cuda").to(torch.bfloat16); k = torch.randn(n, hkv, 128, device="cuda").to(torch.bfloat16); v = torch.randn(n, hkv, 128, device="cuda").to(torch.bfloat16)
cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
o = torch.empty_like(q)
for _ in range(3):
    ops.attn_prefill_varlen(q, k, v, cu, cu, max(lens), 128 ** -0.5, out=o)
torch.cuda.synchronize()
