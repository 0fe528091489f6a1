import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
from oracle import ops as ref
BF16 = torch.bfloat16
ops.load_library()
def case(lens, hq=16, hkv=8, rand_garbage=True):
    bs = 256
    gen = torch.Generator().manual_seed(0)
    nb = [(n + bs - 1)//bs for n in lens]
    total = sum(nb) + 2
    kc = torch.randn(total, bs, hkv, 128, generator=gen).to(BF16)
    vc = torch.randn(total, bs, hkv, 128, generator=gen).to(BF16)
    if not rand_garbage:
        pass
    bt = torch.full((len(lens), 16), -1, dtype=torch.int32)
    c = 0
    for s, n in enumerate(nb):
        for j in range(n):
            bt[s, j] = c; c += 1
    q = torch.randn(len(lens), hq, 128, generator=gen).to(BF16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    scale = 128 ** -0.5
    o_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale).squeeze(1)
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(len(lens), hq, 4096), dtype=torch.uint8, device="cuda")
    o = ops.paged_attn_decode(q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda(), bt.cuda(), ctx.cuda(), scale, 4096, ws)
    err = (o.cpu().float() - o_ref.float()).abs().amax(dim=(1, 2))
    print(lens, "err per seq", [round(float(e), 4) for e in err], "absmax", round(float(o_ref.float().abs().max()), 3), flush=True)
for lens in [[1], [2], [4], [5], [16], [32], [33], [64], [128], [129], [255], [256], [257], [1, 1], [5, 7], [255, 256, 257], [600]]:
    case(lens)
