import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_amd import ops
from oracle import ops as ref
BF16 = torch.bfloat16
ops.load_library()
def run(n, qzero=False, hq=8, hkv=8):
    bs = 256
    gen = torch.Generator().manual_seed(0)
    kc = torch.randn(2, bs, hkv, 128, generator=gen).to(BF16)
    vc = torch.randn(2, bs, hkv, 128, generator=gen).to(BF16)
    bt = torch.full((1, 16), -1, dtype=torch.int32); bt[0, 0] = 0
    q = torch.randn(1, hq, 128, generator=gen).to(BF16)
    if qzero: q.zero_()
    ctx = torch.tensor([n], dtype=torch.int32)
    scale = 128 ** -0.5
    o_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale).squeeze(1)
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(1, hq, 4096), dtype=torch.uint8, device="cuda")
    o = ops.paged_attn_decode(q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda(), bt.cuda(), ctx.cuda(), scale, 4096, ws).cpu()
    h = 0
    V = vc[0, :n, h if hkv > 1 else 0].float()       # [n, 128]
    K = kc[0, :n, h if hkv > 1 else 0].float()
    w_true = torch.softmax((K @ q[0, h].float()) * scale, 0)
    w_gpu = torch.linalg.lstsq(V.T, o[0, h].float().unsqueeze(1)).solution.squeeze(1)
    print(f"n={n} qzero={qzero} err={float((o.float()-o_ref.float()).abs().max()):.4f}")
    print("  w_true", [round(float(x), 3) for x in w_true[:8]])
    print("  w_gpu ", [round(float(x), 3) for x in w_gpu[:8]], "sum", round(float(w_gpu.sum()), 3))
    wsf = ws.view(torch.float32).cpu()
    max_chunks = 32
    part_o = wsf[: 1*hq*max_chunks*128].view(hq, max_chunks, 128)
    part_ml = wsf[1*hq*max_chunks*128: 1*hq*max_chunks*130].view(hq, max_chunks, 2)
    print("  ml head0 chunk0", part_ml[0, 0].tolist(), " true m(log2)=", float(((K @ q[0, h].float()) * scale * 1.442695).max()))
for n in [2, 3, 5]:
    run(n, False); run(n, True)
