import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from nano_vllm_amd import ops
from oracle import ops as ref
import test_shared_prefix_gpu as T
ops.load_library()
hq, hkv = int(sys.argv[1]), int(sys.argv[2]); b = int(sys.argv[3])
gen = T.g(77)
private = tuple(range(25))
lens = [512 + int(x) for x in torch.randint(16, 257, (b,), generator=gen)]
for z in [int(x) for x in os.environ.get('ZERO','').split(',') if x]: lens[z] = 0
bt, total = T._tables(lens, 2, gen, private)
kc = torch.randn(total, T.BS, hkv, 128, generator=gen).to(T.BF16); vc = torch.randn(total, T.BS, hkv, 128, generator=gen).to(T.BF16)
q = torch.randn(b, hq, 128, generator=gen).to(T.BF16)
ctx = torch.tensor(lens, dtype=torch.int32); scale = 128 ** -0.5
o_ref, lse_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale, return_softmax_lse=True); o_ref = o_ref.squeeze(1)
dq, dk, dv = q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda(); dctx, dbt = ctx.cuda(), bt.cuda()
ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, T.MAX_CTX), dtype=torch.uint8, device="cuda")
o0 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, T.MAX_CTX, ws, plan=ops.decode_plan(dctx, hq, hkv, T.MAX_CTX))
shp = T._shp(2, lens, private)
plan = ops.decode_plan(dctx, hq, hkv, T.MAX_CTX, shared_prefix=shp, block_size=T.BS)
o1 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, T.MAX_CTX, torch.zeros_like(ws), plan=plan)
torch.cuda.synchronize()
live = [i for i, n in enumerate(lens) if n > 0]
e0 = (o0.cpu().float() - o_ref.float()).abs().amax(dim=2); e1 = (o1.cpu().float() - o_ref.float()).abs().amax(dim=2)
bad = [(r, h) for r, h in (e1 > 0.02).nonzero().tolist() if lens[r] > 0]
print(f"hq={hq} hkv={hkv} b={b} items/wg={os.environ.get('NVL_PX_ITEMS_PER_WG')}: plain max err {float(e0.max()):.4f}, shared max err {float(e1.max()):.4f}, bad (row, head) count {len(bad)}; rows {sorted({r for r, _ in bad})[:40]} heads {sorted({h for _, h in bad})}")
