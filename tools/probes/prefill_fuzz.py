"""Random-shape comparison of nvl_attn_prefill_varlen with the CPU oracle: python tools/probes/prefill_fuzz.py [n] [--paged]
(packed K / V, or — --paged — a paged cache with shuffled block tables and Lq <= Lk: prefix-cache hits / chunk continuations).
Run with NVL_PREFILL_W64=2 to put every case on the generated one-wave-per-SIMD loop. Prints one line per case, fails loudly."""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nano_vllm_amd import ops
from oracle import ops as ref
ops.load_library()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 40
paged = "--paged" in sys.argv
rnd = random.Random(1234)
worst = 0.0
for case in range(n):
    hkv = rnd.choice([1, 2, 4, 8])
    g = rnd.choice([1, 2, 4, 5, 8])
    hq = hkv * g
    nseq = rnd.choice([1, 1, 2, 3, 5, 9])
    lens = [rnd.choice([1, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700, 1000, 1500, 2049, 2304]) for _ in range(nseq)]
    tot = sum(lens)
    gen = torch.Generator().manual_seed(case)
    q = torch.randn(tot, hq, 128, generator=gen).to(torch.bfloat16)
    k = torch.randn(tot, hkv, 128, generator=gen).to(torch.bfloat16)
    v = torch.randn(tot, hkv, 128, generator=gen).to(torch.bfloat16)
    if case % 3 == 0:            # a dominating key late in the first sequence (deferred-rescale branch)
        i = lens[0] - 1
        k[i // 2, 0] = (q[i, 0].float() * rnd.choice([0.5, 3.0])).to(torch.bfloat16)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    scale = 128 ** -0.5
    lse = torch.zeros(tot, hq, dtype=torch.float32, device="cuda")
    if paged:
        bs = rnd.choice([256, 512])
        lks = [l + rnd.choice([0, 1, 64, 255, 256, 700, 2048]) for l in lens]          # cached prefix in front of every prompt
        nb_each = [(x + bs - 1) // bs for x in lks]
        total = sum(nb_each) + 3
        perm = torch.randperm(total, generator=gen).tolist()
        bt = torch.full((nseq, max(nb_each)), -1, dtype=torch.int32)
        kc = torch.randn(total, bs, hkv, 128, generator=gen).to(torch.bfloat16)
        vc = torch.randn(total, bs, hkv, 128, generator=gen).to(torch.bfloat16)
        it = iter(perm)
        for si, nb in enumerate(nb_each):
            for j in range(nb):
                bt[si, j] = next(it)
        cuk = torch.tensor([0] + torch.tensor(lks).cumsum(0).tolist(), dtype=torch.int32)
        o_ref, lse_ref = ref.flash_attn_varlen_func(q, kc, vc, max(lens), cu, max(lks), cuk, scale, True, bt, return_softmax_lse=True)
        o = ops.attn_prefill_varlen(q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda(), cu.cuda(), cuk.cuda(),
                                    max(lens), scale, block_tables=bt.cuda(), lse=lse)
    else:
        o_ref, lse_ref = ref.flash_attn_varlen_func(q, k, v, max(lens), cu, max(lens), cu, scale, True, None, return_softmax_lse=True)
        o = ops.attn_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), cu.cuda(), cu.cuda(), max(lens), scale, lse=lse)
    err = (o.cpu().float() - o_ref.float()).abs().max().item() / (o_ref.float().abs().max().item() + 1e-9)
    lerr = float((lse.cpu() - lse_ref).abs().max())
    worst = max(worst, err)
    print(f"case {case}: hq={hq} hkv={hkv} lens={lens} rel err {err:.4f} lse err {lerr:.5f}", flush=True)
    assert err <= 2e-2 + 1e-3 and lerr <= 1e-2, "MISMATCH"
print("ok, worst rel err", worst)
