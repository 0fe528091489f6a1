"""Shared-prefix pass of the decode attention (nvl_decode_plan's `shared_prefix`, ABI v5; group ids: v6) against the plain
launch and against the CPU oracle.

The reference's prefix cache (engine/block_manager.py:58-82) gives every request that starts with the same tokens the
SAME leading block ids; its attention call (layers/attention.py:72-74, flash_attn_with_kvcache) still reads those blocks
once per sequence. The pass reads them once per pack of 16 / G sequences and merges its result like any other split:
same value as the plain launch up to the fp32 summation order, so the bars are the attention bars of
tests/test_kernels_gpu.py — |dO| <= 2e-2 * absmax vs the oracle, |dLSE| <= 2e-3 — plus bit-identical K/V caches (the
pass never writes them) and bit-identical outputs whenever the pass has nothing to do.
"""
import pytest
import torch

from oracle import ops as ref

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
BS, MAX_CTX = 256, 4096


@pytest.fixture(scope="module")
def ops():
    from nano_vllm_amd import ops as _ops
    _ops.load_library()
    return _ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def _tables(lens, shared, gen, private=()):
    """Block tables whose first `shared` columns hold the same block ids in every row that is not in `private` (what the
    prefix cache hands out; `private` rows hold their own copies, like the requests prefilled before the prefix was
    registered), the rest private and shuffled; -1 padded to the engine's width (model_runner.py:125)."""
    nb = [(n + BS - 1) // BS for n in lens]
    total = shared + sum(n if i in private else max(n - shared, 0) for i, n in enumerate(nb)) + 3
    perm = torch.randperm(total, generator=gen).tolist()
    common, rest = perm[:shared], iter(perm[shared:])
    bt = torch.full((len(lens), MAX_CTX // BS), -1, dtype=torch.int32)
    for i, n in enumerate(nb):
        for j in range(n):
            bt[i, j] = common[j] if (j < shared and i not in private) else next(rest)
    return bt, total


def _shp(k, lens, private=()):
    """The plan's shared-prefix argument: [k, member flag per row]. Padding rows (context 0) carry a stale flag of 1 on
    purpose — the engine does not clear flags of rows a smaller batch no longer uses."""
    return torch.tensor([k] + [0 if i in private else 1 for i in range(len(lens))], dtype=torch.int32, device="cuda")


def _rope_table():
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128))
    fr = torch.arange(MAX_CTX).float()[:, None] * inv[None]
    return torch.cat([fr.cos(), fr.sin()], -1).contiguous()


# rows: live lengths all beyond the shared blocks, padding rows (context 0) inside and at the end of packs, a row whose
# newest token is the first one behind the shared blocks, 19 rows = not a multiple of any pack size (2, 4, 8); rows 1, 12
# and 13 are NOT members (private copies of their leading blocks; row 1 is even too short to hold two shared blocks):
# a pack with one non-member (rows 0-1 at G = 8), and a pack of non-members only (rows 12-13 at G = 8)
LENS = [513, 300, 0, 1024, 1025, 2048, 515, 0, 0, 640, 4096, 513, 900, 901, 0, 777, 1300, 520, 3000]
PRIVATE = (1, 12, 13)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (8, 1), (16, 2), (64, 8), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("shared", [1, 2])
@pytest.mark.parametrize("kv", ["bf16", "fp8"])
def test_fused_decode_with_a_shared_prefix_pass(ops, hq, hkv, shared, kv):
    """nvl_paged_attn_decode_fused driven by a plan that carries `shared` common blocks vs the same launch with a
    plain plan: K/V caches bit for bit, outputs to 1e-2 * absmax (order of the fp32 merge), and output + LSE against
    the oracle on the cache the kernels left behind; padded rows stay zero. A count of 0 through the same
    pointer: the pass finds nothing to do and the launch is the plain one bit for bit."""
    gen = g(300 + shared)
    b = len(LENS)
    bt, total = _tables(LENS, shared, gen, PRIVATE)
    dt = torch.float8_e4m3fn if kv == "fp8" else BF16
    kc = (torch.randn(total, hkv, BS, 128, generator=gen) * 0.7).to(dt)
    vc = (torch.randn(total, hkv, BS, 128, generator=gen) * 0.7).to(dt)
    qkv = torch.randn(b, (hq + 2 * hkv) * 128, generator=gen).to(BF16).cuda()
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16).cuda()
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16).cuda()
    table = _rope_table().cuda()
    ctx = torch.tensor(LENS, dtype=torch.int32)
    dctx, dbt = ctx.cuda(), bt.cuda()
    scale = 128 ** -0.5
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, MAX_CTX), dtype=torch.uint8, device="cuda")
    assert ops.decode_attention_shares_prefixes(hq, hkv, BS)

    def run(plan):
        k1, v1 = kc.clone().cuda(), vc.clone().cuda()
        lse = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
        o = ops.paged_attn_decode_fused(qkv, qw, kw, 1e-6, table, k1, v1, dbt, dctx, hq, scale, MAX_CTX,
                                        torch.zeros_like(ws), plan=plan, lse=lse)
        torch.cuda.synchronize()
        return o, lse, k1, v1

    o0, lse0, k0, v0 = run(ops.decode_plan(dctx, hq, hkv, MAX_CTX))
    shp = _shp(shared, LENS, PRIVATE)
    plan_px = ops.decode_plan(dctx, hq, hkv, MAX_CTX, shared_prefix=shp, block_size=BS)
    o1, lse1, k1, v1 = run(plan_px)
    assert torch.equal(k1.view(torch.uint8), k0.view(torch.uint8)) and torch.equal(v1.view(torch.uint8), v0.view(torch.uint8))
    live = [i for i, n in enumerate(LENS) if n > 0]
    dead = [i for i, n in enumerate(LENS) if n == 0]
    absmax = float(o0.float().abs().max())
    assert float((o1.float() - o0.float()).abs().max()) <= 1e-2 * absmax
    assert float((lse1[live] - lse0[live]).abs().max()) <= 2e-3
    assert not o1[dead].any() and o1[live].any()
    assert not torch.equal(o1, o0), "the pass did not change a single bit: did it run?"
    # against the oracle, on the cache the kernels left behind (new token included), q rebuilt by the unfused prologue
    if kv == "bf16":
        q1 = torch.empty(b, hq, 128, dtype=BF16, device="cuda")
        pos = (ctx.long() - 1).clamp(min=0).cuda()
        ops.qknorm_rope_kvstore(qkv, pos, qw, kw, 1e-6, table, None, q1, None, None, None, hq, hkv)
        o_ref, lse_ref = ref.flash_attn_with_kvcache(q1.cpu().unsqueeze(1), ref.from_head_major(k1.cpu()),
                                                     ref.from_head_major(v1.cpu()), ctx, bt, scale, return_softmax_lse=True)
        o_ref = o_ref.squeeze(1)
        assert float((o1.cpu().float()[live] - o_ref.float()[live]).abs().max()) <= 2e-2 * float(o_ref.float()[live].abs().max())
        assert float((lse1.cpu()[live] - lse_ref[live]).abs().max()) <= 2e-3
    # the same plan buffer, re-planned with a count of zero: nothing shared this step
    shp[0] = 0
    ops.decode_plan(dctx, hq, hkv, MAX_CTX, plan=plan_px, shared_prefix=shp, block_size=BS)
    o2, lse2, k2, v2 = run(plan_px)
    assert torch.equal(o2, o0) and torch.equal(lse2, lse0) and torch.equal(k2.view(torch.uint8), k0.view(torch.uint8))


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (8, 1), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("splits", [2, 5])
def test_shared_prefix_pass_on_qkv_split_k_slabs(ops, hq, hkv, splits):
    """The deep-K models hand the fused decode attention their qkv projection as fp32 split-K slabs (Qwen3-8B in BASELINE
    config 3 does, at the batch sizes whose GEMM plan splits K): the pass kernel's q prologue sums them with the same
    helper as the stream-K kernel's — output, LSE and both caches are bit-identical to the launch on the reduced bf16
    matrix, with the shared-prefix pass running in both."""
    gen = g(330 + splits)
    b = len(LENS)
    bt, total = _tables(LENS, 2, gen, PRIVATE)
    kc = (torch.randn(total, hkv, BS, 128, generator=gen) * 0.7).to(BF16)
    vc = (torch.randn(total, hkv, BS, 128, generator=gen) * 0.7).to(BF16)
    slabs = (torch.randn(splits, b, (hq + 2 * hkv) * 128, generator=gen) / splits ** 0.5).cuda()
    acc = slabs[0].clone()
    for s_ in range(1, splits):
        acc += slabs[s_]                                   # fp32, slab order: what the slab-reduce launch would write
    qkv = acc.to(BF16)
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16).cuda()
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16).cuda()
    table = _rope_table().cuda()
    dctx, dbt = torch.tensor(LENS, dtype=torch.int32).cuda(), bt.cuda()
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, MAX_CTX), dtype=torch.uint8, device="cuda")
    shp = _shp(2, LENS, PRIVATE)
    outs = []
    for src in (qkv, slabs):
        k1, v1 = kc.clone().cuda(), vc.clone().cuda()
        lse = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
        plan = ops.decode_plan(dctx, hq, hkv, MAX_CTX, shared_prefix=shp, block_size=BS)
        o = ops.paged_attn_decode_fused(src, qw, kw, 1e-6, table, k1, v1, dbt, dctx, hq, 128 ** -0.5, MAX_CTX,
                                        torch.zeros_like(ws), plan=plan, lse=lse)
        torch.cuda.synchronize()
        outs.append((o, lse, k1, v1))
    assert all(torch.equal(a, c) for a, c in zip(outs[0], outs[1]))
    # ... and the pass really ran on the slabs: the plain launch on the same slabs differs in the summation order
    k1, v1 = kc.clone().cuda(), vc.clone().cuda()
    o_plain = ops.paged_attn_decode_fused(slabs, qw, kw, 1e-6, table, k1, v1, dbt, dctx, hq, 128 ** -0.5, MAX_CTX,
                                          torch.zeros_like(ws), plan=ops.decode_plan(dctx, hq, hkv, MAX_CTX))
    assert not torch.equal(o_plain, outs[1][0])
    assert float((o_plain.float() - outs[1][0].float()).abs().max()) <= 1e-2 * float(o_plain.float().abs().max())


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (8, 1), (40, 8), (20, 4), (5, 1), (24, 8)])
def test_unfused_decode_clamps_the_shared_prefix_to_the_shortest_row(ops, hq, hkv):
    """nvl_paged_attn_decode (K/V already stored) with two common blocks claimed while one member ends INSIDE the second
    one (500 tokens: its tail is the common block's content) and one is a single token: the device clamps the pass to
    floor((min member len - 1) / 32) tiles — 0 with the one-token member in the batch (bit-identical to the plain
    launch), 15 tiles = 480 tokens without it or when that row is not a member — and every row still matches the oracle."""
    gen = g(311)
    for lens, private in (([600, 500, 1, 2048, 513], ()), ([600, 500, 0, 2048, 513, 512, 900], ()),
                          ([600, 500, 1, 2048, 513], (2,))):
        b = len(lens)
        bt, total = _tables(lens, 2, gen, private)
        kc = torch.randn(total, BS, hkv, 128, generator=gen).to(BF16)       # token-major (the oracle's layout)
        vc = torch.randn(total, BS, hkv, 128, generator=gen).to(BF16)
        q = torch.randn(b, hq, 128, generator=gen).to(BF16)
        ctx = torch.tensor(lens, dtype=torch.int32)
        scale = 128 ** -0.5
        o_ref, lse_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale, return_softmax_lse=True)
        o_ref = o_ref.squeeze(1)
        dq, dk, dv = q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda()
        dctx, dbt = ctx.cuda(), bt.cuda()
        ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, MAX_CTX), dtype=torch.uint8, device="cuda")
        lse0 = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
        o0 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, MAX_CTX, ws, plan=ops.decode_plan(dctx, hq, hkv, MAX_CTX),
                                   lse=lse0)
        shp = _shp(2, lens, private)
        plan = ops.decode_plan(dctx, hq, hkv, MAX_CTX, shared_prefix=shp, block_size=BS)
        lse1 = torch.zeros_like(lse0)
        o1 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, MAX_CTX, torch.zeros_like(ws), plan=plan, lse=lse1)
        torch.cuda.synchronize()
        live = [i for i, n in enumerate(lens) if n > 0]
        if 1 in lens and not private:
            assert torch.equal(o1, o0) and torch.equal(lse1, lse0)
        else:
            assert not torch.equal(o1, o0)
        assert float((o1.cpu().float()[live] - o_ref.float()[live]).abs().max()) <= 2e-2 * float(o_ref.float().abs().max()) + 1e-3
        assert float((lse1.cpu()[live] - lse_ref[live]).abs().max()) <= 2e-3
        for i, n in enumerate(lens):
            if n == 0:
                assert not o1[i].any()


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (8, 1), (40, 8)])
@pytest.mark.parametrize("slots", [2, 4])
def test_several_shared_prefixes_in_one_step(ops, hq, hkv, slots):
    """Two (three) system prompts in one batch (ABI 6: the member flags are group ids, `prefix_groups` = group slots per
    pack): rows of group 1 share two blocks among themselves, rows of group 2 two OTHER blocks, rows of group 3 a third
    pair; groups are interleaved in row order, so packs hold one, two or three groups, plus non-members and padding rows.
    Every row must match the oracle on ITS OWN block table (output and LSE), and the launch must differ from the plain one
    on the rows of every group (each group's prefix really went through the pass). With one slot only (the ABI 5 form)
    the same flags serve the lowest group id of a pack and nothing else of that pack — which this test shows is NOT
    enough for mixed packs (outputs of surplus groups' rows are then wrong): sizing the slots is the caller's job."""
    gen = g(340 + slots)
    groups = [1, 2, 1, 0, 2, 1, 3, 2, 0, 1, 3, 2, 1, 0, 0, 2, 3, 1, 2, 1, 1, 2, 0, 3]          # 0 = shares nothing
    lens = [513 + 41 * i for i in range(len(groups))]
    lens[8], lens[14] = 0, 0                                                                    # padding rows
    b = len(lens)
    nb = [(n + BS - 1) // BS for n in lens]
    total = 6 + sum(nb) + 3
    perm = torch.randperm(total, generator=gen).tolist()
    common = {1: perm[0:2], 2: perm[2:4], 3: perm[4:6]}
    rest = iter(perm[6:])
    bt = torch.full((b, MAX_CTX // BS), -1, dtype=torch.int32)
    for i, n in enumerate(nb):
        for j in range(n):
            bt[i, j] = common[groups[i]][j] if (groups[i] and j < 2) else next(rest)
    kc = torch.randn(total, BS, hkv, 128, generator=gen).to(BF16)           # token-major (the oracle's layout)
    vc = torch.randn(total, BS, hkv, 128, generator=gen).to(BF16)
    q = torch.randn(b, hq, 128, generator=gen).to(BF16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale, return_softmax_lse=True)
    o_ref = o_ref.squeeze(1)
    dq, dk, dv = q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda()
    dctx, dbt = ctx.cuda(), bt.cuda()
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, MAX_CTX), dtype=torch.uint8, device="cuda")
    o0 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, MAX_CTX, ws, plan=ops.decode_plan(dctx, hq, hkv, MAX_CTX))
    shp = torch.tensor([2] + groups, dtype=torch.int32, device="cuda")
    plan = ops.decode_plan(dctx, hq, hkv, MAX_CTX, shared_prefix=shp, block_size=BS, prefix_groups=slots)
    lse1 = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
    o1 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, MAX_CTX, torch.zeros_like(ws), plan=plan, lse=lse1)
    torch.cuda.synchronize()
    live = [i for i, n in enumerate(lens) if n > 0]
    pack = 16 // (hq // hkv)
    worst_pack = max(len({groups[i] for i in range(p0, min(p0 + pack, b)) if groups[i] and lens[i] > 0}) for p0 in range(0, b, pack))
    err = (o1.cpu().float() - o_ref.float()).abs().amax(dim=(1, 2))
    tol = 2e-2 * float(o_ref.float().abs().max()) + 1e-3
    if slots >= worst_pack:
        assert float(err[live].max()) <= tol
        assert float((lse1.cpu()[live] - lse_ref[live]).abs().max()) <= 2e-3
        for gid in (1, 2, 3):
            rows = [i for i in live if groups[i] == gid]
            assert not torch.equal(o1[rows], o0[rows]), f"group {gid} did not go through the pass"
    else:
        assert float(err[live].max()) > tol           # three groups in a pack, two slots: a group's prefix is missing
    for i, n in enumerate(lens):
        if n == 0:
            assert not o1[i].any()


@pytest.mark.parametrize("hq,hkv", [(8, 2), (10, 2)])
def test_a_full_batch_on_the_divided_grid(ops, hq, hkv):
    """BASELINE config 3's step shape at kernel level: 256 rows, a 512-token prefix shared by all but the first 25 (which
    hold private copies, block_manager.py:110-120), own suffixes of 16 ... 256 tokens. Large enough that (a) the launch's
    workgroups are really DIVIDED between the stream-K grid and the pack workgroups (attn_decode.hip: px_split — the small
    cases above leave the stream-K shares as the plain plan makes them), (b) every pack workgroup walks several (pack, kv
    head) items one after the other. Hq / Hkv = 4 takes the one-launch form, 5 (runtime group size) the pass as its own
    launch. Every live row against the oracle (output and LSE), and against the plain launch."""
    gen = g(77)
    b, private = 256, tuple(range(25))
    lens = [512 + int(x) for x in torch.randint(16, 257, (b,), generator=gen)]
    lens[40], lens[41], lens[255] = 0, 0, 0                                  # graph-padding rows inside and at the end
    bt, total = _tables(lens, 2, gen, private)
    kc = torch.randn(total, BS, hkv, 128, generator=gen).to(BF16)           # token-major (the oracle's layout)
    vc = torch.randn(total, BS, hkv, 128, generator=gen).to(BF16)
    q = torch.randn(b, hq, 128, generator=gen).to(BF16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale, return_softmax_lse=True)
    o_ref = o_ref.squeeze(1)
    dq, dk, dv = q.cuda(), ref.to_head_major(kc).cuda(), ref.to_head_major(vc).cuda()
    dctx, dbt = ctx.cuda(), bt.cuda()
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, MAX_CTX), dtype=torch.uint8, device="cuda")
    o0 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, MAX_CTX, ws, plan=ops.decode_plan(dctx, hq, hkv, MAX_CTX))
    shp = _shp(2, lens, private)              # (the plan keeps a POINTER to the flags: they must outlive its launches, nvl.h)
    plan = ops.decode_plan(dctx, hq, hkv, MAX_CTX, shared_prefix=shp, block_size=BS)
    lse1 = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
    o1 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, MAX_CTX, torch.zeros_like(ws), plan=plan, lse=lse1)
    torch.cuda.synchronize()
    live = [i for i, n in enumerate(lens) if n > 0]
    absmax = float(o_ref.float()[live].abs().max())
    assert float((o1.cpu().float()[live] - o_ref.float()[live]).abs().max()) <= 2e-2 * absmax
    assert float((lse1.cpu()[live] - lse_ref[live]).abs().max()) <= 2e-3
    assert float((o1.float() - o0.float()).abs().max()) <= 1e-2 * float(o0.float().abs().max())
    members = [i for i in live if i not in private]
    assert not torch.equal(o1[members], o0[members]), "the pass did not change a single bit: did it run?"
    assert not o1[[40, 41, 255]].any()


def test_shared_prefix_count_is_read_when_the_captured_plan_replays(ops):
    """The count lives in device memory and the plan kernel reads it when it RUNS: one captured graph (plan + fused
    attention) serves steps with and without a shared prefix — replayed with the count at 2, at 0 and at 2 again it
    reproduces the eager launches of the same inputs bit for bit."""
    hq, hkv = 32, 8
    gen = g(320)
    lens = [513 + 37 * i for i in range(24)]
    b = len(lens)
    bt, total = _tables(lens, 2, gen)
    kc = (torch.randn(total, hkv, BS, 128, generator=gen) * 0.7).to(BF16).cuda()
    vc = (torch.randn(total, hkv, BS, 128, generator=gen) * 0.7).to(BF16).cuda()
    qkv = torch.randn(b, (hq + 2 * hkv) * 128, generator=gen).to(BF16).cuda()
    nw = torch.ones(128, dtype=BF16, device="cuda")
    table = _rope_table().cuda()
    dctx, dbt = torch.tensor(lens, dtype=torch.int32).cuda(), bt.cuda()
    scale = 128 ** -0.5
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, MAX_CTX), dtype=torch.uint8, device="cuda")
    shp = _shp(2, lens)
    plan = torch.zeros(ops.decode_plan_bytes(), dtype=torch.uint8, device="cuda")
    out = torch.empty(b, hq, 128, dtype=BF16, device="cuda")

    def step():
        ops.decode_plan(dctx, hq, hkv, MAX_CTX, plan=plan, shared_prefix=shp, block_size=BS)
        ops.paged_attn_decode_fused(qkv, nw, nw, 1e-6, table, kc, vc, dbt, dctx, hq, scale, MAX_CTX, ws, out=out, plan=plan)

    eager = {}
    for n in (2, 0):
        shp[0] = n
        step()
        torch.cuda.synchronize()
        eager[n] = out.clone()
    assert not torch.equal(eager[2], eager[0])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for n in (2, 0, 2):
        shp[0] = n
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager[n]), n


def test_shared_prefix_needs_the_matrix_core_kernel_and_aligned_blocks(ops):
    """Hq / Hkv = 1 runs on the packed-dot kernel, which knows nothing of a shared pass; a block size that is not a
    multiple of 128 tokens is refused as well — reported through the error channel, nothing launched."""
    ctx = torch.tensor([600, 700], dtype=torch.int32, device="cuda")
    shp = torch.tensor([1, 1, 1], dtype=torch.int32, device="cuda")
    assert not ops.decode_attention_shares_prefixes(8, 8, 256)
    with pytest.raises(ops.NvlError, match="matrix-core"):
        ops.decode_plan(ctx, 8, 8, MAX_CTX, shared_prefix=shp, block_size=256)
    with pytest.raises(ops.NvlError, match="block_size"):
        ops.decode_plan(ctx, 16, 8, MAX_CTX, shared_prefix=shp, block_size=96)
