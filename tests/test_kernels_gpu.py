"""Parity of every HIP kernel (called through the C ABI, nano_vllm_amd.ops -> libnvl_hip.so)
against the CPU oracle (oracle/ops.py) on identical seeded inputs.

Tolerances: pointwise ops <= 1 bf16 ulp vs the fp32 restatement that rounds where
the compiled reference rounds (rotary and KV store are bit-exact); attention max-abs-diff
<= 2e-2 * absmax (flash tolerance; P is rounded to bf16 before P.V in both) PLUS a checksum on the
softmax log-sum-exp (|dLSE| <= 2e-3: fp32 arithmetic on both sides, only the summation order differs).
"""
import math
import os

import pytest
import torch

from oracle import ops as ref

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from nano_vllm_amd import ops as _ops
    _ops.load_library()
    return _ops


def dev(t):
    return t.cuda()


def g(seed):
    return torch.Generator().manual_seed(seed)


def max_ulp(a, b):
    return int(ref.bf16_ulp_diff(a.cpu(), b.cpu()).max())


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,hidden", [(1, 1024), (131, 1024), (7, 4096), (33, 5120), (5, 2048), (3, 8192), (2, 64)])
def test_rmsnorm(ops, rows, hidden):
    x = (torch.randn(rows, hidden, generator=g(1)) * 3).to(BF16)
    w = (1 + 0.1 * torch.randn(hidden, generator=g(2))).to(BF16)
    y = ops.rmsnorm(dev(x), dev(w), 1e-6)
    assert max_ulp(y, ref.rms_forward(x, w, 1e-6)) <= 1


@pytest.mark.parametrize("n,h,hkv", [(1, 16, 8), (37, 16, 8), (130, 32, 8), (9, 8, 1)])
def test_rmsnorm_head_view(ops, n, h, hkv):
    """q/k-norm over strided views of the qkv GEMM output (models/qwen3.py:78-84)."""
    qkv = torch.randn(n, (h + 2 * hkv) * 128, generator=g(3)).to(BF16)
    w = (1 + 0.1 * torch.randn(128, generator=g(4))).to(BF16)
    q = qkv[:, : h * 128].view(n, h, 128)
    k = qkv[:, h * 128: (h + hkv) * 128].view(n, hkv, 128)
    dq = dev(qkv)
    yq = ops.rmsnorm(dq[:, : h * 128].view(n, h, 128), dev(w), 1e-6)
    yk = ops.rmsnorm(dq[:, h * 128: (h + hkv) * 128].view(n, hkv, 128), dev(w), 1e-6)
    assert max_ulp(yq, ref.rms_forward(q, w, 1e-6)) <= 1
    assert max_ulp(yk, ref.rms_forward(k, w, 1e-6)) <= 1


@pytest.mark.parametrize("rows,hidden", [(1, 1024), (131, 1024), (16, 4096), (9, 5120)])
def test_add_rmsnorm(ops, rows, hidden):
    x = torch.randn(rows, hidden, generator=g(5)).to(BF16)
    r = (torch.randn(rows, hidden, generator=g(6)) * 2).to(BF16)
    w = (1 + 0.1 * torch.randn(hidden, generator=g(7))).to(BF16)
    y_ref, r_ref = ref.add_rms_forward(x, r, w, 1e-6)
    dr = dev(r.clone())
    y = ops.add_rmsnorm(dev(x), dr, dev(w), 1e-6)
    assert torch.equal(dr.cpu(), r_ref)  # residual = bf16(x + r): exact
    assert max_ulp(y, y_ref) <= 1


@pytest.mark.parametrize("rows,inter", [(1, 3072), (131, 3072), (17, 12288), (5, 3200), (3, 8)])
def test_silu_mul(ops, rows, inter):
    x = (torch.randn(rows, 2 * inter, generator=g(8)) * 2).to(BF16)
    y = ops.silu_mul(dev(x))
    assert max_ulp(y, ref.silu_and_mul(x)) <= 1


# ------------------------------------------------------------------------------------------
# skinny decode linears (nvl_linear_decode): oracle = fp32 GEMM rounded where the reference's bf16
# F.linear rounds (layers/linear.py:54-156), then the reference's SiluAndMul / add-RMSNorm on top.
LINEAR_SHAPES = [(4096, 1024), (6144, 1024), (512, 256), (2048, 768), (1280, 512)]     # one K pass per wave: K <= 1024


def _close_to_rounded(y, acc, atol=2e-5):
    """y (bf16) is the fp32 value `acc` rounded to bf16, up to fp32 summation-order noise: within one
    bf16 spacing of acc (2^-8 relative) plus an absolute floor for near-cancelled sums (where an ulp
    count is meaningless)."""
    err = (y.cpu().float() - acc).abs()
    return bool((err <= acc.abs() * 2.0 ** -8 + atol).all())


def _lin_inputs(m, n, k, seed):
    x = (torch.randn(m, k, generator=g(seed)) * 0.5).to(BF16)
    w = (torch.randn(n, k, generator=g(seed + 1)) * 0.05).to(BF16)
    return x, w, x.float() @ w.float().t()


@pytest.mark.parametrize("m", [1, 7, 16, 48, 64, 100, 131, 144, 256, 300])
@pytest.mark.parametrize("n,k", LINEAR_SHAPES)
def test_linear_decode_bf16(ops, m, n, k):
    x, w, acc = _lin_inputs(m, n, k, 20)
    assert ops.linear_decode_splits(m, n, k, ops.LINEAR_BF16) == 1
    y = ops.linear_decode(dev(x), dev(w), ops.LINEAR_BF16)
    assert y.shape == (m, n) and _close_to_rounded(y, acc)
    # tile-packed weights (what the engine streams): other addresses, the same arithmetic in the same order
    assert torch.equal(ops.linear_decode(dev(x), ops.pack_weight_tiles(dev(w)), ops.LINEAR_BF16, packed=True), y)
    assert float((ref.bf16_ulp_diff(y.cpu(), acc.to(BF16)) > 0).float().mean()) < 0.01   # order-of-summation flips only


@pytest.mark.parametrize("m", [1, 16, 131, 144, 256, 300])
@pytest.mark.parametrize("n,k", [(6144, 1024), (512, 256), (1536, 768)])
def test_linear_decode_silu(ops, m, n, k):
    """gate|up projection with SiluAndMul as the epilogue (models/qwen3.py:90-113, activation.py:8-11)."""
    x, w, acc = _lin_inputs(m, n, k, 22)
    y = ops.linear_decode(dev(x), dev(w), ops.LINEAR_SILU)
    assert torch.equal(ops.linear_decode(dev(x), ops.pack_weight_tiles(dev(w)), ops.LINEAR_SILU, packed=True), y)
    want = ref.silu_and_mul(acc.to(BF16))
    # the GEMM output may be 1 ulp off before the activation: allow 2 ulp after it, and check closeness
    assert y.shape == (m, n // 2)
    d = (y.cpu().float() - want.float()).abs()
    assert float(d.max()) <= 2e-2 * float(want.float().abs().max())
    assert float((ref.bf16_ulp_diff(y.cpu(), want) > 1).float().mean()) < 0.02


@pytest.mark.parametrize("m", [1, 16, 64, 100, 131, 144, 208, 256, 300])
@pytest.mark.parametrize("n,k", [(1024, 2048), (1024, 3072), (256, 512), (512, 256)])
def test_linear_decode_partials_into_add_rmsnorm(ops, m, n, k):
    """o_proj / down_proj as fp32 split-K partials + the fused slab-sum/add/RMSNorm consumer."""
    x, w, acc = _lin_inputs(m, n, k, 24)
    splits = ops.linear_decode_splits(m, n, k, ops.LINEAR_PARTIAL)
    assert splits >= 1
    parts = ops.linear_decode(dev(x), dev(w), ops.LINEAR_PARTIAL)
    assert parts.shape == (splits, m, n) and parts.dtype == torch.float32
    assert torch.equal(ops.linear_decode(dev(x), ops.pack_weight_tiles(dev(w)), ops.LINEAR_PARTIAL, packed=True), parts)
    s = parts.sum(0).cpu()
    assert float((s - acc).abs().max()) <= 1e-4 * float(acc.abs().max()) + 1e-5
    r = (torch.randn(m, n, generator=g(26)) * 2).to(BF16)
    wn = (1 + 0.1 * torch.randn(n, generator=g(27))).to(BF16)
    gemm_bf16 = parts.sum(0).to(BF16).cpu()          # the rounding point of the bf16 GEMM it replaces
    y_ref, r_ref = ref.add_rms_forward(gemm_bf16, r, wn, 1e-6)
    dr = dev(r.clone())
    y = ops.add_rmsnorm_splitk(parts, dr, dev(wn), 1e-6)
    assert float((ref.bf16_ulp_diff(dr.cpu(), r_ref) > 0).float().mean()) < 1e-3   # slab-sum order: rare 1-ulp flips
    assert max_ulp(dr, r_ref) <= 1 and max_ulp(y, y_ref) <= 2


@pytest.mark.parametrize("m", [768, 784, 1024, 2048, 4096])
def test_linear_decode_splits_query_and_launch_agree_up_to_4096_rows(ops, m):
    """The ABI contract is 'query nvl_linear_decode_splits first': whatever row count it reports as covered must launch
    (round-4 advisor finding: the three-row-group rule of the o_proj shape left m > 768 with > 16 row tiles per group)."""
    n, k = 1024, 2048
    for mode in (ops.LINEAR_BF16, ops.LINEAR_PARTIAL):
        splits = ops.linear_decode_splits(m, n, k, mode)
        if not splits:
            continue
        x, w, acc = _lin_inputs(m, n, k, 30)
        y = ops.linear_decode(dev(x), dev(w), mode)
        got = y.sum(0).cpu() if mode == ops.LINEAR_PARTIAL else y.cpu().float()
        assert float((got - acc).abs().max()) <= 2.0 ** -7 * float(acc.abs().max()) + 1e-4


def test_linear_decode_unsupported_shapes_are_reported(ops):
    assert ops.linear_decode_splits(16, 4096, 1000, ops.LINEAR_BF16) == 0      # K not a multiple of 256
    assert ops.linear_decode_splits(16, 4096, 128, ops.LINEAR_BF16) == 0       # K < 256
    assert ops.linear_decode_splits(16, 6144, 4096, ops.LINEAR_BF16) == 0      # > 1 K pass per wave: library GEMM
    assert ops.linear_decode_splits(16, 4100, 1024, ops.LINEAR_BF16) == 0      # N not a multiple of 16
    x = torch.zeros(16, 1000, dtype=BF16, device="cuda")
    w = torch.zeros(4096, 1000, dtype=BF16, device="cuda")
    with pytest.raises(ops.NvlError):
        ops.linear_decode(x, w, ops.LINEAR_BF16)


# ------------------------------------------------------------------------------------------
# wide-tile streaming linears (nvl_linear_wide): the deep-K decode projections of Qwen3-8B / 32B (full width and
# per-rank TP shapes), same oracle and rounding points as the skinny kernel's tests above. Shapes cover: whole
# workgroups and ragged last ones (n not a multiple of the workgroup's columns), K splits with a step tail
# (k / 128 / splits not a multiple of the ring depth), one and two row groups (m > 144), the lm_head shape.
WIDE_SHAPES = [(6144, 4096), (1280, 5120), (5120, 1024), (1296, 640), (4096, 12288), (48, 128)]


def _same_product(a, b, absmax_of=None):
    """Two kernels' outputs of the SAME product: bit-identical when both walk K in the same order (row-major vs tile-packed
    weights on one decomposition), else (145-256 rows: the packed layout may take the four-consumer tile kernel, whose
    workgroups cover other column ranges and start their rotated K walk elsewhere) equal up to the fp32 summation order,
    i.e. bf16 results that differ by one rounding flip on a small fraction of the elements."""
    if torch.equal(a, b):
        return True
    a, b = a.cpu(), b.cpu()
    if a.dtype == torch.float32:                     # split-K slabs: compare the sums
        a, b = a.sum(0), b.sum(0)
        return float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-5
    flips = ref.bf16_ulp_diff(a, b)
    tol = 2e-2 * float(a.float().abs().max())
    return float((flips > 0).float().mean()) < 0.04 and float((flips > 1).float().mean()) < 0.004 and \
        float((a.float() - b.float()).abs().max()) <= tol


@pytest.mark.parametrize("m", [1, 16, 131, 144, 200, 300])
@pytest.mark.parametrize("n,k", WIDE_SHAPES)
def test_linear_wide_bf16(ops, m, n, k):
    x, w, acc = _lin_inputs(m, n, k, 30)
    plan = ops.linear_wide_plan(m, n, k, ops.LINEAR_BF16)
    assert plan is not None
    y = ops.linear_wide(dev(x), dev(w), ops.LINEAR_BF16)
    assert y.shape == (m, n) and _close_to_rounded(y, acc, atol=1e-4)
    assert float((ref.bf16_ulp_diff(y.cpu(), acc.to(BF16)) > 0).float().mean()) < 0.02   # order-of-summation flips only
    # the tile-packed weight copy (what the engine streams): other addresses, the same arithmetic in the same order
    assert _same_product(ops.linear_wide(dev(x), ops.pack_weight_tiles(dev(w)), ops.LINEAR_BF16, packed=True), y)


@pytest.mark.parametrize("m", [1, 16, 144, 256])
@pytest.mark.parametrize("n,k", [(24576, 4096), (12800, 5120), (1632, 384)])
def test_linear_wide_silu(ops, m, n, k):
    """gate|up projection with SiluAndMul as the epilogue — in-kernel when K is not split, in the slab-reduce kernel
    when it is (models/qwen3.py:90-113, activation.py:8-11)."""
    x, w, acc = _lin_inputs(m, n, k, 32)
    y = ops.linear_wide(dev(x), dev(w), ops.LINEAR_SILU)
    want = ref.silu_and_mul(acc.to(BF16))
    assert y.shape == (m, n // 2)
    d = (y.cpu().float() - want.float()).abs()
    assert float(d.max()) <= 2e-2 * float(want.float().abs().max())
    assert float((ref.bf16_ulp_diff(y.cpu(), want) > 1).float().mean()) < 0.02
    assert _same_product(ops.linear_wide(dev(x), ops.pack_weight_tiles(dev(w)), ops.LINEAR_SILU, packed=True), y)


def test_pack_weight_tiles_layout(ops):
    """packed[n/16][k/32][64 lanes][8] holds the 16 x 32 sub-matrix of (tile, k-block) in v_mfma_f32_16x16x32_bf16
    A-operand lane order: lane = 16 * (k % 32 // 8) + row % 16."""
    n, k = 96, 256
    w = torch.arange(n * k, dtype=torch.float32).remainder(30011).to(BF16).view(n, k)
    got = ops.pack_weight_tiles(dev(w)).cpu()
    want = w.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(n, k)
    assert torch.equal(got, want)


@pytest.mark.parametrize("m", [1, 16, 131, 144, 256])
@pytest.mark.parametrize("n,k", [(4096, 4096), (4096, 12288), (5120, 3200), (5120, 1024)])
def test_linear_wide_partials_into_add_rmsnorm(ops, m, n, k):
    """o_proj / down_proj as fp32 split-K slabs + the fused slab-sum/add/RMSNorm consumer."""
    x, w, acc = _lin_inputs(m, n, k, 34)
    splits, ws = ops.linear_wide_plan(m, n, k, ops.LINEAR_PARTIAL)
    assert splits >= 1 and ws == 0
    parts = ops.linear_wide(dev(x), dev(w), ops.LINEAR_PARTIAL)
    assert parts.shape == (splits, m, n) and parts.dtype == torch.float32
    assert _same_product(ops.linear_wide(dev(x), ops.pack_weight_tiles(dev(w)), ops.LINEAR_PARTIAL, packed=True), parts)
    s = parts.sum(0).cpu()
    assert float((s - acc).abs().max()) <= 1e-4 * float(acc.abs().max()) + 1e-5
    r = (torch.randn(m, n, generator=g(36)) * 2).to(BF16)
    wn = (1 + 0.1 * torch.randn(n, generator=g(37))).to(BF16)
    seq = parts[0].cpu().clone()                      # slabs summed in split order, like the consumer does
    for i in range(1, splits):
        seq += parts[i].cpu()
    gemm_bf16 = seq.to(BF16)                          # the rounding point of the bf16 GEMM it replaces
    y_ref, r_ref = ref.add_rms_forward(gemm_bf16, r, wn, 1e-6)
    dr = dev(r.clone())
    y = ops.add_rmsnorm_splitk(parts, dr, dev(wn), 1e-6)
    assert max_ulp(dr, r_ref) <= 1 and max_ulp(y, y_ref) <= 2


def test_linear_wide_forced_plans_agree(ops, monkeypatch):
    """Every (columns per wave, waves, K split) decomposition computes the same product (NVL_WIDE_* force the plan)."""
    import importlib
    m, n, k = 131, 2560, 5120
    x, w, acc = _lin_inputs(m, n, k, 38)
    for nt, nw, split in [(1, 3, 1), (1, 4, 5), (2, 3, 2), (2, 4, 8), (2, 4, 20)]:
        monkeypatch.setenv("NVL_WIDE_NT", str(nt))
        monkeypatch.setenv("NVL_WIDE_NW", str(nw))
        monkeypatch.setenv("NVL_WIDE_SPLIT", str(split))
        ops._wide_cache.clear()
        assert ops.linear_wide_plan(m, n, k, ops.LINEAR_BF16)[0] == split
        y = ops.linear_wide(dev(x), dev(w), ops.LINEAR_BF16)
        assert _close_to_rounded(y, acc, atol=1e-4), (nt, nw, split)
    ops._wide_cache.clear()


@pytest.mark.parametrize("ct", [4, 6, 8])
@pytest.mark.parametrize("m", [160, 200, 256])
def test_linear_tile4_every_instantiation(ops, m, ct, monkeypatch):
    """The four-consumer tile kernel of nvl_linear_wide (145-256 rows, tile-packed weights: both operands through LDS, x by
    LDS-DMA, W by the loader waves' register rings; hand-scheduled consumer loop) forced onto every output mode with 4 / 6
    / 8 column tiles per workgroup and 3 / 4 row tiles per wave: against the fp32 product with the reference's rounding
    points, and against the one-wave-per-SIMD kernel on row-major weights (same product, other summation order). Shapes:
    whole and ragged last workgroups, a K walk shorter than the W loaders' ring (6 steps), split K with slabs."""
    monkeypatch.setenv("NVL_WIDE_TILE4", "2")
    monkeypatch.setenv("NVL_WIDE_CT", str(ct))
    monkeypatch.setenv("NVL_WIDE_NT", "2")           # (a decomposition with ONE row group of 12 / 16 row tiles: the plans the
    monkeypatch.setenv("NVL_WIDE_NW", "3")           #  tile kernel takes over)
    for n, k, mode in [(2560, 5120, ops.LINEAR_BF16), (1296, 640, ops.LINEAR_BF16), (1632, 384, ops.LINEAR_SILU),
                       (12800, 5120, ops.LINEAR_SILU), (4096, 4096, ops.LINEAR_PARTIAL), (5120, 3200, ops.LINEAR_PARTIAL)]:
        x, w, acc = _lin_inputs(m, n, k, 40 + ct)
        ops._wide_cache.clear()
        assert ops.linear_wide_plan(m, n, k, mode) is not None
        y = ops.linear_wide(dev(x), ops.pack_weight_tiles(dev(w)), mode, packed=True)
        y_wide = ops.linear_wide(dev(x), dev(w), mode)                       # row-major weights: never the tile kernel
        assert _same_product(y, y_wide), (n, k, mode)
        if mode == ops.LINEAR_BF16:
            assert _close_to_rounded(y, acc, atol=1e-4), (n, k)
        elif mode == ops.LINEAR_SILU:
            want = ref.silu_and_mul(acc.to(BF16))
            assert float((y.cpu().float() - want.float()).abs().max()) <= 2e-2 * float(want.float().abs().max()), (n, k)
            assert float((ref.bf16_ulp_diff(y.cpu(), want) > 1).float().mean()) < 0.02
        else:
            assert float((y.sum(0).cpu() - acc).abs().max()) <= 1e-4 * float(acc.abs().max()) + 1e-5, (n, k)
    ops._wide_cache.clear()


def test_linear_wide_unsupported_shapes_are_reported(ops):
    assert ops.linear_wide_plan(16, 4096, 1000, ops.LINEAR_BF16) is None       # K not a multiple of 128
    assert ops.linear_wide_plan(16, 4100, 1024, ops.LINEAR_BF16) is None       # N not a multiple of 16
    assert ops.linear_wide_plan(16, 4112, 1024, ops.LINEAR_SILU) is None       # gate|up: N not a multiple of 32
    x = torch.zeros(16, 1000, dtype=BF16, device="cuda")
    w = torch.zeros(4096, 1000, dtype=BF16, device="cuda")
    with pytest.raises(ops.NvlError):
        ops.linear_wide(x, w, ops.LINEAR_BF16)


# ------------------------------------------------------------------------------------------
def test_rope_bit_exact(ops):
    n, h, hkv = 77, 16, 8
    table = ref.rope_table(128, 4096, 1e6)
    pos = torch.randint(0, 4096, (n,), generator=g(9))
    q = torch.randn(n, h, 128, generator=g(10)).to(BF16)
    k = torch.randn(n, hkv, 128, generator=g(11)).to(BF16)
    q_ref, k_ref = ref.rotary_forward(pos, q, k, table)
    dt, dp = dev(table), dev(pos)
    assert torch.equal(ops.rope_neox(dp, dt, dev(q)).cpu(), q_ref)
    assert torch.equal(ops.rope_neox(dp, dt, dev(k)).cpu(), k_ref)


def test_store_kvcache_bit_exact(ops):
    n, hkv, nblk, bs = 300, 8, 6, 256
    k = torch.randn(n, hkv, 128, generator=g(12)).to(BF16)
    qkv = torch.randn(n, 4096, generator=g(13)).to(BF16)
    v = qkv[:, 3072:].view(n, hkv, 128)  # strided view, as in the reference
    slots = torch.randperm(nblk * bs, generator=g(14))[:n].to(torch.int32)
    slots[::7] = -1
    kc_ref = torch.zeros(nblk, bs, hkv, 128, dtype=BF16)
    vc_ref = torch.zeros_like(kc_ref)
    ref.store_kvcache(k, v, kc_ref, vc_ref, slots)
    kc = torch.zeros(nblk, hkv, bs, 128, dtype=BF16, device="cuda")
    vc = torch.zeros_like(kc)
    dqkv = dev(qkv)
    ops.store_kvcache(dev(k), dqkv[:, 3072:].view(n, hkv, 128), kc, vc, dev(slots))
    assert torch.equal(ref.from_head_major(kc.cpu()), kc_ref)
    assert torch.equal(ref.from_head_major(vc.cpu()), vc_ref)


@pytest.mark.parametrize("n,h,hkv", [(1, 16, 8), (131, 16, 8), (40, 32, 8), (19, 8, 1)])
def test_fused_qknorm_rope_kvstore(ops, n, h, hkv):
    nblk, bs = 4, 256
    qkv = torch.randn(n, (h + 2 * hkv) * 128, generator=g(15)).to(BF16)
    qw = (1 + 0.1 * torch.randn(128, generator=g(16))).to(BF16)
    kw = (1 + 0.1 * torch.randn(128, generator=g(17))).to(BF16)
    table = ref.rope_table(128, 2048, 1e6)
    pos = torch.randint(0, 2048, (n,), generator=g(18))
    slots = torch.randperm(nblk * bs, generator=g(19))[:n].to(torch.int32)
    if n > 3:
        slots[2] = -1
    # oracle: the reference's four separate steps (models/qwen3.py:78-85, attention.py:63)
    q, k, v = qkv.split([h * 128, hkv * 128, hkv * 128], dim=-1)
    q = ref.rms_forward(q.reshape(n, h, 128), qw, 1e-6)
    k = ref.rms_forward(k.reshape(n, hkv, 128), kw, 1e-6)
    q_ref, k_ref = ref.rotary_forward(pos, q, k, table)
    kc_ref = torch.zeros(nblk, bs, hkv, 128, dtype=BF16)
    vc_ref = torch.zeros_like(kc_ref)
    ref.store_kvcache(k_ref, v.reshape(n, hkv, 128), kc_ref, vc_ref, slots)

    q_out = torch.empty(n, h, 128, dtype=BF16, device="cuda")
    k_out = torch.empty(n, hkv, 128, dtype=BF16, device="cuda")
    kc = torch.zeros(nblk, hkv, bs, 128, dtype=BF16, device="cuda")
    vc = torch.zeros_like(kc)
    ops.qknorm_rope_kvstore(dev(qkv), dev(pos), dev(qw), dev(kw), 1e-6, dev(table), dev(slots), q_out, k_out, kc, vc,
                            h, hkv)
    # norm is <=1 ulp vs the oracle; the rotation of a 1-ulp-different input can move the
    # output by a couple of ulps, so compare with a small absolute tolerance as well
    assert (q_out.cpu().float() - q_ref.float()).abs().max() <= 2 ** -5
    assert (k_out.cpu().float() - k_ref.float()).abs().max() <= 2 ** -5
    assert torch.equal(ref.from_head_major(vc.cpu()), vc_ref)
    # the fused kernel must equal the unfused HIP kernels bit for bit
    dq = dev(qkv)
    qn = ops.rmsnorm(dq[:, : h * 128].view(n, h, 128), dev(qw), 1e-6)
    kn = ops.rmsnorm(dq[:, h * 128: (h + hkv) * 128].view(n, hkv, 128), dev(kw), 1e-6)
    q2 = ops.rope_neox(dev(pos), dev(table), qn)
    k2 = ops.rope_neox(dev(pos), dev(table), kn)
    assert torch.equal(q_out, q2) and torch.equal(k_out, k2)
    kc2 = torch.zeros_like(kc)
    vc2 = torch.zeros_like(vc)
    ops.store_kvcache(k2, dq[:, (h + hkv) * 128:].view(n, hkv, 128), kc2, vc2, dev(slots))
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)


# ------------------------------------------------------------------------------------------
def _paged_setup(lens, hkv, bs, seed, extra_blocks=3):
    """Random K/V for each sequence scattered into a paged cache (reference layout) with a
    shuffled block table, -1 padded like engine/model_runner.py:125."""
    gen = g(seed)
    nb_each = [(n + bs - 1) // bs for n in lens]
    total = sum(nb_each) + extra_blocks
    perm = torch.randperm(total, generator=gen).tolist()
    width = max(max(nb_each), 1)
    bt = torch.full((len(lens), width), -1, dtype=torch.int32)
    kc = torch.randn(total, bs, hkv, 128, generator=gen).to(BF16)  # garbage beyond the context
    vc = torch.randn(total, bs, hkv, 128, generator=gen).to(BF16)
    it = iter(perm)
    for s, nb in enumerate(nb_each):
        for j in range(nb):
            bt[s, j] = next(it)
    return kc, vc, bt


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (32, 8), (8, 1), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("lens", [[1], [255, 256, 257], [1, 100, 1023, 1024, 1025, 2048, 17], [4096, 3, 0, 700]])
def test_paged_attn_decode(ops, hq, hkv, lens):
    bs = 256
    kc, vc, bt = _paged_setup(lens, hkv, bs, seed=20 + len(lens))
    b = len(lens)
    q = torch.randn(b, hq, 128, generator=g(21)).to(BF16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    scale = 128 ** -0.5
    o_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale).squeeze(1)
    max_ctx = 4096
    btw = torch.full((b, max_ctx // bs), -1, dtype=torch.int32)
    btw[:, : bt.shape[1]] = bt
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, max_ctx), dtype=torch.uint8, device="cuda")
    o = ops.paged_attn_decode(dev(q), dev(ref.to_head_major(kc)), dev(ref.to_head_major(vc)), dev(btw), dev(ctx), scale,
                              max_ctx, ws)
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err
    for i, n in enumerate(lens):
        if n == 0:
            assert torch.count_nonzero(o[i]) == 0  # padded rows produce zeros


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (32, 8), (8, 1), (16, 2), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("lens", [[1], [255, 256, 257], [1, 100, 1023, 1024, 1025, 2048, 17, 0, 0, 640], [4096, 3, 0, 700],
                                  [0, 0, 5, 0, 0, 0, 900, 0]])
def test_paged_attn_decode_lse_checksum_and_per_step_plan(ops, hq, hkv, lens):
    """(a) The softmax log-sum-exp the kernel reports (flash-attn's softmax_lse) against the oracle's: a checksum on
    the normaliser that the 2e-2 * absmax output tolerance cannot see — |dLSE| <= 2e-3 (fp32 arithmetic on exact
    bf16 products; only the summation order differs). (b) The same launch driven by a per-step plan (nvl_decode_plan)
    must reproduce the unplanned launch BIT FOR BIT (same grid, same shares, same split partials), also with padded
    rows (context_len 0) in the middle of the batch."""
    bs = 256
    kc, vc, bt = _paged_setup(lens, hkv, bs, seed=120 + len(lens))
    b = len(lens)
    q = torch.randn(b, hq, 128, generator=g(121)).to(BF16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale, return_softmax_lse=True)
    max_ctx = 4096
    btw = torch.full((b, max_ctx // bs), -1, dtype=torch.int32)
    btw[:, : bt.shape[1]] = bt
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, max_ctx), dtype=torch.uint8, device="cuda")
    dq, dk, dv, dbt, dctx = dev(q), dev(ref.to_head_major(kc)), dev(ref.to_head_major(vc)), dev(btw), dev(ctx)
    lse = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
    o = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, max_ctx, ws, lse=lse)
    live = torch.tensor([n > 0 for n in lens])
    assert torch.isinf(lse.cpu()[~live]).all() and (lse.cpu()[~live] < 0).all()
    err = (lse.cpu()[live] - lse_ref[live]).abs().max().item()
    assert err <= 2e-3, err
    plan = ops.decode_plan(dctx, hq, hkv, max_ctx)
    lse2 = torch.zeros_like(lse)
    ws2 = torch.zeros_like(ws)
    o2 = ops.paged_attn_decode(dq, dk, dv, dbt, dctx, scale, max_ctx, ws2, plan=plan, lse=lse2)
    if hq // hkv > 1:        # the matrix-core kernel consumes the plan; group size 1 ignores it
        assert torch.equal(o2, o) and torch.equal(lse2, lse)
    err = (o2.cpu().float() - o_ref.squeeze(1).float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (32, 8), (8, 1), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("lens", [[1], [255, 256, 257, 33], [1, 100, 1023, 1024, 1025, 2048, 17, 0, 640], [4096, 3, 0, 700]])
@pytest.mark.parametrize("with_norm", [True, False])
def test_paged_attn_decode_fused_equals_unfused(ops, hq, hkv, lens, with_norm):
    """nvl_paged_attn_decode_fused vs nvl_qknorm_rope_kvstore + nvl_paged_attn_decode with positions /
    slots derived from context_lens / block_tables as the runner builds them
    (engine/model_runner.py:172-188): both caches bit for bit; outputs to flash tolerance (the fused
    kernel feeds the new token to the online softmax first instead of last: same math, different
    fp32 summation order)."""
    bs, max_ctx = 256, 4096
    gen = g(40)
    b = len(lens)
    nb = [(n + bs - 1) // bs for n in lens]
    total = sum(nb) + 3
    kc = torch.randn(total, hkv, bs, 128, generator=gen).to(BF16)
    vc = torch.randn(total, hkv, bs, 128, generator=gen).to(BF16)
    perm = torch.randperm(total, generator=gen).tolist()
    bt = torch.full((b, max_ctx // bs), -1, dtype=torch.int32)
    c = 0
    for s_, n in enumerate(nb):
        for j in range(n):
            bt[s_, j] = perm[c]
            c += 1
    qkv = torch.randn(b, (hq + 2 * hkv) * 128, generator=gen).to(BF16)
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16) if with_norm else None
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16) if with_norm else None
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128))
    fr = torch.arange(max_ctx).float()[:, None] * inv[None]
    table = torch.cat([fr.cos(), fr.sin()], -1).contiguous()
    ctx = torch.tensor(lens, dtype=torch.int32)
    pos = (ctx.long() - 1).clamp(min=0)
    slots = torch.tensor([int(bt[i, (n - 1) // bs]) * bs + (n - 1) % bs if n > 0 else -1 for i, n in enumerate(lens)],
                         dtype=torch.int32)
    scale = 128 ** -0.5
    d = dict(qkv=dev(qkv), qw=dev(qw) if with_norm else None, kw=dev(kw) if with_norm else None, table=dev(table),
             bt=dev(bt), ctx=dev(ctx))
    # unfused
    kc1, vc1 = dev(kc.clone()), dev(vc.clone())
    q1 = torch.empty(b, hq, 128, dtype=BF16, device="cuda")
    ops.qknorm_rope_kvstore(d["qkv"], dev(pos), d["qw"], d["kw"], 1e-6, d["table"], dev(slots), q1, None, kc1, vc1, hq, hkv)
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, max_ctx), dtype=torch.uint8, device="cuda")
    o1 = ops.paged_attn_decode(q1, kc1, vc1, d["bt"], d["ctx"], scale, max_ctx, ws)
    # fused
    kc2, vc2 = dev(kc.clone()), dev(vc.clone())
    ws2 = torch.zeros_like(ws)
    o2 = ops.paged_attn_decode_fused(d["qkv"], d["qw"], d["kw"], 1e-6, d["table"], kc2, vc2, d["bt"], d["ctx"], hq, scale,
                                     max_ctx, ws2)
    torch.cuda.synchronize()
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    err = float((o1.float() - o2.float()).abs().max())
    assert err <= 1e-2 * max(1e-6, float(o1.float().abs().max()))
    # and against the CPU oracle directly (token-major caches, flash_attn_with_kvcache semantics)
    live = [i for i, n in enumerate(lens) if n > 0]
    if live:
        o_ref, lse_ref = ref.flash_attn_with_kvcache(q1.cpu().unsqueeze(1), ref.from_head_major(kc1.cpu()),
                                                     ref.from_head_major(vc1.cpu()), ctx, bt, scale, return_softmax_lse=True)
        o_ref = o_ref.squeeze(1)
        dd = (o2.cpu().float()[live] - o_ref.float()[live]).abs().max()
        assert float(dd) <= 2e-2 * float(o_ref.float()[live].abs().max())
    for i, n in enumerate(lens):
        if n == 0:
            assert not o2[i].any()
    # the fused launch driven by the per-step plan, reporting its softmax LSE: same bits as the unplanned launch
    # (matrix-core kernel), caches again identical, LSE (new token included) against the oracle's
    kc3, vc3 = dev(kc.clone()), dev(vc.clone())
    lse = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
    plan = ops.decode_plan(d["ctx"], hq, hkv, max_ctx)
    o3 = ops.paged_attn_decode_fused(d["qkv"], d["qw"], d["kw"], 1e-6, d["table"], kc3, vc3, d["bt"], d["ctx"], hq, scale,
                                     max_ctx, torch.zeros_like(ws), plan=plan, lse=lse)
    torch.cuda.synchronize()
    assert torch.equal(kc3, kc2) and torch.equal(vc3, vc2)
    if hq // hkv > 1:
        assert torch.equal(o3, o2)
    if live:
        assert float((lse.cpu()[live] - lse_ref[live]).abs().max()) <= 2e-3


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (8, 1), (16, 2), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("splits", [1, 2, 5, 8])
@pytest.mark.parametrize("kv", ["bf16", "fp8"])
def test_paged_attn_decode_fused_sums_qkv_split_k_slabs(ops, hq, hkv, splits, kv):
    """Round 5: the fused decode attention takes the qkv projection as the fp32 split-K slabs of nvl_linear_wide mode 2
    ([S, B, (Hq + 2 Hkv) * 128]) and sums + rounds them in its prologue — what the slab-reduce launch between the two
    kernels used to do (reference: the bf16 F.linear output of QKVParallelLinear, layers/linear.py:96-128, feeding
    qwen3.py:83-85 + attention.py:63,72-74). Same sum order (slab 0, 1, ...) and one rounding => output, K cache and V
    cache are bit-identical to the launch on the reduced bf16 matrix, with and without a plan; padded rows stay zero."""
    bs, max_ctx = 256, 2048
    gen = g(44 + splits)
    lens = [1, 255, 256, 257, 0, 1024, 1500, 33, 2048]
    b = len(lens)
    nb = [(n + bs - 1) // bs for n in lens]
    total = sum(nb) + 2
    dt = torch.float8_e4m3fn if kv == "fp8" else BF16
    kc = (torch.randn(total, hkv, bs, 128, generator=gen) * 0.5).to(dt)
    vc = (torch.randn(total, hkv, bs, 128, generator=gen) * 0.5).to(dt)
    perm = torch.randperm(total, generator=gen).tolist()
    bt = torch.full((b, max_ctx // bs), -1, dtype=torch.int32)
    c = 0
    for s_, n in enumerate(nb):
        for j in range(n):
            bt[s_, j] = perm[c]
            c += 1
    width = (hq + 2 * hkv) * 128
    slabs = dev(torch.randn(splits, b, width, generator=gen) / splits ** 0.5)
    acc = slabs[0].clone()
    for s_ in range(1, splits):
        acc += slabs[s_]                               # fp32, slab order: slab_reduce_kernel's order
    qkv = acc.to(BF16)
    qw = dev((1 + 0.1 * torch.randn(128, generator=gen)).to(BF16))
    kw = dev((1 + 0.1 * torch.randn(128, generator=gen)).to(BF16))
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128))
    table = dev(torch.cat([(torch.arange(max_ctx).float()[:, None] * inv[None]).cos(),
                           (torch.arange(max_ctx).float()[:, None] * inv[None]).sin()], -1).contiguous())
    ctx, btd = dev(torch.tensor(lens, dtype=torch.int32)), dev(bt)
    scale = 128 ** -0.5
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, max_ctx), dtype=torch.uint8, device="cuda")
    assert ops.decode_attention_takes_qkv_slabs(hq, hkv)
    outs = []
    for src in (qkv, slabs):
        for planned in (False, True):
            k1, v1 = dev(kc.clone()), dev(vc.clone())
            plan = ops.decode_plan(ctx, hq, hkv, max_ctx) if planned else None
            o = ops.paged_attn_decode_fused(src, qw, kw, 1e-6, table, k1, v1, btd, ctx, hq, scale, max_ctx,
                                            torch.zeros_like(ws), plan=plan)
            torch.cuda.synchronize()
            outs.append((o, k1.view(torch.uint8), v1.view(torch.uint8)))
    for o, k1, v1 in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(k1, outs[0][1]) and torch.equal(v1, outs[0][2])
    assert not outs[2][0][4].any() and outs[2][0][0].any()          # the padded row stays zero, live rows are written


def test_paged_attn_decode_fused_refuses_slabs_for_the_packed_dot_kernel(ops):
    """Hq / Hkv = 1 runs on the packed-dot kernel, whose prologue reads bf16 only: slabs are refused, not misread."""
    assert not ops.decode_attention_takes_qkv_slabs(8, 8)
    slabs = torch.zeros(2, 1, 24 * 128, device="cuda")
    kc = torch.zeros(2, 8, 256, 128, dtype=BF16, device="cuda")
    table = torch.zeros(256, 128, device="cuda")
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(1, 8, 256), dtype=torch.uint8, device="cuda")
    with pytest.raises(ops.NvlError):
        ops.paged_attn_decode_fused(slabs, None, None, 1e-6, table, kc, kc.clone(), torch.zeros(1, 1, dtype=torch.int32, device="cuda"),
                                    torch.ones(1, dtype=torch.int32, device="cuda"), 8, 0.1, 256, ws)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (16, 2), (8, 1), (32, 8), (40, 8), pytest.param(20, 4, marks=pytest.mark.slow),
                                    pytest.param(5, 1, marks=pytest.mark.slow), pytest.param(24, 8, marks=pytest.mark.slow)])
def test_paged_attn_decode_large_batch(ops, hq, hkv):
    """bench-like shape: batch 256, contexts 100..2048; group size 2 (Qwen3-0.6B), 8 with two / one kv heads
    (Qwen3-32B per-rank shapes at TP = 4 / 8: the matrix-core decode kernel), 4 (Qwen3-8B)."""
    gen = g(30)
    lens = torch.randint(100, 2049, (256,), generator=gen).tolist()
    bs = 256
    kc, vc, bt = _paged_setup(lens, hkv, bs, seed=31)
    q = torch.randn(256, hq, 128, generator=gen).to(BF16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    scale = 128 ** -0.5
    o_ref = ref.flash_attn_with_kvcache(q.unsqueeze(1), kc, vc, ctx, bt, scale).squeeze(1)
    max_ctx = 4096
    btw = torch.full((256, max_ctx // bs), -1, dtype=torch.int32)
    btw[:, : bt.shape[1]] = bt
    planned = ops.decode_plan(dev(ctx), hq, hkv, max_ctx)
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(256, hq, max_ctx), dtype=torch.uint8, device="cuda")
    o = ops.paged_attn_decode(dev(q), dev(ref.to_head_major(kc)), dev(ref.to_head_major(vc)), dev(btw), dev(ctx), scale,
                              max_ctx, ws)
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err
    o_planned = ops.paged_attn_decode(dev(q), dev(ref.to_head_major(kc)), dev(ref.to_head_major(vc)), dev(btw), dev(ctx),
                                      scale, max_ctx, torch.zeros_like(ws), plan=planned)
    assert torch.equal(o_planned, o)


# ------------------------------------------------------------------------------------------
def _cu(lens):
    return torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (8, 1), (32, 8), (16, 2), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("lens", [[1], [128], [129, 64, 300], [5, 1000, 33, 257]])
def test_prefill_contiguous(ops, hq, hkv, lens):
    n = sum(lens)
    gen = g(40)
    q = torch.randn(n, hq, 128, generator=gen).to(BF16)
    qkv = torch.randn(n, 3 * hkv * 128, generator=gen).to(BF16)
    k = qkv[:, hkv * 128: 2 * hkv * 128].view(n, hkv, 128)     # token-strided views
    v = qkv[:, 2 * hkv * 128:].view(n, hkv, 128)
    cu = _cu(lens)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_varlen_func(q, k, v, max(lens), cu, max(lens), cu, scale, True, None,
                                                return_softmax_lse=True)
    dqkv = dev(qkv)
    lse = torch.zeros(n, hq, dtype=torch.float32, device="cuda")
    o = ops.attn_prefill_varlen(dev(q), dqkv[:, hkv * 128: 2 * hkv * 128].view(n, hkv, 128),
                                dqkv[:, 2 * hkv * 128:].view(n, hkv, 128), dev(cu), dev(cu), max(lens), scale, lse=lse)
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err
    # checksum on the softmax normaliser (flash-attn's softmax_lse), which the output tolerance cannot see
    assert float((lse.cpu() - lse_ref).abs().max()) <= 2e-3


def test_prefill_many_short_sequences(ops):
    """600 sequences of 1-70 tokens in one launch: the on-device (sequence, q-block) lookup runs its prefix scan
    in several 256-wide rounds, most tiles are partial, and the last sequence ends exactly at the end of the K/V
    tensors (the buffer descriptors' range check, not a clamp, keeps the partial tiles in bounds)."""
    gen = g(44)
    lens = torch.randint(1, 71, (600,), generator=gen).tolist()
    n, hq, hkv = sum(lens), 4, 2
    q = torch.randn(n, hq, 128, generator=gen).to(BF16)
    k = torch.randn(n, hkv, 128, generator=gen).to(BF16)
    v = torch.randn(n, hkv, 128, generator=gen).to(BF16)
    cu = _cu(lens)
    scale = 128 ** -0.5
    o_ref = ref.flash_attn_varlen_func(q, k, v, max(lens), cu, max(lens), cu, scale, True, None)
    o = ops.attn_prefill_varlen(dev(q), dev(k), dev(v), dev(cu), dev(cu), max(lens), scale)
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err


@pytest.mark.parametrize("nseq", [1, 63, 64, 65])
def test_prefill_tile_list_in_registers_and_in_lds(ops, nseq):
    """Up to 64 sequences a wave derives the (sequence, q-block) of its workgroup in registers (shuffle scan + ballot +
    readlane), beyond that through the LDS prefix: both sides of the switch, ragged lengths whose last q-block leaves
    one, two or three of the four waves without rows (those skip the MFMA work), outputs AND softmax LSE vs the oracle."""
    gen = g(4400 + nseq)
    lens = torch.randint(1, 300, (nseq,), generator=gen).tolist()
    lens[0] = 129          # 1 valid row in the second q-block: three idle waves
    n, hq, hkv = sum(lens), 4, 2
    q = torch.randn(n, hq, 128, generator=gen).to(BF16)
    k = torch.randn(n, hkv, 128, generator=gen).to(BF16)
    v = torch.randn(n, hkv, 128, generator=gen).to(BF16)
    cu = _cu(lens)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_varlen_func(q, k, v, max(lens), cu, max(lens), cu, scale, True, None,
                                                return_softmax_lse=True)
    lse = torch.empty(n, hq, dtype=torch.float32, device="cuda")
    o = ops.attn_prefill_varlen(dev(q), dev(k), dev(v), dev(cu), dev(cu), max(lens), scale, lse=lse)
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err
    assert float((lse.cpu() - lse_ref).abs().max()) <= 2e-3


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 1), (32, 8), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("lq_lk", [[(1, 257)], [(100, 356), (256, 256), (7, 1031)], [(300, 812), (64, 64)]])
def test_prefill_paged_prefix(ops, hq, hkv, lq_lk):
    """Prefix-cache / chunked-prefill path: Lq < Lk, K/V from the paged cache, mask bottom-right."""
    bs = 256
    lqs = [a for a, _ in lq_lk]
    lks = [b for _, b in lq_lk]
    kc, vc, bt = _paged_setup(lks, hkv, bs, seed=50)
    gen = g(51)
    q = torch.randn(sum(lqs), hq, 128, generator=gen).to(BF16)
    cuq, cuk = _cu(lqs), _cu(lks)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_varlen_func(q, kc, vc, max(lqs), cuq, max(lks), cuk, scale, True, bt,
                                                return_softmax_lse=True)
    lse = torch.zeros(sum(lqs), hq, dtype=torch.float32, device="cuda")
    o = ops.attn_prefill_varlen(dev(q), dev(ref.to_head_major(kc)), dev(ref.to_head_major(vc)), dev(cuq), dev(cuk),
                                max(lqs), scale, block_tables=dev(bt), lse=lse)
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err
    assert float((lse.cpu() - lse_ref).abs().max()) <= 2e-3


def test_prefill_softmax_rescale_branch(ops):
    """Force a large running-max jump at a late tile (guide §5.4 rule 26): spike one key."""
    hq = hkv = 8
    lens = [512]
    gen = g(60)
    q = torch.randn(512, hq, 128, generator=gen).to(BF16)
    k = torch.randn(512, hkv, 128, generator=gen).to(BF16)
    v = torch.randn(512, hkv, 128, generator=gen).to(BF16)
    k[300] = (q[400] * 4).to(BF16)  # key 300 dominates query 400 (and is visible to it)
    cu = _cu(lens)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_varlen_func(q, k, v, 512, cu, 512, cu, scale, True, None, return_softmax_lse=True)
    lse = torch.zeros(512, hq, dtype=torch.float32, device="cuda")
    o = ops.attn_prefill_varlen(dev(q), dev(k), dev(v), dev(cu), dev(cu), 512, scale, lse=lse)
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err
    assert float((lse.cpu() - lse_ref).abs().max()) <= 1e-2          # scores up to ~180 here: fp32 spacing 1.5e-5 x sums



_RESCALE_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from nano_vllm_amd import ops
ops.load_library()
d = torch.load(sys.argv[2])
out = {}
for name, (q, k, v, n) in d.items():
    cu = torch.tensor([0, n], dtype=torch.int32, device="cuda")
    lse = torch.zeros(n, q.shape[1], dtype=torch.float32, device="cuda")
    o = ops.attn_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), cu, cu, n, 128 ** -0.5, lse=lse)
    out[name] = (o.cpu(), lse.cpu())
torch.save(out, sys.argv[3])
"""


def test_prefill_deferred_rescale_equals_immediate_rescale(ops, tmp_path):
    """The prefill kernel brings O / l to a new row maximum only when it grew by more than 8 (log2 units) since the last
    rescale (cdna_hip_programming.md T13; flash-attn rescales whenever it moves). Guide rule 26: a passing comparison on
    bounded random data says nothing about the branch — so three inputs, each run in a child process per threshold
    (NVL_PREFILL_RESCALE_THR is read once): plain random data (maxima creep: deferred path only), one key that lifts a
    query's maximum by ~5 (still deferred: P up to 2^5 against the stale maximum) and one that lifts it by far more than
    the threshold at a late tile (the branch). THR = 0 == THR = 8 to rounding, and both == the oracle."""
    import subprocess
    import sys
    hq, hkv, n = 8, 4, 2304          # 8-wave shape (>= 2048 rows), several 64-key tiles after the spikes
    gen = g(61)
    cases = {}
    for name, gain in (("random", 0.0), ("small_jump", 0.62), ("big_jump", 4.0)):
        q = torch.randn(n, hq, 128, generator=gen).to(BF16)
        k = torch.randn(n, hkv, 128, generator=gen).to(BF16)
        v = torch.randn(n, hkv, 128, generator=gen).to(BF16)
        if gain:                     # key 1500 aligned with query 2000 of head 2 (kv head 1): raw score ~ gain * |q|^2
            k[1500, 1] = (q[2000, 2].float() * gain).to(BF16)
        cases[name] = (q, k, v, n)
    inp = tmp_path / "in.pt"
    torch.save(cases, inp)
    res = {}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for thr in ("0", "8"):
        outp = tmp_path / f"out{thr}.pt"
        r = subprocess.run([sys.executable, "-c", _RESCALE_CHILD, root, str(inp), str(outp)],
                           env=dict(os.environ, NVL_PREFILL_RESCALE_THR=thr), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[thr] = torch.load(outp)
    scale = 128 ** -0.5
    for name, (q, k, v, _) in cases.items():
        cu = _cu([n])
        o_ref, lse_ref = ref.flash_attn_varlen_func(q, k, v, n, cu, n, cu, scale, True, None, return_softmax_lse=True)
        amax = o_ref.float().abs().max().item()
        (o0, l0), (o8, l8) = res["0"][name], res["8"][name]
        for o, lse in ((o0, l0), (o8, l8)):
            assert (o.float() - o_ref.float()).abs().max().item() <= 2e-2 * amax + 1e-3, name
            assert float((lse - lse_ref).abs().max()) <= 1e-2, name
        assert (o0.float() - o8.float()).abs().max().item() <= 1e-2 * amax + 1e-3, name     # both are roundings of one value
        assert float((l0 - l8).abs().max()) <= 1e-3, name
    # the jumps really are what the docstring says (log2 units, query 2000 of head 2 against the rest of its row)
    q, k, _, _ = cases["small_jump"]
    s = (q[2000, 2].float() @ k[:2001, 1].float().T) * scale * 1.4426950408889634
    assert 2.0 < (s[1500] - torch.cat([s[:1500], s[1501:]]).max()).item() < 8.0
    q, k, _, _ = cases["big_jump"]
    s = (q[2000, 2].float() @ k[:2001, 1].float().T) * scale * 1.4426950408889634
    assert (s[1500] - torch.cat([s[:1500], s[1501:]]).max()).item() > 16.0


def _oracle_attend_chunked(q, k, v, scale, off, chunk=1024):
    """oracle/ops.py::_attend over query blocks (its [Hq, Lq, Lk] fp32 score tensor would not fit at 16 k): rows
    [i0, i1) of a bottom-right aligned causal problem see keys j <= i + off."""
    out = torch.empty_like(q)
    for i0 in range(0, q.shape[0], chunk):
        i1 = min(q.shape[0], i0 + chunk)
        kend = i1 + off
        out[i0:i1] = ref._attend(q[i0:i1], k[:kend], v[:kend], scale, i0 + off)
    return out


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 1)])
@pytest.mark.parametrize("n", [4096, 16384])
def test_prefill_long_sequence_vs_chunked_oracle(ops, hq, hkv, n):
    """BASELINE config 5's kernel shape (one 16 k-token prompt per step; 8/1 heads = Qwen3-32B per rank at TP=8):
    256 key tiles, longest-first dispatch, per-wave causal tile skipping. The oracle arithmetic (fp32 scores and
    softmax, P rounded to bf16 before P.V — oracle/ops.py::_attend) runs on the GPU in query blocks."""
    gen = g(70 + hq)
    q = torch.randn(n, hq, 128, generator=gen).to(BF16).cuda()
    k = torch.randn(n, hkv, 128, generator=gen).to(BF16).cuda()
    v = torch.randn(n, hkv, 128, generator=gen).to(BF16).cuda()
    cu = dev(_cu([n]))
    scale = 128 ** -0.5
    o = ops.attn_prefill_varlen(q, k, v, cu, cu, n, scale)
    o_ref = _oracle_attend_chunked(q, k, v, scale, 0)
    err = (o.float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err
    # the early rows attend over few keys (large values), the late rows over 16 k (values ~ 1/sqrt(n)): judge the
    # tail on its own scale too
    tail = slice(n - 1024, n)
    err_t = (o[tail].float() - o_ref[tail].float()).abs().max().item()
    assert err_t <= 2e-2 * o_ref[tail].float().abs().max().item() + 1e-3, err_t


@pytest.mark.parametrize("hq,hkv", [(8, 1), (16, 8), (40, 8), (20, 4), (5, 1), (24, 8)])
def test_prefill_paged_continuation_of_a_prompt_longer_than_the_token_budget(ops, hq, hkv):
    """scheduler.py:42-46 chunk rule at max_num_batched_tokens = 16384: a 20,000-token prompt is prefilled as
    16,384 tokens, then 3,616 tokens whose keys are ALL 20,000 tokens read from the paged cache through the
    block table (layers/attention.py:65-66) with the bottom-right aligned mask."""
    bs, lk, lq = 256, 20000, 3616
    kc, vc, bt = _paged_setup([lk], hkv, bs, seed=80)
    gen = g(81)
    q = torch.randn(lq, hq, 128, generator=gen).to(BF16)
    cuq, cuk = _cu([lq]), _cu([lk])
    scale = 128 ** -0.5
    o = ops.attn_prefill_varlen(dev(q), dev(ref.to_head_major(kc)), dev(ref.to_head_major(vc)), dev(cuq), dev(cuk),
                                lq, scale, block_tables=dev(bt))
    ks = ref._gather_paged(kc, bt[0], lk).cuda()
    vs = ref._gather_paged(vc, bt[0], lk).cuda()
    o_ref = _oracle_attend_chunked(q.cuda(), ks, vs, scale, lk - lq)
    err = (o.float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err


@pytest.mark.slow      # (19 s: re-runs the prefill tests of this file in a child process with the other workgroup shape forced)
def test_prefill_other_workgroup_shape_passes_the_same_tests():
    """The prefill kernel exists in two workgroup shapes (4 waves = one q-head, 8 waves = two q-heads of a kv group
    sharing the staged K/V tile), chosen per launch by max_seqlen_q unless NVL_PREFILL_WAVES forces one: run the oracle
    comparisons of this file once more in a child process with the 8-wave shape forced for EVERY launch (the default
    only takes it for long sequences)."""
    import subprocess
    import sys
    env = dict(os.environ, NVL_PREFILL_WAVES="8")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "prefill and not other_workgroup_shape"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.slow      # (20 s: the prefill tests once more with the one-wave-per-SIMD shape forced for every packed launch)
def test_prefill_w64_shape_passes_the_same_tests():
    """attn_prefill64.hip (generated asm main loop, 64 rows per wave) serves packed launches of long prompts by default — the
    16 k / 4 k comparisons and the deferred-rescale test of this file run on it; here every packed launch of the file takes it
    (NVL_PREFILL_W64=2): ragged short sequences, rows past the end of a 256-row tile, masks on every tile, the LSE output."""
    import subprocess
    import sys
    env = dict(os.environ, NVL_PREFILL_W64="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "prefill and not other_workgroup_shape and not w64_shape"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------
# Opt-in fp8 (OCP e4m3) KV cache. Oracle = the same restatement evaluated on a cache whose values
# went through torch's own float8_e4m3fn cast: the store must reproduce that cast BIT FOR BIT (round to nearest
# even), the attention kernels must agree with the oracle on the dequantised cache to the bf16-path tolerance.
def _fp8_roundtrip(t):
    return t.to(torch.float8_e4m3fn).to(BF16)


@pytest.mark.parametrize("n,h,hkv", [(1, 16, 8), (131, 16, 8), (40, 32, 8)])
def test_fp8_kv_store_matches_torch_cast_bit_for_bit(ops, n, h, hkv):
    gen = g(120 + n)
    bs, nblk = 256, 4
    qkv = (torch.randn(n, (h + 2 * hkv) * 128, generator=gen) * 3).to(BF16)
    pos = torch.randint(0, 2000, (n,), generator=gen)
    slots = torch.randperm(nblk * bs, generator=gen)[:n].to(torch.int32)
    slots[0] = -1 if n > 1 else slots[0]
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16)
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16)
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128))
    fr = torch.arange(2048).float()[:, None] * inv[None]
    table = torch.cat([fr.cos(), fr.sin()], -1).contiguous()
    caches = {}
    for name, dt in (("bf16", BF16), ("fp8", torch.float8_e4m3fn)):
        kc = torch.zeros(nblk, hkv, bs, 128, dtype=torch.uint8 if dt != BF16 else BF16, device="cuda")
        vc = torch.zeros_like(kc)
        if dt != BF16:
            kc, vc = kc.view(dt), vc.view(dt)
        q = torch.empty(n, h, 128, dtype=BF16, device="cuda")
        ops.qknorm_rope_kvstore(dev(qkv), dev(pos), dev(qw), dev(kw), 1e-6, dev(table), dev(slots), q, None, kc, vc, h, hkv)
        caches[name] = (kc, vc)
    for i in (0, 1):
        want = caches["bf16"][i].cpu().to(torch.float8_e4m3fn)
        assert torch.equal(caches["fp8"][i].cpu().view(torch.uint8), want.view(torch.uint8))
    # plain store entry point
    k = (torch.randn(n, hkv, 128, generator=gen) * 2).to(BF16)
    v = (torch.randn(n, hkv, 128, generator=gen) * 2).to(BF16)
    kc = torch.zeros(nblk, hkv, bs, 128, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn)
    vc = torch.zeros_like(kc)
    ops.store_kvcache(dev(k), dev(v), kc, vc, dev(slots))
    ref_k = torch.zeros(nblk * bs, hkv, 128, dtype=BF16)
    keep = slots >= 0
    ref_k[slots[keep].long()] = k[keep]
    got = kc.cpu().view(torch.uint8).view(nblk, hkv, bs, 128).permute(0, 2, 1, 3).reshape(nblk * bs, hkv, 128)
    assert torch.equal(got, ref_k.to(torch.float8_e4m3fn).view(torch.uint8))
    # magnitudes beyond the e4m3 range saturate at +-448 instead of turning into NaN (which would poison every later
    # decode step of the sequence that reads the row back)
    big = torch.zeros(n, hkv, 128, dtype=BF16)
    big[-1, 0, :4] = torch.tensor([1000.0, -3.0e4, 448.0, 460.0]).to(BF16)
    kc.zero_()
    vc.zero_()
    s2 = torch.arange(n, dtype=torch.int32)
    ops.store_kvcache(dev(big), dev(big), kc, vc, dev(s2))
    row = kc.cpu().float().view(nblk, hkv, bs, 128)[(n - 1) // bs, 0, (n - 1) % bs, :4]
    assert row.tolist() == [448.0, -448.0, 448.0, 448.0] and not torch.isnan(kc.cpu().float()).any()


@pytest.mark.parametrize("hq,hkv", [(16, 8), (8, 8), (32, 8), (8, 1), (16, 2), (64, 8), (40, 8), (20, 4), (5, 1), (24, 8)])
@pytest.mark.parametrize("lens", [[1], [255, 256, 257, 33], [1, 100, 1023, 1024, 1025, 2048, 17, 0, 640], [4096, 3, 0, 700]])
def test_fp8_kv_decode_fused_vs_oracle_on_the_dequantised_cache(ops, hq, hkv, lens):
    """nvl_paged_attn_decode_fused with an fp8 cache: (1) the new token's K/V rows land in the cache as the fp8 cast
    of what the bf16 path stores; (2) the output equals the oracle's flash_attn_with_kvcache over the DEQUANTISED
    cache (which includes the new token's quantised row: this step sees what later steps will read)."""
    bs, max_ctx = 256, 4096
    gen = g(130)
    b = len(lens)
    nb = [(n + bs - 1) // bs for n in lens]
    total = sum(nb) + 3
    kc16 = _fp8_roundtrip(torch.randn(total, hkv, bs, 128, generator=gen).to(BF16))
    vc16 = _fp8_roundtrip(torch.randn(total, hkv, bs, 128, generator=gen).to(BF16))
    perm = torch.randperm(total, generator=gen).tolist()
    bt = torch.full((b, max_ctx // bs), -1, dtype=torch.int32)
    c = 0
    for s_, n in enumerate(nb):
        for j in range(n):
            bt[s_, j] = perm[c]
            c += 1
    qkv = torch.randn(b, (hq + 2 * hkv) * 128, generator=gen).to(BF16)
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16)
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16)
    inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128))
    fr = torch.arange(max_ctx).float()[:, None] * inv[None]
    table = torch.cat([fr.cos(), fr.sin()], -1).contiguous()
    ctx = torch.tensor(lens, dtype=torch.int32)
    pos = (ctx.long() - 1).clamp(min=0)
    slots = torch.tensor([int(bt[i, (n - 1) // bs]) * bs + (n - 1) % bs if n > 0 else -1 for i, n in enumerate(lens)],
                         dtype=torch.int32)
    scale = 128 ** -0.5
    # reference: bf16 pipeline for q / new k, v; then quantise the cache rows
    kc_ref, vc_ref = dev(kc16.clone()), dev(vc16.clone())
    q1 = torch.empty(b, hq, 128, dtype=BF16, device="cuda")
    ops.qknorm_rope_kvstore(dev(qkv), dev(pos), dev(qw), dev(kw), 1e-6, dev(table), dev(slots), q1, None, kc_ref, vc_ref, hq, hkv)
    kc_deq, vc_deq = _fp8_roundtrip(kc_ref.cpu()), _fp8_roundtrip(vc_ref.cpu())
    # ours: fp8 cache holding the same values
    kc8 = dev(kc16.to(torch.float8_e4m3fn))
    vc8 = dev(vc16.to(torch.float8_e4m3fn))
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, max_ctx), dtype=torch.uint8, device="cuda")
    o = ops.paged_attn_decode_fused(dev(qkv), dev(qw), dev(kw), 1e-6, dev(table), kc8, vc8, dev(bt), dev(ctx), hq, scale,
                                    max_ctx, ws)
    torch.cuda.synchronize()
    assert torch.equal(kc8.cpu().view(torch.uint8), kc_deq.to(torch.float8_e4m3fn).view(torch.uint8))
    assert torch.equal(vc8.cpu().view(torch.uint8), vc_deq.to(torch.float8_e4m3fn).view(torch.uint8))
    live = [i for i, n in enumerate(lens) if n > 0]
    o_ref = ref.flash_attn_with_kvcache(q1.cpu().unsqueeze(1), ref.from_head_major(kc_deq), ref.from_head_major(vc_deq),
                                        ctx, bt, scale).squeeze(1)
    d = (o.cpu().float()[live] - o_ref.float()[live]).abs().max()
    assert float(d) <= 2e-2 * float(o_ref.float()[live].abs().max()) + 1e-3
    for i, n in enumerate(lens):
        if n == 0:
            assert not o[i].any()
    # unfused entry on the same cache
    o2 = ops.paged_attn_decode(q1, kc8, vc8, dev(bt), dev(ctx), scale, max_ctx, torch.zeros_like(ws))
    d2 = (o2.cpu().float()[live] - o_ref.float()[live]).abs().max()
    assert float(d2) <= 2e-2 * float(o_ref.float()[live].abs().max()) + 1e-3


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (40, 8), (20, 4), (5, 1), (24, 8)])
def test_fp8_kv_prefill_paged_prefix(ops, hq, hkv):
    """Prefix-cache / chunk-continuation prefill reading an fp8 cache == the oracle on the dequantised cache."""
    bs = 256
    lq_lk = [(100, 356), (256, 256), (7, 1031), (300, 812)]
    lqs = [a for a, _ in lq_lk]
    lks = [b for _, b in lq_lk]
    kc, vc, bt = _paged_setup(lks, hkv, bs, seed=140)
    kc, vc = _fp8_roundtrip(kc), _fp8_roundtrip(vc)
    gen = g(141)
    q = torch.randn(sum(lqs), hq, 128, generator=gen).to(BF16)
    cuq, cuk = _cu(lqs), _cu(lks)
    scale = 128 ** -0.5
    o_ref = ref.flash_attn_varlen_func(q, kc, vc, max(lqs), cuq, max(lks), cuk, scale, True, bt)
    o = ops.attn_prefill_varlen(dev(q), dev(ref.to_head_major(kc).to(torch.float8_e4m3fn)),
                                dev(ref.to_head_major(vc).to(torch.float8_e4m3fn)), dev(cuq), dev(cuk), max(lqs), scale,
                                block_tables=dev(bt))
    err = (o.cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item() + 1e-3, err


# ------------------------------------------------------------------------------------------
# The pool re-sized for 288 GB: element offsets beyond 2^31 (SURVEY.md §0-12 — the reference's Triton store computes
# `slot * D` in int32, attention.py:22-30, and overflows at this size). A one-layer pool of 8,448 blocks x 8 kv heads
# (4.4 GB each for K and V in bf16): block 8,192 starts at element 2^31; every kernel that addresses the cache is run
# on sequences living at block ids >= 8,192 (plus one below, so both sides of the boundary are in one launch) against
# the oracle, whose small caches hold the same blocks under small ids.
@pytest.mark.parametrize("kv", ["bf16", "fp8"])
def test_kv_pool_beyond_2_pow_31_elements(ops, kv):
    bs, hkv, hq, nblk, max_ctx = 256, 8, 16, 8448, 2048
    fp8 = kv == "fp8"
    gen = g(150)
    free, _ = torch.cuda.mem_get_info()
    need = 2 * nblk * hkv * bs * 128 * (1 if fp8 else 2)
    if free < need + (6 << 30):
        pytest.skip(f"needs {need / 2**30:.1f} GiB of free device memory")
    dt_dev = torch.float8_e4m3fn if fp8 else BF16
    kc = torch.zeros(nblk, hkv, bs, 128, dtype=torch.uint8 if fp8 else BF16, device="cuda")
    vc = torch.zeros_like(kc)
    if fp8:
        kc, vc = kc.view(dt_dev), vc.view(dt_dev)
    assert kc.numel() > 2 ** 31 and (nblk - 1) * hkv * bs * 128 > 2 ** 31
    # sequences: (context length incl. the token this step appends) on blocks picked from the top of the pool
    lens = [700, 257, 1024, 33, 512, 2048]
    nb = [(n + bs - 1) // bs for n in lens]
    big_ids = [8447, 8192, 8193, 8300, 8446, 8200, 8250, 8400, 8199, 8345, 8222, 8333, 8444, 8211, 8191, 8195,
               8196, 8197, 8198, 8201, 8202, 8203, 8204, 4097, 8205]
    assert len(set(big_ids)) == len(big_ids) >= sum(nb) and max(big_ids) < nblk
    small_of = {bid: i for i, bid in enumerate(big_ids)}             # the oracle's ids for the same blocks
    bt_big = torch.full((len(lens), max_ctx // bs), -1, dtype=torch.int32)
    c = 0
    for s_, n in enumerate(nb):
        for j in range(n):
            bt_big[s_, j] = big_ids[c]
            c += 1
    bt_small = bt_big.clone()
    for bid, sid in small_of.items():
        bt_small[bt_big == bid] = sid
    # fill the used blocks (product layout [Hkv, bs, 128] per block) and mirror them into the oracle's small caches
    quant = (lambda t: _fp8_roundtrip(t)) if fp8 else (lambda t: t)
    kc_small = torch.zeros(len(big_ids), bs, hkv, 128, dtype=BF16)
    vc_small = torch.zeros_like(kc_small)
    for bid, sid in small_of.items():
        kb = quant(torch.randn(hkv, bs, 128, generator=gen).to(BF16))
        vb = quant(torch.randn(hkv, bs, 128, generator=gen).to(BF16))
        kc[bid].copy_(kb.to(dt_dev))
        vc[bid].copy_(vb.to(dt_dev))
        kc_small[sid] = kb.permute(1, 0, 2)
        vc_small[sid] = vb.permute(1, 0, 2)

    def rows(cache, bid):                                              # block `bid` of the product pool, token-major bf16
        return cache[bid].cpu().to(BF16).permute(1, 0, 2) if fp8 else cache[bid].cpu().permute(1, 0, 2)

    # ---- nvl_store_kvcache at slots beyond 2^31 / (Hkv * 128) --------------------------------------------------------
    n_new = 40
    k_new = quant(torch.randn(n_new, hkv, 128, generator=gen).to(BF16))
    v_new = quant(torch.randn(n_new, hkv, 128, generator=gen).to(BF16))
    spare = [8440, 8441, 5]                                             # blocks no sequence uses
    slots_big = torch.tensor([spare[i % 3] * bs + (7 * i) % bs for i in range(n_new)], dtype=torch.int32)
    slots_big[3] = -1
    ops.store_kvcache(dev(k_new), dev(v_new), kc, vc, dev(slots_big))
    for i in range(n_new):
        if i == 3:
            continue
        bid, off = int(slots_big[i]) // bs, int(slots_big[i]) % bs
        assert torch.equal(rows(kc, bid)[off], k_new[i]) and torch.equal(rows(vc, bid)[off], v_new[i])
    assert not kc[8442].view(torch.uint8).any() and not kc[6].view(torch.uint8).any()    # neighbours untouched

    # ---- fused decode (q/k-norm + RoPE + KV store + attention), plain and planned -------------------------------------
    b = len(lens)
    qkv = torch.randn(b, (hq + 2 * hkv) * 128, generator=gen).to(BF16)
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16)
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).to(BF16)
    table = ref.rope_table(128, max_ctx, 1e6)
    ctx = torch.tensor(lens, dtype=torch.int32)
    pos = ctx.long() - 1
    slots_small = torch.tensor([int(bt_small[i, (n - 1) // bs]) * bs + (n - 1) % bs for i, n in enumerate(lens)],
                               dtype=torch.int32)
    slots_dec = torch.tensor([int(bt_big[i, (n - 1) // bs]) * bs + (n - 1) % bs for i, n in enumerate(lens)],
                             dtype=torch.int32)
    # oracle: the reference's separate steps on the small caches (qwen3.py:83-85, attention.py:63, :72-74)
    q, k, v = qkv.split([hq * 128, hkv * 128, hkv * 128], dim=-1)
    qn = ref.rms_forward(q.reshape(b, hq, 128), qw, 1e-6)
    kn = ref.rms_forward(k.reshape(b, hkv, 128), kw, 1e-6)
    q_ref, k_ref = ref.rotary_forward(pos, qn, kn, table)
    ref.store_kvcache(quant(k_ref), quant(v.reshape(b, hkv, 128)), kc_small, vc_small, slots_small)
    scale = 128 ** -0.5
    o_ref, lse_ref = ref.flash_attn_with_kvcache(q_ref.unsqueeze(1), kc_small, vc_small, ctx, bt_small, scale,
                                                 return_softmax_lse=True)
    o_ref = o_ref.squeeze(1)
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(b, hq, max_ctx), dtype=torch.uint8, device="cuda")
    lse = torch.zeros(b, hq, dtype=torch.float32, device="cuda")
    o = ops.paged_attn_decode_fused(dev(qkv), dev(qw), dev(kw), 1e-6, dev(table), kc, vc, dev(bt_big), dev(ctx), hq, scale,
                                    max_ctx, ws, lse=lse)
    torch.cuda.synchronize()
    tol = 2e-2 * float(o_ref.float().abs().max()) + (1e-3 if fp8 else 0.0)
    assert float((o.cpu().float() - o_ref.float()).abs().max()) <= tol
    assert float((lse.cpu() - lse_ref).abs().max()) <= (6e-3 if fp8 else 2e-3)
    for i, n in enumerate(lens):                                        # the appended row sits in its high block
        bid, off = int(slots_dec[i]) // bs, int(slots_dec[i]) % bs
        got_k, want_k = rows(kc, bid)[off].float(), kc_small[small_of[bid], off].float()
        assert float((got_k - want_k).abs().max()) <= (0.51 if fp8 else 2 ** -5)          # (1 ulp of norm -> rope)
        assert torch.equal(rows(vc, bid)[off], vc_small[small_of[bid], off])
    plan = ops.decode_plan(dev(ctx), hq, hkv, max_ctx)
    o2 = ops.paged_attn_decode_fused(dev(qkv), dev(qw), dev(kw), 1e-6, dev(table), kc, vc, dev(bt_big), dev(ctx), hq,
                                     scale, max_ctx, torch.zeros_like(ws), plan=plan)
    assert torch.equal(o2, o)                                           # (the store is idempotent: same row, same value)
    # unfused entry point on the same pool
    q1 = torch.empty(b, hq, 128, dtype=BF16, device="cuda")
    ops.qknorm_rope_kvstore(dev(qkv), dev(pos), dev(qw), dev(kw), 1e-6, dev(table), dev(slots_dec), q1, None, kc, vc, hq, hkv)
    o3 = ops.paged_attn_decode(q1, kc, vc, dev(bt_big), dev(ctx), scale, max_ctx, torch.zeros_like(ws))
    assert float((o3.cpu().float() - o_ref.float()).abs().max()) <= tol

    # ---- paged prefill (prefix cache / chunk continuation) reading the same high blocks -------------------------------
    lks = lens
    lqs = [100, 1, 300, 33, 256, 7]
    qp = torch.randn(sum(lqs), hq, 128, generator=gen).to(BF16)
    cuq, cuk = _cu(lqs), _cu(lks)
    op_ref, lsep_ref = ref.flash_attn_varlen_func(qp, kc_small, vc_small, max(lqs), cuq, max(lks), cuk, scale, True,
                                                  bt_small, return_softmax_lse=True)
    lsep = torch.zeros(sum(lqs), hq, dtype=torch.float32, device="cuda")
    op = ops.attn_prefill_varlen(dev(qp), kc, vc, dev(cuq), dev(cuk), max(lqs), scale, block_tables=dev(bt_big), lse=lsep)
    assert float((op.cpu().float() - op_ref.float()).abs().max()) <= 2e-2 * float(op_ref.float().abs().max()) + 1e-3
    assert float((lsep.cpu() - lsep_ref).abs().max()) <= 2e-3
    del kc, vc
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------
def test_sampler_greedy_exact(ops):
    b, vocab = 37, 151936
    logits = torch.randn(b, vocab, generator=g(70)).to(BF16)
    logits[3, 100] = logits[3, 90000] = 50.0  # tie -> lowest index
    t = torch.zeros(b)
    ws = torch.empty(ops.sample_workspace_bytes(b), dtype=torch.uint8, device="cuda")
    out = ops.sample(dev(logits), dev(t), seed=1, offset=0, workspace=ws)
    assert torch.equal(out.cpu(), ref.greedy(logits))
    assert out[3].item() == 100


def test_sampler_replay_with_host_rng(ops):
    """T>0: the GPU pick must be the argmax of l/T - log E under the SAME Philox draw."""
    b, vocab = 5, 4099  # ragged vocab exercises the tail path
    logits = (torch.randn(b, vocab, generator=g(71)) * 2).to(BF16)
    t = torch.tensor([0.6, 1.0, 0.3, 2.0, 0.6])
    ws = torch.empty(ops.sample_workspace_bytes(b), dtype=torch.uint8, device="cuda")
    out = ops.sample(dev(logits), dev(t), seed=1234, offset=77, workspace=ws).cpu()
    for r in range(b):
        e = torch.from_numpy(ops.sample_exponentials_host(1234, 77, r, 0, vocab))
        keys = ref.sampler_keys(logits[r: r + 1], t[r: r + 1], e.unsqueeze(0))[0]
        assert keys[out[r]] >= keys.max() - 1e-3


def test_sampler_replay_at_the_real_vocabulary_with_the_restated_draw(ops):
    """T > 0 at the headline shape: B = 131 rows (the bench's mean decode batch) x V = 151,936, keyed rows (request
    ordinal | position << 32, as the engine stages them) — every pick must be the argmax of `l/T - log E` with E rebuilt
    by oracle/philox.py (an independent numpy restatement of the draw, checked against the Philox KATs on CPU), both for
    the full-row kernel and for 8 vocabulary shards + merge (what a TP = 8 step runs)."""
    import numpy as np
    from oracle.philox import race_keys
    b, vocab, seed = 131, 151936, (5 << 32) | 4242
    logits = (torch.randn(b, vocab, generator=g(73)) * 2.5).to(BF16)
    t = torch.tensor([(0.6, 1.0, 0.3, 1.7)[i % 4] for i in range(b)])
    ordinal = torch.arange(b, dtype=torch.int64) * 3 + 1
    position = 100 + 7 * torch.arange(b, dtype=torch.int64)
    position[5] = (1 << 24) + 9                                    # exercises the high counter word of the position
    keys = ordinal | (position << 32)
    ws = torch.empty(ops.sample_workspace_bytes(512), dtype=torch.uint8, device="cuda")
    out = ops.sample(dev(logits), dev(t), seed=seed, offset=0, workspace=ws, row_keys=dev(keys)).cpu()
    per = vocab // 8
    packed = torch.zeros(8, 512, 2, dtype=torch.int32, device="cuda")
    dl = dev(logits)
    for r in range(8):
        ops.sample_shard(dl[:, r * per:(r + 1) * per].contiguous(), dev(t), r * per, seed, 0, ws, packed[r],
                         row_keys=dev(keys))
    merged = ops.sample_merge(packed, 8, b, torch.empty(b, dtype=torch.int64, device="cuda")).cpu()
    assert torch.equal(merged, out)
    lf = logits.float().numpy()
    exact = 0
    for r in range(b):
        k = race_keys(lf[r], float(t[r]), seed, int(ordinal[r]), int(position[r]))
        assert k[int(out[r])] >= k.max() - 2e-4, (r, int(out[r]), int(k.argmax()))
        exact += int(out[r]) == int(k.argmax())
    assert exact >= b - 1                                          # (a last-bit tie between hardware log and numpy's)
    assert len(set(out.tolist())) > b // 2


def test_sampler_distribution(ops):
    """chi-square goodness of fit of 20000 draws against softmax(l/T) on a small vocab."""
    vocab, draws = 64, 20000
    base = (torch.randn(vocab, generator=g(72)) * 1.5).to(BF16)
    logits = base.unsqueeze(0).repeat(draws, 1).contiguous()
    t = torch.full((draws,), 0.8)
    ws = torch.empty(ops.sample_workspace_bytes(draws), dtype=torch.uint8, device="cuda")
    out = ops.sample(dev(logits), dev(t), seed=99, offset=5, workspace=ws).cpu()
    p = torch.softmax(base.float() / 0.8, dim=-1)
    counts = torch.bincount(out, minlength=vocab).float()
    expected = p * draws
    keep = expected >= 5
    chi2 = (((counts - expected) ** 2) / expected)[keep].sum().item()
    dof = int(keep.sum()) - 1
    assert chi2 < dof + 5 * math.sqrt(2 * dof), (chi2, dof)


def test_sampler_row_keys_make_the_draw_independent_of_the_batch_row(ops):
    """row_keys = sequence | position << 32: a sequence draws the same token wherever it sits in the batch and
    whatever else is in the batch; without keys the draw follows the batch row."""
    b, v = 12, 4096
    logits = (torch.randn(b, v, generator=g(91)) * 2).to(BF16)
    temps = torch.full((b,), 0.9)
    keys = torch.arange(b, dtype=torch.int64) * 7 + 3 + (torch.arange(b, dtype=torch.int64) + 100 << 32)
    ws = torch.empty(ops.sample_workspace_bytes(512), dtype=torch.uint8, device="cuda")
    base = ops.sample(dev(logits), dev(temps), 5, 0, ws, row_keys=dev(keys)).cpu()
    perm = torch.randperm(b, generator=g(92))
    shuffled = ops.sample(dev(logits[perm]), dev(temps), 5, 0, ws, row_keys=dev(keys[perm])).cpu()
    assert torch.equal(shuffled, base[perm])
    sub = perm[:5]                                             # a smaller batch holding some of the same sequences
    assert torch.equal(ops.sample(dev(logits[sub]), dev(temps[:5]), 5, 0, ws, row_keys=dev(keys[sub])).cpu(), base[sub])
    # the key's high word is an offset: (row 3, position p) without keys == key (3 | p << 32) at offset 0
    unkeyed = ops.sample(dev(logits), dev(temps), 5, 100 + 3, ws).cpu()
    k3 = torch.tensor([3 + ((100 + 3) << 32)], dtype=torch.int64)
    assert int(ops.sample(dev(logits[3:4]), dev(temps[:1]), 5, 0, ws, row_keys=dev(k3)).cpu()) == int(unkeyed[3])
    assert not torch.equal(ops.sample(dev(logits[perm]), dev(temps), 5, 0, ws).cpu(), base[perm]) or b < 3
    # vocab-parallel shards with keys merge to the keyed full-row draw
    half = v // 2
    packed = torch.zeros(2, 512, 2, dtype=torch.int32, device="cuda")
    for r in range(2):
        ops.sample_shard(dev(logits[:, r * half:(r + 1) * half].contiguous()), dev(temps), r * half, 5, 0, ws, packed[r],
                         row_keys=dev(keys))
    out = torch.empty(b, dtype=torch.int64, device="cuda")
    assert torch.equal(ops.sample_merge(packed, 2, b, out).cpu(), base)


def test_sampler_shards_merge_to_the_full_row_draw(ops):
    """Vocab-parallel sampling (nvl_sample_shard + nvl_sample_merge) == nvl_sample on the concatenated row, for
    T > 0 (same Philox stream: keyed by the GLOBAL column) and T = 0 (lowest index on ties across shards)."""
    gen = g(55)
    b, vocab = 37, 151936
    logits = (torch.randn(b, vocab, generator=gen) * 3).to(BF16)
    logits[3, 100] = logits[3, 100000] = 30.0                 # a tie between shards at T = 0: lowest index wins
    temps = torch.tensor([0.0 if i % 3 == 0 else 0.7 for i in range(b)])
    ws = torch.empty(ops.sample_workspace_bytes(512), dtype=torch.uint8, device="cuda")
    full = ops.sample(dev(logits), dev(temps), seed=11, offset=3, workspace=ws).cpu()
    for parts in (2, 4, 8):
        per = vocab // parts
        packed = torch.zeros(parts, 512, 2, dtype=torch.int32, device="cuda")
        dl = dev(logits)
        for r in range(parts):
            shard = dl[:, r * per:(r + 1) * per].contiguous()
            ops.sample_shard(shard, dev(temps), r * per, 11, 3, ws, packed[r])
        out = torch.empty(b, dtype=torch.int64, device="cuda")
        ops.sample_merge(packed, parts, b, out)
        assert torch.equal(out.cpu(), full), parts
    assert int(full[3]) == 100


def test_feed_tokens(ops):
    """nvl_feed_tokens: ids[i] = prev[src[i]] where src[i] >= 0, untouched elsewhere (bit-exact index work)."""
    gen = g(50)
    prev = torch.randint(0, 151936, (300,), generator=gen)
    ids = torch.randint(0, 151936, (257,), generator=gen)
    src = torch.randint(-1, 300, (257,), generator=gen).to(torch.int32)
    want = torch.where(src >= 0, prev[src.clamp(min=0).long()], ids)
    d_ids = dev(ids.clone())
    ops.feed_tokens(d_ids, dev(src), dev(prev))
    assert torch.equal(d_ids.cpu(), want)


def test_errors_are_reported_not_thrown(ops):
    x = torch.zeros(4, 1000 + 4, dtype=BF16, device="cuda")  # hidden not a multiple of 8
    w = torch.zeros(1004, dtype=BF16, device="cuda")
    with pytest.raises(ops.NvlError, match="multiple of 8"):
        ops.rmsnorm(x, w, 1e-6)


def test_decode_plan_is_refused_for_a_launch_it_was_not_built_for(ops):
    """A plan is a list of per-wave records for ONE (batch, Hkv, max_context): the attention entry points compare the
    geometry recorded when nvl_decode_plan ran with the launch at hand and refuse a mismatch (the kernel would index the
    records by wave id and read the wrong segments), as well as a buffer nvl_decode_plan never filled."""
    bs, hq, hkv, max_ctx = 256, 16, 8, 1024
    lens = [300, 17, 1024, 5]
    kc, vc, bt = _paged_setup(lens, hkv, bs, seed=77)
    kc, vc = dev(ref.to_head_major(kc)), dev(ref.to_head_major(vc))
    btw = torch.full((len(lens), max_ctx // bs), -1, dtype=torch.int32)
    btw[:, :bt.shape[1]] = bt
    ctx = dev(torch.tensor(lens, dtype=torch.int32))
    q = dev(torch.randn(len(lens), hq, 128, generator=g(78)).to(BF16))
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(8, hq, 4096), dtype=torch.uint8, device="cuda")
    plan = ops.decode_plan(ctx, hq, hkv, max_ctx)
    good = ops.paged_attn_decode(q, kc, vc, dev(btw), ctx, 0.088, max_ctx, ws, plan=plan)
    assert torch.equal(good, ops.paged_attn_decode(q, kc, vc, dev(btw), ctx, 0.088, max_ctx, ws))
    with pytest.raises(ops.NvlError, match="plan was built for batch=4"):
        ops.paged_attn_decode(q[:3], kc, vc, dev(btw[:3]), ctx[:3].contiguous(), 0.088, max_ctx, ws, plan=plan)
    with pytest.raises(ops.NvlError, match="plan was built for"):
        wide = torch.full((len(lens), 2048 // bs), -1, dtype=torch.int32)
        wide[:, :btw.shape[1]] = btw
        ops.paged_attn_decode(q, kc, vc, dev(wide), ctx, 0.088, 2048, ws, plan=plan)
    # a buffer this step's nvl_decode_plan never filled: unknown to the library — or, when the allocator hands out an
    # address an earlier plan lived at, known with THAT plan's geometry; refused either way
    stray = torch.zeros(ops.decode_plan_bytes(), dtype=torch.uint8, device="cuda")
    with pytest.raises(ops.NvlError, match="not produced by nvl_decode_plan|plan was built for"):
        ops.paged_attn_decode(q, kc, vc, dev(btw), ctx, 0.088, max_ctx, ws, plan=stray)
