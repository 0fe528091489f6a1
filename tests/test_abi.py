"""The C-ABI library builds for gfx950, loads, and exports exactly what include/nvl.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nvl.h")).read()
    return sorted(set(re.findall(r"\b(nvl_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from nano_vllm_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    for sym in _declared():
        assert hasattr(lib, sym), f"{sym} declared in include/nvl.h but not exported"
    lib.nvl_abi_version.restype = ctypes.c_int
    assert lib.nvl_abi_version() == 6


def test_ctypes_binding_matches_header():
    from nano_vllm_amd import ops
    assert sorted(ops.SIGNATURES) == _declared()
    # argument counts of the binding match the header prototypes
    text = open(os.path.join(ROOT, "include", "nvl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, (_, argtypes) in ops.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        assert n == len(argtypes), f"{name}: header has {n} parameters, binding has {len(argtypes)}"


def test_host_side_argument_validation_without_gpu():
    """Entry points validate on the host before launching: bad arguments return NVL_EINVAL with a
    message, even on a box without a GPU."""
    from nano_vllm_amd import ops
    lib = ops.load_library()
    rc = lib.nvl_rmsnorm(None, 0, None, None, 0, 1, 1, 1024, 1e-6, None)
    assert rc == -1 and b"null pointer" in lib.nvl_last_error()
    rc = lib.nvl_paged_attn_decode(16, 16, 16, 16, 16, 16, 16, 4, 16, 6, 256, 8, 4096, 0.1, 16, 1 << 30, 0, None, None, None)
    assert rc == -1 and b"not a multiple" in lib.nvl_last_error()
    rc = lib.nvl_paged_attn_decode(16, 16, 16, 16, 16, 16, 16, 4, 16, 8, 256, 8, 4096, 0.1, 16, 1 << 30, 7, None, None, None)
    assert rc == -1 and b"kv_dtype" in lib.nvl_last_error()
    # every group size Hq / Hkv from 1 to 16 has a kernel (round 6: Qwen3-14B is 40 / 8 = 5); beyond 16 the heads of a kv
    # group do not fit the kernel's one 16-column matrix tile: refused (either cache dtype), not emulated
    for kv_dtype in (0, 1):
        rc = lib.nvl_paged_attn_decode(16, 16, 16, 16, 16, 16, 16, 4, 34, 2, 256, 16, 4096, 0.1, 16, 1 << 30, kv_dtype, None, None, None)
        assert rc == -1 and b"group size" in lib.nvl_last_error()
    # fused decode entry: rope table is mandatory; q/k norm weights come as a pair
    rc = lib.nvl_paged_attn_decode_fused(16, 4096, None, None, 1e-6, None, 0, 16, 16, 16, 16, 16, 16, 4, 16, 8, 256, 8,
                                         4096, 0.1, 16, 1 << 30, 0, None, None, 0, 0, None)
    assert rc == -1 and b"rope table" in lib.nvl_last_error()
    rc = lib.nvl_paged_attn_decode_fused(16, 4096, 16, None, 1e-6, 16, 4096, 16, 16, 16, 16, 16, 16, 4, 16, 8, 256, 8,
                                         4096, 0.1, 16, 1 << 30, 0, None, None, 0, 0, None)
    assert rc == -1 and b"both be set or both NULL" in lib.nvl_last_error()
    # qkv as fp32 split-K slabs: at most 8, a slab at least one [batch, row] image long, matrix-core group sizes only
    rc = lib.nvl_paged_attn_decode_fused(16, 4096, None, None, 1e-6, 16, 4096, 16, 16, 16, 16, 16, 16, 4, 16, 8, 256, 8,
                                         4096, 0.1, 16, 1 << 30, 0, None, None, 9, 1 << 20, None)
    assert rc == -1 and b"qkv_splits" in lib.nvl_last_error()
    rc = lib.nvl_paged_attn_decode_fused(16, 4096, None, None, 1e-6, 16, 4096, 16, 16, 16, 16, 16, 16, 4, 16, 8, 256, 8,
                                         4096, 0.1, 16, 1 << 30, 0, None, None, 2, 100, None)
    assert rc == -1 and b"qkv_split_stride" in lib.nvl_last_error()
    rc = lib.nvl_paged_attn_decode_fused(16, 4096, None, None, 1e-6, 16, 4096, 16, 16, 16, 16, 16, 16, 4, 8, 8, 256, 8,
                                         4096, 0.1, 16, 1 << 30, 0, None, None, 2, 1 << 20, None)
    assert rc == -3 and b"matrix-core" in lib.nvl_last_error()          # Hq / Hkv = 1: the packed-dot kernel
    # per-step decode plan: buffer size is validated on the host
    rc = lib.nvl_decode_plan(16, 4, 16, 8, 4096, None, 0, 1, 16, 8, None)
    assert rc == -1 and b"plan buffer" in lib.nvl_last_error()
    # ... with a shared-prefix count (ABI v5): block size and head geometry are checked before anything is launched
    rc = lib.nvl_decode_plan(16, 4, 16, 8, 4096, 16, 96, 1, 16, 1 << 20, None)
    assert rc == -1 and b"block_size % 128" in lib.nvl_last_error()
    rc = lib.nvl_decode_plan(16, 4, 8, 8, 4096, 16, 256, 1, 16, 1 << 20, None)
    assert rc == -1 and b"matrix-core" in lib.nvl_last_error()
    rc = lib.nvl_decode_plan(16, 4, 16, 8, 4096, 16, 256, 9, 16, 1 << 20, None)      # (ABI v6: group slots of the pass)
    assert rc == -1 and b"shared_prefix_groups" in lib.nvl_last_error()
    # skinny linear: shape coverage is a query, an uncovered shape is EUNSUPPORTED (-3), never a silent fallback
    assert lib.nvl_linear_decode_splits(144, 4096, 1024, 0) == 1
    assert lib.nvl_linear_decode_splits(144, 1024, 2048, 2) == 2      # (three row groups x a 2-way K split from 9 row tiles on)
    assert lib.nvl_linear_decode_splits(64, 1024, 2048, 2) == 4
    assert lib.nvl_linear_decode_splits(144, 6144, 1024, 1) == 1
    assert lib.nvl_linear_decode_splits(144, 6144, 4096, 0) == 0          # deep-K shapes: not this kernel's (nvl_linear_wide)
    # collectives: a communicator must be created and connected first
    assert lib.nvl_allreduce_run(None, 16, 16, 4, 1024, None) == -1
    assert lib.nvl_linear_decode(16, 16, 16, 144, 6144, 4096, 0, 0, None) == -3 and b"not covered" in lib.nvl_last_error()
    assert lib.nvl_linear_decode(None, 16, 16, 144, 4096, 1024, 0, 0, None) == -1
    # wide-tile deep-K linear: plan query works without a GPU; uncovered shapes are EUNSUPPORTED, null pointers EINVAL
    import ctypes
    sp, ws = ctypes.c_int(0), ctypes.c_size_t(0)
    assert lib.nvl_linear_wide_plan(144, 6144, 4096, 0, ctypes.byref(sp), ctypes.byref(ws)) == 1
    assert sp.value >= 1 and ws.value == (sp.value * 144 * 6144 * 4 if sp.value > 1 else 0)
    assert lib.nvl_linear_wide_plan(144, 4096, 4096, 2, ctypes.byref(sp), ctypes.byref(ws)) == 1 and ws.value == 0
    assert lib.nvl_linear_wide_plan(144, 6144, 1000, 0, None, None) == 0
    assert lib.nvl_linear_wide(16, 16, 16, 144, 6144, 1000, 0, 0, None, 0, None) == -3 and b"not covered" in lib.nvl_last_error()
    assert lib.nvl_linear_wide(None, 16, 16, 144, 6144, 4096, 0, 0, None, 0, None) == -1
    assert lib.nvl_linear_wide(16, 16, 16, 144, 6144, 4096, 0, 7, None, 0, None) == -1 and b"weight_layout" in lib.nvl_last_error()
    assert lib.nvl_pack_weight_tiles(16, 32, 24, 64, None) == -1 and b"multiple of 16" in lib.nvl_last_error()


def test_linear_wide_plan_covers_the_model_shapes_on_the_host():
    """The plan query is pure host code: every decode projection of Qwen3-8B / 32B (full width and per-rank at TP 2 / 4 /
    8) is covered for every row count the engine can ask for, a K split always divides the plan's k steps (128 columns;
    64 in the one-row-group form for 193-256 rows), and the scratch size follows from it."""
    import ctypes
    from nano_vllm_amd import ops
    lib = ops.load_library()
    shapes = []
    for hidden, inter, heads, kv in ((4096, 12288, 32, 8), (5120, 25600, 64, 8)):
        for tp in (1, 2, 4, 8):
            shapes += [((heads + 2 * kv) * 128 // tp, hidden, 0), (hidden, heads * 128 // tp, 2),
                       (2 * inter // tp, hidden, 1), (hidden, inter // tp, 2), (hidden, heads * 128 // tp, 0)]
    sp, ws = ctypes.c_int(0), ctypes.c_size_t(0)
    for n, k, mode in shapes:
        for m in (1, 2, 8, 16, 17, 100, 131, 144, 145, 200, 256):
            assert lib.nvl_linear_wide_plan(m, n, k, mode, ctypes.byref(sp), ctypes.byref(ws)) == 1, (m, n, k, mode)
            step = 64 if m > 192 else 128
            assert 1 <= sp.value <= 32 and (k // step) % sp.value == 0, (m, n, k, mode, sp.value)
            want_ws = sp.value * m * n * 4 if (mode != 2 and sp.value > 1) else 0
            assert ws.value == want_ws, (m, n, k, mode, sp.value, ws.value)
    assert lib.nvl_add_rmsnorm_splitk(16, 0, 16, 16, 16, 4, 1024, 1e-6, None) == -1 and b"splits" in lib.nvl_last_error()


def test_wide_plan_tuned_entries_resolve_on_the_host():
    """The measured-best decompositions of gemm_wide.hip (kTuned) are plain data: each must resolve to a plan whose K
    split is the recorded one for every row count of its range, the neighbouring row counts must keep the model's pick,
    and NVL_WIDE_TUNED=0 must switch the table off."""
    import ctypes
    from nano_vllm_amd import ops
    lib = ops.load_library()
    sp, ws = ctypes.c_int(0), ctypes.c_size_t(0)

    def split(m, n, k, mode):
        assert lib.nvl_linear_wide_plan(m, n, k, mode, ctypes.byref(sp), ctypes.byref(ws)) == 1
        return sp.value

    for n, k, mode, rows, want in ((4096, 12288, 2, (64, 96, 131, 144), 8), (4096, 12288, 2, (208, 256), 4),
                                   (5120, 25600, 2, (96,), 4), (5120, 25600, 2, (112, 131, 144), 8),
                                   (5120, 25600, 2, (208, 256), 8), (10240, 5120, 0, (32, 64, 131, 144), 2),
                                   (10240, 5120, 0, (208, 256), 1), (4096, 4096, 2, (208,), 4)):
        for m in rows:
            assert split(m, n, k, mode) == want, (m, n, k, mode, sp.value)
    assert split(48, 4096, 12288, 2) == 4 and split(160, 4096, 12288, 2) == 4      # outside the ranges: the model's picks
    os.environ["NVL_WIDE_TUNED"] = "0"
    try:
        assert split(131, 4096, 12288, 2) == 4 and split(131, 10240, 5120, 0) == 1
    finally:
        del os.environ["NVL_WIDE_TUNED"]


def test_no_product_import_of_the_oracle():
    """The product path must never route through the oracle (or any CPU fallback)."""
    bad = []
    for pkg in ("nano_vllm_amd", "nanovllm"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from nano_vllm_amd import ops
    x = torch.zeros(4, 1024, dtype=torch.bfloat16)
    with pytest.raises(ops.NvlError, match="no CPU fallback"):
        ops.rmsnorm(x, torch.ones(1024, dtype=torch.bfloat16), 1e-6)


def test_generated_asm_cores_are_fresh():
    """csrc/gemm_wide_core.inc and gemm_tile4_core.inc are GENERATED (tools/gen_wide_asm.py is the schedule); the committed
    files must be what the generator writes."""
    import subprocess
    import sys
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_wide_asm.py"), "--check"]).returncode
    assert rc == 0, "run `python tools/gen_wide_asm.py` and commit the .inc files"
    # the prefill attention's main loop (csrc/attn_prefill64_core.inc) likewise
    env = {k: v for k, v in os.environ.items() if not k.startswith("NVL_PF64_")}
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_prefill_asm.py"), "--core", "--check"], env=env).returncode
    assert rc == 0, "run `python tools/gen_prefill_asm.py --core` and commit csrc/attn_prefill64_core.inc"
