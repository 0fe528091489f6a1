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
    assert lib.nvl_abi_version() == 1


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
    rc = lib.nvl_paged_attn_decode(16, 16, 16, 16, 16, 16, 16, 4, 16, 6, 256, 8, 4096, 0.1, 16, 1 << 30, None)
    assert rc == -1 and b"not a multiple" in lib.nvl_last_error()


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
