"""Import the REAL reference (GeeeekExplorer/nano-vllm at /root/reference) on CPU — build
container only; /root/reference does not exist on the GPU box. TEST INFRASTRUCTURE.

What is needed to make `import nanovllm` (the reference) work here:
  * `flash_attn` is not installed: a stub module exposing flash_attn_varlen_func /
    flash_attn_with_kvcache is registered in sys.modules, implemented by the restatements in
    oracle/ops.py. (So at the flash-attn boundary the "reference" IS the oracle: unpinned.)
  * layers query torch.distributed in __init__: a 1-rank gloo group is created.
  * the reference package is also called `nanovllm`, like our alias package: it is imported
    under a private module cache (`load_reference()` returns the module objects and restores
    sys.modules afterwards), so tests can hold both.
Everything except engine/model_runner.py (hard-wired to "nccl"/"cuda") and the Triton
store_kvcache launcher runs unmodified.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nanovllm"))


def _flash_attn_stub() -> types.ModuleType:
    from . import ops
    mod = types.ModuleType("flash_attn")

    def flash_attn_varlen_func(q, k, v, max_seqlen_q, cu_seqlens_q, max_seqlen_k, cu_seqlens_k, softmax_scale,
                               causal=True, block_table=None):
        return ops.flash_attn_varlen_func(q, k, v, max_seqlen_q, cu_seqlens_q, max_seqlen_k, cu_seqlens_k,
                                          softmax_scale, causal, block_table)

    def flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens=None, block_table=None, softmax_scale=None,
                                causal=True):
        return ops.flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale, causal)

    mod.flash_attn_varlen_func = flash_attn_varlen_func
    mod.flash_attn_with_kvcache = flash_attn_with_kvcache
    return mod


_cache: dict | None = None


def load_reference(eager: bool = True) -> dict:
    """Returns {"nanovllm.engine.scheduler": module, ...} for the reference's modules.
    eager=True disables torch.compile (TORCHDYNAMO_DISABLE=1) before torch is first used by them."""
    global _cache
    if _cache is not None:
        return _cache
    if not available():
        raise RuntimeError("/root/reference is not present (GPU box?): use the committed tests/golden fixtures")
    if eager:
        os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    import torch.distributed as dist
    if not dist.is_initialized():
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    saved = {k: v for k, v in sys.modules.items() if k == "nanovllm" or k.startswith("nanovllm.") or k == "flash_attn"}
    for k in saved:
        del sys.modules[k]
    saved_meta = list(sys.meta_path)
    sys.meta_path[:] = [f for f in sys.meta_path if getattr(f, "__name__", "") != "_AliasFinder"]
    sys.modules["flash_attn"] = _flash_attn_stub()
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        names = ["nanovllm.config", "nanovllm.sampling_params", "nanovllm.engine.sequence",
                 "nanovllm.engine.block_manager", "nanovllm.engine.scheduler", "nanovllm.utils.context",
                 "nanovllm.layers.layernorm", "nanovllm.layers.rotary_embedding", "nanovllm.layers.activation",
                 "nanovllm.layers.sampler", "nanovllm.layers.attention", "nanovllm.layers.linear",
                 "nanovllm.layers.embed_head", "nanovllm.models.qwen3", "nanovllm.utils.loader"]
        # the reference's package __init__ imports llm -> llm_engine -> model_runner (fine on CPU: only
        # constructing ModelRunner needs CUDA)
        mods = {n: importlib.import_module(n) for n in names}
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k in [k for k in sys.modules if k == "nanovllm" or k.startswith("nanovllm.")]:
            del sys.modules[k]
        sys.modules.pop("flash_attn", None)
        sys.modules.update(saved)
        sys.meta_path[:] = saved_meta
    _cache = mods
    return mods
