"""nano_vllm_amd — MI355X-native paged-KV inference engine, drop-in for nano-vllm's
`LLM` / `SamplingParams` / `generate()` (also importable as `nanovllm`).

Python host on PyTorch-ROCm -> C-ABI library of hand-written gfx950 HIP kernels
(include/nvl.h, nano_vllm_amd/csrc). Importing this package does not need a GPU; constructing
`LLM` does.
"""
from .api import Config, SamplingParams

__all__ = ["LLM", "SamplingParams", "Config", "LLMEngine"]


def __getattr__(name):      # engine (and torch) are imported on first use
    if name in ("LLM", "LLMEngine"):
        from .engine import core
        return getattr(core, name)
    raise AttributeError(f"module 'nano_vllm_amd' has no attribute {name!r}")
