"""Public configuration surface of the drop-in: `Config` and `SamplingParams`.

Mirrors the reference's field names, defaults and assertions (nano-vllm config.py:7-25,
sampling_params.py:5-11) so callers of `nanovllm.LLM(model, **kwargs)` switch over unchanged.
Extensions (all optional, behaviour-neutral when unset):
  * SamplingParams.temperature == 0 selects greedy decoding (the reference forbids it; the
    parity harness needs it — SURVEY.md §0-7).
  * Config.seed seeds the counter-based sampler RNG.
  * Config.dummy_weights = True initialises random weights on device instead of reading
    *.safetensors (no 8B/32B checkpoints exist offline).
  * Config.kv_cache_dtype = "fp8" stores K/V as OCP fp8 e4m3: the decode step is bound by
    K/V bytes, this halves them. It changes numerics, so every parity run keeps the default "bf16".
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass(slots=True)
class SamplingParams:
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False

    def __post_init__(self):
        assert self.temperature == 0 or self.temperature > 1e-10, "temperature must be 0 (greedy) or > 1e-10"
        assert self.max_tokens >= 1


@dataclass(slots=True)
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: object | None = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1
    # --- extensions -------------------------------------------------------------------------
    seed: int = 0
    dummy_weights: bool = False
    kv_cache_dtype: str = "bf16"         # "fp8": OCP e4m3 KV cache (halves decode bytes; outside the reference's numerics)

    def __post_init__(self):
        assert os.path.isdir(self.model), f"model directory not found: {self.model}"
        assert self.kvcache_block_size % 256 == 0
        assert 1 <= self.tensor_parallel_size <= 8
        if self.hf_config is None:
            from transformers import AutoConfig
            self.hf_config = AutoConfig.from_pretrained(self.model)
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
        assert self.max_num_batched_tokens >= 1 and self.max_num_seqs >= 1
        assert self.kv_cache_dtype in ("bf16", "fp8"), "kv_cache_dtype must be 'bf16' or 'fp8'"


def model_geometry(hf_config, tp: int = 1) -> dict:
    """Per-rank shapes of a Qwen3-family config (models/qwen3.py:29-39)."""
    heads = hf_config.num_attention_heads
    kv_heads = hf_config.num_key_value_heads
    head_dim = getattr(hf_config, "head_dim", None) or hf_config.hidden_size // heads
    assert heads % tp == 0 and kv_heads % tp == 0 and hf_config.vocab_size % tp == 0
    assert hf_config.intermediate_size % tp == 0
    rope_theta = getattr(hf_config, "rope_theta", None)
    for attr in ("rope_scaling", "rope_parameters"):
        d = getattr(hf_config, attr, None)
        if isinstance(d, dict) and "rope_theta" in d:
            rope_theta = d["rope_theta"]
    if rope_theta is None:
        rope_theta = 1000000
    dtype = getattr(hf_config, "dtype", None) or getattr(hf_config, "torch_dtype", None)
    return dict(
        hidden=hf_config.hidden_size, layers=hf_config.num_hidden_layers, heads=heads // tp, kv_heads=kv_heads // tp,
        head_dim=head_dim, inter=hf_config.intermediate_size // tp, vocab=hf_config.vocab_size,
        vocab_per_rank=hf_config.vocab_size // tp, eps=hf_config.rms_norm_eps, rope_theta=float(rope_theta),
        max_pos=hf_config.max_position_embeddings, tie=bool(getattr(hf_config, "tie_word_embeddings", False)),
        qk_norm=not getattr(hf_config, "attention_bias", True), dtype=dtype)    # (default as models/qwen3.py:133)
