"""CPU restatement of the reference's Qwen3 forward pass (TEST INFRASTRUCTURE — see
oracle/__init__.py). Functional, TP=1, weights as a dict of HF-named bf16 tensors.

Follows nano-vllm models/qwen3.py (file:line cited inline) on top of oracle/ops.py. Attention
metadata is passed explicitly (`meta`) instead of through the reference's global Context
(utils/context.py); the KV cache uses the REFERENCE layout [L][num_blocks, block, Hkv, D].
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import ops


@dataclass
class Meta:
    """What the reference's Context carries (utils/context.py:6-14)."""
    is_prefill: bool
    cu_seqlens_q: torch.Tensor | None = None
    cu_seqlens_k: torch.Tensor | None = None
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    slot_mapping: torch.Tensor | None = None
    context_lens: torch.Tensor | None = None
    block_tables: torch.Tensor | None = None


class OracleQwen3:

    def __init__(self, cfg: dict, weights: dict[str, torch.Tensor], compiled: bool = True, device=None):
        """`device`: where the restatement's torch ops run. None / "cpu" is the oracle proper; a HIP device runs the
        SAME Python on torch's own kernels (hipBLASLt GEMMs, fp32 accumulate) — used by the GPU tests for workloads the
        CPU cannot finish in minutes (tests/test_e2e_gpu.py checks it against the CPU run first). Step metadata
        (cu_seqlens, context lengths, block tables) stays on the host; only bulk data lives on the device."""
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        weights = {k: v.to(self.dev) for k, v in weights.items()}
        self.cfg, self.w, self.compiled = cfg, weights, compiled
        self.h, self.hkv, self.d = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
        self.eps = cfg["rms_norm_eps"]
        self.L = cfg["num_hidden_layers"]
        self.table = ops.rope_table(self.d, cfg["max_position_embeddings"], cfg["rope_theta"]).to(self.dev)
        self.k_cache: list[torch.Tensor] = []
        self.v_cache: list[torch.Tensor] = []
        # fused projections exactly as the loader packs them (qwen3.py:187-193, linear.py:114-128)
        self.qkv, self.gate_up = [], []
        for i in range(self.L):
            p = f"model.layers.{i}."
            self.qkv.append(torch.cat([weights[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0))
            self.gate_up.append(torch.cat([weights[p + "mlp.gate_proj.weight"], weights[p + "mlp.up_proj.weight"]], 0))
        self.lm_head = weights["model.embed_tokens.weight"] if cfg["tie_word_embeddings"] else weights["lm_head.weight"]

    def allocate_cache(self, num_blocks: int, block_size: int):
        shape = (num_blocks, block_size, self.hkv, self.d)          # model_runner.py:115 (per layer)
        # dtype follows the weights: bf16 as the reference (hf_config.torch_dtype); an fp32 weight dict gives the
        # "exact arithmetic" variant used as the yardstick of the logits-error test (no intermediate rounding)
        dt = self.lm_head.dtype
        self.k_cache = [torch.zeros(shape, dtype=dt, device=self.dev) for _ in range(self.L)]
        self.v_cache = [torch.zeros(shape, dtype=dt, device=self.dev) for _ in range(self.L)]

    # -- layers/attention.py:59-75 ---------------------------------------------------------------
    def _attention(self, layer: int, q, k, v, meta: Meta):
        if self.k_cache:
            ops.store_kvcache(k, v, self.k_cache[layer], self.v_cache[layer], meta.slot_mapping.to(self.dev))   # :63
        scale = self.d ** -0.5                                                                       # qwen3.py:39
        if meta.is_prefill:
            if meta.block_tables is not None:                                                        # :65-66
                k, v = self.k_cache[layer], self.v_cache[layer]
            return ops.flash_attn_varlen_func(q, k, v, meta.max_seqlen_q, meta.cu_seqlens_q, meta.max_seqlen_k,
                                              meta.cu_seqlens_k, scale, True, meta.block_tables)     # :67-70
        o = ops.flash_attn_with_kvcache(q.unsqueeze(1), self.k_cache[layer], self.v_cache[layer],
                                        meta.context_lens, meta.block_tables, scale, True)           # :72-74
        return o.squeeze(1)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, meta: Meta) -> torch.Tensor:
        """Qwen3Model.forward (qwen3.py:173-183) -> hidden states [N, hidden]."""
        w, c = self.w, self.compiled
        input_ids, positions = input_ids.to(self.dev), positions.to(self.dev)
        hidden = F.embedding(input_ids, w["model.embed_tokens.weight"])                              # :178
        residual = None
        for i in range(self.L):
            p = f"model.layers.{i}."
            if residual is None:                                                                     # :152-153
                residual = hidden
                hidden = ops.rms_forward(hidden, w[p + "input_layernorm.weight"], self.eps, c)
            else:                                                                                    # :155
                hidden, residual = ops.add_rms_forward(hidden, residual, w[p + "input_layernorm.weight"], self.eps, c)
            qkv = F.linear(hidden, self.qkv[i])                                                      # :77
            q, k, v = qkv.split([self.h * self.d, self.hkv * self.d, self.hkv * self.d], dim=-1)     # :78
            q = q.reshape(-1, self.h, self.d)
            k = k.reshape(-1, self.hkv, self.d)
            v = v.reshape(-1, self.hkv, self.d)
            q = ops.rms_forward(q, w[p + "self_attn.q_norm.weight"], self.eps, c)                    # :83
            k = ops.rms_forward(k, w[p + "self_attn.k_norm.weight"], self.eps, c)                    # :84
            q, k = ops.rotary_forward(positions, q, k, self.table)                                   # :85
            o = self._attention(i, q, k, v, meta)                                                    # :86
            hidden = F.linear(o.flatten(1, -1), w[p + "self_attn.o_proj.weight"])                    # :87
            hidden, residual = ops.add_rms_forward(hidden, residual, w[p + "post_attention_layernorm.weight"],
                                                   self.eps, c)                                      # :157
            gate_up = F.linear(hidden, self.gate_up[i])                                              # :114
            hidden = F.linear(ops.silu_and_mul(gate_up, c), w[p + "mlp.down_proj.weight"])           # :115-116
        hidden, _ = ops.add_rms_forward(hidden, residual, w["model.norm.weight"], self.eps, c)       # :182
        return hidden

    def compute_logits(self, hidden: torch.Tensor, meta: Meta) -> torch.Tensor:
        """ParallelLMHead.forward at tp=1 (embed_head.py:56-61)."""
        if meta.is_prefill:
            last = (meta.cu_seqlens_q[1:] - 1).long().to(hidden.device)
            hidden = hidden[last].contiguous()
        return F.linear(hidden, self.lm_head)


def load_weights(path: str) -> tuple[dict, dict[str, torch.Tensor]]:
    """config.json + every *.safetensors tensor under `path`."""
    import json
    import os
    from glob import glob
    from safetensors import safe_open
    with open(os.path.join(path, "config.json")) as fh:
        cfg = json.load(fh)
    weights = {}
    for file in sorted(glob(os.path.join(path, "*.safetensors"))):
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():
                weights[name] = f.get_tensor(name)
    return cfg, weights
