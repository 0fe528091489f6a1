"""Qwen3 decoder built on the libnvl kernels — the only model family the reference ships
(nano-vllm models/qwen3.py:14-216). Same module tree and parameter names as the reference /
HF checkpoints (so `packed_modules_mapping`-style loading works), but the per-layer dataflow
is arranged for MI355X:

   add+RMSNorm -> qkv GEMM -> [q/k-norm + RoPE + KV-cache store: ONE launch]
               -> paged decode attention | MFMA prefill attention -> o GEMM (+all-reduce)
   add+RMSNorm -> gate_up GEMM -> SiLU*mul -> down GEMM (+all-reduce)

i.e. 10 launches per layer instead of the reference's 13 (each dependent kernel boundary is
~1.5 us on this chip even inside a hipGraph — MI355X_MICROARCH.md "boundary"; a small dependent
kernel costs ~5 us whatever it does). On decode-sized batches (<= 256 rows, TP=1, single-K-pass
shapes) the chain is 8 launches:

   add+RMSNorm(fp32 slabs) -> qkv GEMM -> [q/k-norm + RoPE + KV store + paged attention: ONE launch]
   -> split merge -> o GEMM (fp32 split-K slabs) -> add+RMSNorm(slabs) -> gate_up GEMM + SiLU*mul epilogue
   -> down GEMM (slabs)

with the GEMMs on the hand-written skinny kernel (nvl_linear_decode).
"""
from __future__ import annotations

import torch
from torch import nn

from .api import model_geometry
from .attn_meta import get_context
from .layers import (Attention, MergedColumnParallelLinear, ParallelLMHead, QKVParallelLinear, RMSNorm,
                     RowParallelLinear, SiluAndMul, VocabParallelEmbedding, get_rope)
from . import tp


class Qwen3Attention(nn.Module):

    def __init__(self, geo: dict, total_heads: int, total_kv_heads: int, qkv_bias: bool):
        super().__init__()
        self.num_heads, self.num_kv_heads, self.head_dim = geo["heads"], geo["kv_heads"], geo["head_dim"]
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.eps = geo["eps"]
        self.qkv_proj = QKVParallelLinear(geo["hidden"], self.head_dim, total_heads, total_kv_heads, bias=qkv_bias)
        self.o_proj = RowParallelLinear(total_heads * self.head_dim, geo["hidden"], bias=False)
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, geo["max_pos"], geo["rope_theta"])
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_kv_heads)
        self.has_qk_norm = not qkv_bias                      # models/qwen3.py:68-70
        if self.has_qk_norm:
            self.q_norm = RMSNorm(self.head_dim, eps=self.eps)
            self.k_norm = RMSNorm(self.head_dim, eps=self.eps)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor) -> torch.Tensor:
        ctx = get_context()
        # (a step whose plan carries a shared prefix takes bf16 qkv: the attention kernel that also serves the shared-prefix packs
        #  in the SAME launch has no scalar registers left for the slab-sum prologue — attn_decode.hip — and with slabs the pass
        #  would be a launch of its own on a divided grid)
        if self.attn.fuses_decode_step(ctx) and not ctx.shared_prefix:
            qkv = self.qkv_proj.forward_for_fused_decode(hidden_states)      # bf16 [N, out], or fp32 split-K slabs
        else:
            qkv = self.qkv_proj(hidden_states)
        qw = self.q_norm.weight if self.has_qk_norm else None
        kw = self.k_norm.weight if self.has_qk_norm else None
        o = self.attn.forward_fused(qkv, positions, qw, kw, self.eps, self.rotary_emb.cos_sin_cache)
        return self.o_proj.forward_decode(o.view(o.shape[0], -1))


class Qwen3MLP(nn.Module):

    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str):
        super().__init__()
        assert hidden_act == "silu"
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=False)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=False)
        self.act_fn = SiluAndMul()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj.forward_decode(self.gate_up_proj.forward_silu(x))


class Qwen3DecoderLayer(nn.Module):

    def __init__(self, config, geo: dict):
        super().__init__()
        self.self_attn = Qwen3Attention(geo, config.num_attention_heads, config.num_key_value_heads,
                                        qkv_bias=getattr(config, "attention_bias", True))     # default as models/qwen3.py:133
        self.mlp = Qwen3MLP(config.hidden_size, config.intermediate_size, config.hidden_act)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, residual):
        if residual is None:                                 # first layer: residual = embeddings
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class Qwen3Model(nn.Module):

    def __init__(self, config, geo: dict):
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Qwen3DecoderLayer(config, geo) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class Qwen3ForCausalLM(nn.Module):
    # checkpoint-name fragment -> (fused parameter fragment, shard id)   (models/qwen3.py:187-193)
    packed_modules_mapping = {
        "q_proj": ("qkv_proj", "q"),
        "k_proj": ("qkv_proj", "k"),
        "v_proj": ("qkv_proj", "v"),
        "gate_proj": ("gate_up_proj", 0),
        "up_proj": ("gate_up_proj", 1),
    }

    def __init__(self, config):
        super().__init__()
        _, size = tp.world()
        self.geo = model_geometry(config, size)
        self.model = Qwen3Model(config, self.geo)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.geo["tie"]:
            self.lm_head.weight.data = self.model.embed_tokens.weight.data

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.lm_head(hidden_states)

    def compute_logits_shard(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """This rank's vocabulary slice of the logits (TP > 1: sampled shard-wise, never gathered)."""
        return self.lm_head.forward_shard(hidden_states)
