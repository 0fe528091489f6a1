"""Weight loading (HF safetensors -> TP-sharded parameters) and synthetic checkpoints.

`load_model` plays the role of the reference's utils/loader.py:12-28: every `*.safetensors`
tensor is routed by name — through the model's `packed_modules_mapping` for the fused
qkv / gate_up projections — to the owning parameter's `weight_loader(param, tensor[, shard])`.
Unlike the reference (which silently leaves `torch.empty` weights when no file matches), a
model directory without safetensors is an error unless `dummy_weights=True` was requested.

There are no real checkpoints, tokenizers or network in this environment, so
this module can also *write* a synthetic checkpoint directory with the published Qwen3 shapes,
seeded random bf16 weights and an offline byte-level BPE tokenizer (`write_synthetic_checkpoint`),
or fill a model with seeded random weights directly on the GPU (`init_dummy_weights`).
"""
from __future__ import annotations

import json
import os
import zlib
from glob import glob

import torch
from torch import nn

QWEN3_SHAPES = {
    # name: hidden, intermediate, layers, heads, kv_heads, tie_word_embeddings
    "qwen3-0.6b": (1024, 3072, 28, 16, 8, True),
    "qwen3-1.7b": (2048, 6144, 28, 16, 8, True),
    "qwen3-4b": (2560, 9728, 36, 32, 8, True),
    "qwen3-8b": (4096, 12288, 36, 32, 8, False),
    "qwen3-14b": (5120, 17408, 40, 40, 8, False),
    "qwen3-32b": (5120, 25600, 64, 64, 8, False),
    # tiny shapes for tests (same architecture, head_dim 128)
    "qwen3-tiny": (256, 512, 2, 4, 2, True),
    "qwen3-tiny-untied": (256, 512, 3, 8, 2, False),     # G = 4, separate lm_head (8B-like)
    "qwen3-tiny-g8": (512, 768, 2, 8, 1, False),          # G = 8, one kv head (32B / TP=8 per-rank shape)
    "qwen3-tiny-kv8": (512, 1024, 2, 16, 8, False),       # 8 kv heads: shards down to 1 kv head per rank at TP = 8
    "qwen3-tiny-g5": (512, 768, 2, 10, 2, False),         # G = 5 (Qwen3-14B's 40 / 8): a group size that does not divide 16
    # two full-width layers of the large models (tests: the layer code paths real 8B / 32B widths take)
    "qwen3-8b-2l": (4096, 12288, 2, 32, 8, False),
    "qwen3-14b-2l": (5120, 17408, 2, 40, 8, False),
    "qwen3-32b-2l": (5120, 25600, 2, 64, 8, False),
    # what ONE rank of Qwen3-32B at tensor_parallel_size = 8 holds (models/qwen3.py:29-38, layers/linear.py:54-156: 8 query
    # heads, 1 kv head, intermediate 25600 / 8; with vocab_size 151936 / 8 also its embedding / lm_head shard): run as a
    # TP = 1 engine it does a rank's work minus the collectives — bench.py's `tp8_rank_shape` upper bound
    "qwen3-32b-tp8rank": (5120, 3200, 64, 8, 1, False),
    "qwen3-32b-tp4rank": (5120, 6400, 64, 16, 2, False),     # ... and of tensor_parallel_size = 4 (BASELINE config 4)
}


def qwen3_config_dict(name: str, vocab_size: int = 151936, max_position_embeddings: int = 40960) -> dict:
    hidden, inter, layers, heads, kv, tie = QWEN3_SHAPES[name.lower()]
    return {
        "architectures": ["Qwen3ForCausalLM"], "model_type": "qwen3", "hidden_size": hidden,
        "intermediate_size": inter, "num_hidden_layers": layers, "num_attention_heads": heads,
        "num_key_value_heads": kv, "head_dim": 128, "vocab_size": vocab_size, "tie_word_embeddings": tie,
        "rope_theta": 1000000.0, "rms_norm_eps": 1e-6, "max_position_embeddings": max_position_embeddings,
        "torch_dtype": "bfloat16", "attention_bias": False, "hidden_act": "silu", "bos_token_id": 0,
        "eos_token_id": 1,
    }


def parameter_shapes(cfg: dict) -> dict[str, tuple]:
    """HF checkpoint tensor names -> shapes for a Qwen3 config dict."""
    h, inter, d = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"]
    nh, nkv, v = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["vocab_size"]
    out = {"model.embed_tokens.weight": (v, h), "model.norm.weight": (h,)}
    if not cfg["tie_word_embeddings"]:
        out["lm_head.weight"] = (v, h)
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        out.update({
            p + "input_layernorm.weight": (h,), p + "post_attention_layernorm.weight": (h,),
            p + "self_attn.q_proj.weight": (nh * d, h), p + "self_attn.k_proj.weight": (nkv * d, h),
            p + "self_attn.v_proj.weight": (nkv * d, h), p + "self_attn.o_proj.weight": (h, nh * d),
            p + "self_attn.q_norm.weight": (d,), p + "self_attn.k_norm.weight": (d,),
            p + "mlp.gate_proj.weight": (inter, h), p + "mlp.up_proj.weight": (inter, h),
            p + "mlp.down_proj.weight": (h, inter),
        })
    return out


def synth_tensor(name: str, shape: tuple, seed: int, device="cpu") -> torch.Tensor:
    """Seeded random weight: matrices N(0, 0.02^2), norm weights 1 + 0.1*N(0,1); bf16.
    The generator is keyed by (seed, crc32(name)) so any rank can regenerate any tensor."""
    gen = torch.Generator(device=device)
    gen.manual_seed((seed << 32) ^ zlib.crc32(name.encode()))
    x = torch.randn(shape, generator=gen, dtype=torch.float32, device=device)
    x = 1.0 + 0.1 * x if len(shape) == 1 else 0.02 * x
    return x.to(torch.bfloat16)


def build_offline_tokenizer(path: str) -> None:
    """Byte-level BPE tokenizer trained on a few built-in sentences (no network, no real vocab).
    eos = <|im_end|> (Qwen's chat terminator), pad = <|endoftext|>."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=320, special_tokens=["<|endoftext|>", "<|im_end|>", "<|im_start|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    corpus = ["introduce yourself", "list all prime numbers within 100", "Benchmark: ", "hello world",
              "the quick brown fox jumps over the lazy dog", "user assistant system"] * 4
    tok.train_from_iterator(corpus, trainer)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="<|endoftext|>")
    fast.chat_template = ("{% for m in messages %}<|im_start|>{{ m['role'] }}\n{{ m['content'] }}<|im_end|>\n"
                          "{% endfor %}{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")
    fast.save_pretrained(path)


def write_synthetic_checkpoint(path: str, name: str = "qwen3-0.6b", seed: int = 0, with_weights: bool = True,
                               vocab_size: int = 151936, max_position_embeddings: int = 40960) -> str:
    """Create `path` with config.json, tokenizer files and (optionally) model.safetensors."""
    os.makedirs(path, exist_ok=True)
    cfg = qwen3_config_dict(name, vocab_size, max_position_embeddings)
    with open(os.path.join(path, "config.json"), "w") as fh:
        json.dump(cfg, fh, indent=1)
    if not os.path.exists(os.path.join(path, "tokenizer.json")):
        build_offline_tokenizer(path)
    st = os.path.join(path, "model.safetensors")
    if with_weights and not os.path.exists(st):
        from safetensors.torch import save_file
        tensors = {n: synth_tensor(n, s, seed) for n, s in parameter_shapes(cfg).items()}
        save_file(tensors, st)
    return path


# ------------------------------------------------------------------------------------------------
def _route(model: nn.Module, weight_name: str):
    """checkpoint tensor name -> (parameter, shard_id | None)."""
    for frag, (fused, shard) in getattr(model, "packed_modules_mapping", {}).items():
        if frag in weight_name:
            return model.get_parameter(weight_name.replace(frag, fused)), shard
    return model.get_parameter(weight_name), None


def _assign(param: nn.Parameter, tensor: torch.Tensor, shard) -> None:
    loader = getattr(param, "weight_loader", None)
    if loader is None:
        param.data.copy_(tensor)
    elif shard is None:
        loader(param, tensor)
    else:
        loader(param, tensor, shard)


def load_model(model: nn.Module, path: str) -> int:
    """Load every tensor of every *.safetensors under `path`. Returns the number loaded."""
    from safetensors import safe_open
    files = sorted(glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path} (pass dummy_weights=True for random weights)")
    n = 0
    tied = getattr(model, "geo", {}).get("tie", False)
    for file in files:
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():
                if tied and name == "lm_head.weight":
                    continue
                param, shard = _route(model, name)
                _assign(param, f.get_tensor(name), shard)
                n += 1
    return n


def init_dummy_weights(model: nn.Module, hf_config, seed: int = 0) -> None:
    """Seeded random weights generated tensor-by-tensor on the GPU and sharded exactly like a
    real checkpoint would be (every TP rank regenerates the same full tensor)."""
    cfg = {
        "hidden_size": hf_config.hidden_size, "intermediate_size": hf_config.intermediate_size, "head_dim": 128,
        "num_attention_heads": hf_config.num_attention_heads, "num_key_value_heads": hf_config.num_key_value_heads,
        "vocab_size": hf_config.vocab_size, "num_hidden_layers": hf_config.num_hidden_layers,
        "tie_word_embeddings": bool(getattr(hf_config, "tie_word_embeddings", False)),
    }
    device = next(model.parameters()).device
    for name, shape in parameter_shapes(cfg).items():
        param, shard = _route(model, name)
        _assign(param, synth_tensor(name, shape, seed, device=device), shard)
