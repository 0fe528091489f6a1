"""`nanovllm` — the reference's import name, served by nano_vllm_amd (MI355X-native).

`from nanovllm import LLM, SamplingParams` and the op-level module paths of the reference
(`nanovllm.layers.attention`, `nanovllm.utils.context`, ...) resolve to the gfx950
implementations, so the reference's bench.py / example.py and model code run unchanged.
"""
import importlib
import sys
import types

from nano_vllm_amd import Config, SamplingParams  # noqa: F401

# reference module path -> (our module, names it must export)
_ALIASES = {
    "nanovllm.llm": ("nano_vllm_amd.engine.core", ["LLM"]),
    "nanovllm.config": ("nano_vllm_amd.api", ["Config"]),
    "nanovllm.sampling_params": ("nano_vllm_amd.api", ["SamplingParams"]),
    "nanovllm.engine.llm_engine": ("nano_vllm_amd.engine.core", ["LLMEngine"]),
    "nanovllm.engine.scheduler": ("nano_vllm_amd.engine.sched", ["Scheduler"]),
    "nanovllm.engine.block_manager": ("nano_vllm_amd.engine.kv_blocks", ["BlockManager"]),
    "nanovllm.engine.sequence": ("nano_vllm_amd.engine.seq", ["Sequence", "SequenceStatus"]),
    "nanovllm.engine.model_runner": ("nano_vllm_amd.engine.runner", ["ModelRunner"]),
    "nanovllm.models.qwen3": ("nano_vllm_amd.qwen3", ["Qwen3ForCausalLM"]),
    "nanovllm.layers.attention": ("nano_vllm_amd.layers", ["Attention"]),
    "nanovllm.layers.layernorm": ("nano_vllm_amd.layers", ["RMSNorm"]),
    "nanovllm.layers.rotary_embedding": ("nano_vllm_amd.layers", ["RotaryEmbedding", "get_rope"]),
    "nanovllm.layers.activation": ("nano_vllm_amd.layers", ["SiluAndMul"]),
    "nanovllm.layers.sampler": ("nano_vllm_amd.layers", ["Sampler"]),
    "nanovllm.layers.linear": ("nano_vllm_amd.layers", ["ReplicatedLinear", "ColumnParallelLinear",
                                                        "MergedColumnParallelLinear", "QKVParallelLinear",
                                                        "RowParallelLinear"]),
    "nanovllm.layers.embed_head": ("nano_vllm_amd.layers", ["VocabParallelEmbedding", "ParallelLMHead"]),
    "nanovllm.utils.context": ("nano_vllm_amd.attn_meta", ["Context", "get_context", "set_context", "reset_context"]),
    "nanovllm.utils.loader": ("nano_vllm_amd.weights", ["load_model"]),
}


class _AliasFinder:
    """Resolve `nanovllm.<reference path>` imports lazily to the nano_vllm_amd modules."""

    @staticmethod
    def find_spec(fullname, path=None, target=None):
        from importlib.machinery import ModuleSpec
        if fullname in _ALIASES or fullname in ("nanovllm.engine", "nanovllm.layers", "nanovllm.models",
                                                "nanovllm.utils"):
            return ModuleSpec(fullname, _AliasFinder, is_package=fullname not in _ALIASES)
        return None

    @staticmethod
    def create_module(spec):
        if spec.name in _ALIASES:
            return importlib.import_module(_ALIASES[spec.name][0])
        mod = types.ModuleType(spec.name)
        mod.__path__ = []
        return mod

    @staticmethod
    def exec_module(module):
        return None


sys.meta_path.append(_AliasFinder)


def __getattr__(name):
    if name in ("LLM", "LLMEngine"):
        from nano_vllm_amd.engine import core
        return getattr(core, name)
    raise AttributeError(f"module 'nanovllm' has no attribute {name!r}")
