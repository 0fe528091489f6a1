"""Attention-metadata side channel between the runner and the layers.

Plays the role of the reference's `utils/context.py:6-27` — the op-level drop-in boundary:
the runner publishes per-step metadata with `set_context(...)`, `Attention.forward` and
`ParallelLMHead.forward` read it with `get_context()`, and `reset_context()` clears it. Field
names and the positional order of `set_context` follow the reference so reference-style
callers work unchanged; four extra fields carry what the HIP decode kernel needs (its split-KV
workspace, the static `max_context` bound that makes the launch hipGraph-safe, the per-step work
plan shared by every layer's launch — ops.decode_plan — and whether that plan carries a shared prefix).
"""
from __future__ import annotations

_FIELDS = ("is_prefill", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "slot_mapping",
           "context_lens", "block_tables", "decode_workspace", "max_context", "decode_plan", "shared_prefix")
_DEFAULTS = (False, None, None, 0, 0, None, None, None, None, 0, None, False)


class Context:
    """Plain slotted record; tensors are device tensors (int32 unless noted)."""
    __slots__ = _FIELDS

    def __init__(self, *args, **kwargs):
        values = dict(zip(_FIELDS, _DEFAULTS))
        values.update(zip(_FIELDS, args))
        values.update(kwargs)
        for name in _FIELDS:
            object.__setattr__(self, name, values[name])

    def __repr__(self):
        return "Context(" + ", ".join(f"{n}={getattr(self, n)!r}" for n in _FIELDS) + ")"


class _Holder:
    current = Context()


def get_context() -> Context:
    return _Holder.current


def set_context(*args, **kwargs) -> None:
    """set_context(is_prefill, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
    slot_mapping, context_lens, block_tables[, decode_workspace, max_context, decode_plan, shared_prefix])"""
    _Holder.current = Context(*args, **kwargs)


def reset_context() -> None:
    _Holder.current = Context()
