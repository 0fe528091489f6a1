"""Per-request state.

Behavioural mirror of the reference's `Sequence` (nano-vllm engine/sequence.py:14-83): same
public attributes and helpers so the scheduler / block-manager contract (and user code that
inspects sequences) is unchanged. Differences are internal: `__slots__` (no per-instance dict
on the 0.5 ms/step host path), and a compact tuple state for the TP control channel that — like
the reference's (sequence.py:72-83) — ships the token list only while the sequence still
needs prefill.
"""
from __future__ import annotations

from enum import Enum, auto
from itertools import count

from ..api import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    __slots__ = ("seq_id", "status", "token_ids", "last_token", "num_tokens", "num_prompt_tokens",
                 "num_cached_tokens", "num_scheduled_tokens", "is_prefill", "block_table", "table_gen",
                 "temperature", "max_tokens", "ignore_eos", "rng_key")

    block_size = 256          # set by the engine from Config.kvcache_block_size
    counter = count()

    def __init__(self, token_ids: list[int], sampling_params: SamplingParams | None = None):
        sp = sampling_params if sampling_params is not None else SamplingParams()
        self.seq_id = next(Sequence.counter)
        # identity of this request in the sampler's counter-based draw (with the token position): the engine sets
        # it to the request's ordinal within THAT engine, so a fresh engine with the same seed reproduces a run
        self.rng_key = self.seq_id
        self.status = SequenceStatus.WAITING
        self.token_ids = list(token_ids)
        self.last_token = token_ids[-1]
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = self.num_tokens
        self.num_cached_tokens = 0
        self.num_scheduled_tokens = 0
        self.is_prefill = True
        self.block_table: list[int] = []
        self.table_gen = 0            # stamped by BlockManager.allocate: distinguishes re-allocations of the same length
        self.temperature = sp.temperature
        self.max_tokens = sp.max_tokens
        self.ignore_eos = sp.ignore_eos

    def __len__(self) -> int:
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def is_finished(self) -> bool:
        return self.status is SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> list[int]:
        return self.token_ids[: self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> list[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_blocks(self) -> int:
        return -(-self.num_tokens // self.block_size)

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def block(self, i: int) -> list[int]:
        assert 0 <= i < self.num_blocks
        bs = self.block_size
        return self.token_ids[i * bs: (i + 1) * bs]

    def append_token(self, token_id: int) -> None:
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    # --- TP control channel (rank 0 -> workers): slim state, tokens only while prefilling -------
    def __getstate__(self):
        payload = self.token_ids if self.is_prefill else self.last_token
        return (self.seq_id, self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens,
                self.num_scheduled_tokens, self.block_table, self.table_gen, payload)

    def __setstate__(self, state):
        (self.seq_id, self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.num_scheduled_tokens,
         self.block_table, self.table_gen, payload) = state
        if isinstance(payload, list):
            self.token_ids, self.last_token = payload, payload[-1]
        else:
            self.token_ids, self.last_token = [], payload
        # fields the workers never read
        self.status, self.is_prefill = SequenceStatus.RUNNING, isinstance(payload, list)
        self.temperature, self.max_tokens, self.ignore_eos = 1.0, 0, False
