"""CPU restatement of the reference's engine loop (TEST INFRASTRUCTURE — see oracle/__init__.py):
scheduler + block manager + runner batch layout + generate(), driving oracle/model.py.

Deliberately naive and close to the reference's semantics, written
as one small state machine over plain dict records. File:line citations are into
GeeeekExplorer/nano-vllm v0.2.0. Validated against the imported reference's own
Scheduler / BlockManager in tests/test_host_logic.py.
"""
from __future__ import annotations

from collections import deque

import numpy as np
import torch
import xxhash

from . import ops
from .model import Meta, OracleQwen3


def chain_hash(tokens: list[int], prefix: int = -1) -> int:
    """BlockManager.compute_hash, engine/block_manager.py:35-41."""
    h = xxhash.xxh64()
    if prefix != -1:
        h.update(prefix.to_bytes(8, "little"))
    h.update(np.array(tokens).tobytes())
    return h.intdigest()


class OracleEngine:
    """generate() over token-id prompts; temperature 0 = argmax of the logits (the parity mode)."""

    def __init__(self, model: OracleQwen3 | None, num_blocks: int, block_size: int = 256, max_num_seqs: int = 512,
                 max_num_batched_tokens: int = 16384, eos: int = -1, seed: int = 0):
        self.model = model
        self.B = block_size
        self.max_num_seqs, self.max_tokens_per_step, self.eos = max_num_seqs, max_num_batched_tokens, eos
        # block pool (block_manager.py:28-33)
        self.blocks = [dict(ref=0, hash=-1, toks=[]) for _ in range(num_blocks)]
        self.free: deque[int] = deque(range(num_blocks))
        self.used: set[int] = set()
        self.by_hash: dict[int, int] = {}
        self.waiting: deque[dict] = deque()
        self.running: deque[dict] = deque()
        self.next_id = 0
        self.gen = torch.Generator().manual_seed(seed)
        self.trace: list[dict] = []          # per-step record for tests (schedule + tables + logits)
        self.keep_logits = False
        self.choose = None                   # tests: callable(step record, logits) -> tokens, replaces the sampler
        if model is not None:
            model.allocate_cache(num_blocks, block_size)

    # ---------------------------------------------------------------- sequences (sequence.py)
    def add(self, tokens: list[int], temperature: float = 0.0, max_tokens: int = 64, ignore_eos: bool = False):
        s = dict(id=self.next_id, toks=list(tokens), n_prompt=len(tokens), cached=0, sched=0, table=[],
                 temp=temperature, max_tokens=max_tokens, ignore_eos=ignore_eos, done=False)
        self.next_id += 1
        self.waiting.append(s)
        return s

    def _nblocks(self, s) -> int:
        return (len(s["toks"]) + self.B - 1) // self.B

    def _block(self, s, i) -> list[int]:
        return s["toks"][i * self.B:(i + 1) * self.B]

    # ---------------------------------------------------------------- block manager
    def _take_block(self) -> int:                                   # block_manager.py:43-51
        b = self.free.popleft()
        rec = self.blocks[b]
        assert rec["ref"] == 0
        if rec["hash"] != -1 and self.by_hash.get(rec["hash"]) == b:
            del self.by_hash[rec["hash"]]
        rec.update(ref=1, hash=-1, toks=[])
        self.used.add(b)
        return b

    def _can_allocate(self, s) -> int:                              # block_manager.py:58-73
        h, hits, need = -1, 0, self._nblocks(s)
        for i in range(self._nblocks(s) - 1):
            toks = self._block(s, i)
            h = chain_hash(toks, h)
            b = self.by_hash.get(h, -1)
            if b == -1 or self.blocks[b]["toks"] != toks:
                break
            hits += 1
            if b in self.used:
                need -= 1
        return hits if len(self.free) >= need else -1

    def _allocate(self, s, hits: int):                              # block_manager.py:75-92
        h = -1
        for i in range(hits):
            h = chain_hash(self._block(s, i), h)
            b = self.by_hash[h]
            if b in self.used:
                self.blocks[b]["ref"] += 1
            else:
                self.blocks[b]["ref"] = 1
                self.free.remove(b)
                self.used.add(b)
            s["table"].append(b)
        for _ in range(hits, self._nblocks(s)):
            s["table"].append(self._take_block())
        s["cached"] = hits * self.B

    def _release(self, s):                                          # block_manager.py:94-101
        for b in reversed(s["table"]):
            rec = self.blocks[b]
            rec["ref"] -= 1
            if rec["ref"] == 0:
                self.used.remove(b)
                self.free.append(b)
        s["cached"] = 0
        s["table"] = []

    def _hash_new_blocks(self, s):                                  # block_manager.py:110-120
        lo, hi = s["cached"] // self.B, (s["cached"] + s["sched"]) // self.B
        if lo == hi:
            return
        h = self.blocks[s["table"][lo - 1]]["hash"] if lo > 0 else -1
        for i in range(lo, hi):
            toks = self._block(s, i)
            h = chain_hash(toks, h)
            self.blocks[s["table"][i]].update(hash=h, toks=toks)
            self.by_hash[h] = s["table"][i]

    # ---------------------------------------------------------------- scheduler (scheduler.py:25-92)
    def schedule(self) -> tuple[list[dict], bool]:
        batch, used_tokens = [], 0
        while self.waiting and len(batch) < self.max_num_seqs:     # :30
            s = self.waiting[0]
            room = self.max_tokens_per_step - used_tokens
            if room == 0:
                break
            if not s["table"]:
                hits = self._can_allocate(s)
                if hits == -1:
                    break
                todo = len(s["toks"]) - hits * self.B
            else:
                todo = len(s["toks"]) - s["cached"]
            if room < todo and batch:                              # :42
                break
            if not s["table"]:
                self._allocate(s, hits)
            s["sched"] = min(todo, room)
            used_tokens += s["sched"]
            if s["cached"] + s["sched"] == len(s["toks"]):         # :48-51
                self.waiting.popleft()
                self.running.append(s)
            batch.append(s)
        if batch:
            return batch, True
        while self.running and len(batch) < self.max_num_seqs:     # :58
            s = self.running.popleft()
            ok = True
            while len(self.free) < (1 if len(s["toks"]) % self.B == 1 else 0):   # can_append, bm:103-104
                if self.running:
                    self._preempt(self.running.pop())
                else:
                    self._preempt(s)
                    ok = False
                    break
            if ok:
                s["sched"] = 1
                if len(s["toks"]) % self.B == 1:                   # may_append, bm:106-108
                    s["table"].append(self._take_block())
                batch.append(s)
        assert batch
        self.running.extendleft(reversed(batch))                   # :72
        return batch, False

    def _preempt(self, s):                                          # :75-79
        self._release(s)
        self.waiting.appendleft(s)

    def postprocess(self, batch, tokens, is_prefill):               # :81-92
        for s, tok in zip(batch, tokens):
            self._hash_new_blocks(s)
            s["cached"] += s["sched"]
            s["sched"] = 0
            if is_prefill and s["cached"] < len(s["toks"]):
                continue
            s["toks"].append(int(tok))
            n_out = len(s["toks"]) - s["n_prompt"]
            if (not s["ignore_eos"] and tok == self.eos) or n_out == s["max_tokens"]:
                s["done"] = True
                self._release(s)
                self.running.remove(s)

    # ---------------------------------------------------------------- runner batch layout
    def prepare_prefill(self, batch) -> tuple[torch.Tensor, torch.Tensor, Meta]:   # model_runner.py:129-170
        ids, pos, slots, cu_q, cu_k = [], [], [], [0], [0]
        max_q = max_k = 0
        for s in batch:
            start, end = s["cached"], s["cached"] + s["sched"]
            ids += s["toks"][start:end]
            pos += list(range(start, end))
            cu_q.append(cu_q[-1] + end - start)
            cu_k.append(cu_k[-1] + end)
            max_q, max_k = max(max_q, end - start), max(max_k, end)
            for t in range(start, end):                             # :151-161, token by token
                slots.append(s["table"][t // self.B] * self.B + t % self.B)
        tables = None
        if cu_k[-1] > cu_q[-1]:                                     # :162-163
            tables = self._tables(batch)
        meta = Meta(True, torch.tensor(cu_q, dtype=torch.int32), torch.tensor(cu_k, dtype=torch.int32), max_q, max_k,
                    torch.tensor(slots, dtype=torch.int32), None, tables)
        return torch.tensor(ids, dtype=torch.int64), torch.tensor(pos, dtype=torch.int64), meta

    def prepare_decode(self, batch) -> tuple[torch.Tensor, torch.Tensor, Meta]:    # model_runner.py:172-188
        ids = [s["toks"][-1] for s in batch]
        pos = [len(s["toks"]) - 1 for s in batch]
        ctx = [len(s["toks"]) for s in batch]
        slots = [s["table"][-1] * self.B + (len(s["toks"]) - 1) % self.B for s in batch]
        meta = Meta(False, slot_mapping=torch.tensor(slots, dtype=torch.int32),
                    context_lens=torch.tensor(ctx, dtype=torch.int32), block_tables=self._tables(batch))
        return torch.tensor(ids, dtype=torch.int64), torch.tensor(pos, dtype=torch.int64), meta

    def _tables(self, batch) -> torch.Tensor:                       # model_runner.py:123-127
        width = max(len(s["table"]) for s in batch)
        return torch.tensor([s["table"] + [-1] * (width - len(s["table"])) for s in batch], dtype=torch.int32)

    # ---------------------------------------------------------------- step / generate
    @torch.inference_mode()
    def step(self, forced_tokens: list[int] | None = None):
        batch, is_prefill = self.schedule()
        ids, pos, meta = self.prepare_prefill(batch) if is_prefill else self.prepare_decode(batch)
        rec = dict(is_prefill=is_prefill, seq_ids=[s["id"] for s in batch], sched=[s["sched"] for s in batch],
                   cached=[s["cached"] for s in batch], tables=[list(s["table"]) for s in batch],
                   input_ids=ids.tolist(), positions=pos.tolist(), slots=meta.slot_mapping.tolist())
        if self.model is not None:
            logits = self.model.compute_logits(self.model.forward(ids, pos, meta), meta)   # model_runner.py:198
            logits = logits.cpu()                                   # (a device-resident model: judge on the host)
            temps = torch.tensor([s["temp"] for s in batch], dtype=torch.float32)
            greedy = ops.greedy(logits)
            if bool((temps > 0).any()):
                sampled = ops.sampler_forward(logits, temps.clamp_min(1e-10), self.gen)       # sampler.py:8-12
                tokens = torch.where(temps > 0, sampled, greedy).tolist()
            else:
                tokens = greedy.tolist()
            if self.choose is not None:
                tokens = self.choose(rec, logits.float())
            if self.keep_logits:
                rec["logits"] = logits.float()
            top2 = logits.float().topk(2, dim=-1).values
            rec["margin"] = (top2[:, 0] - top2[:, 1]).tolist()
        else:
            tokens = [0] * len(batch)
        if forced_tokens is not None:                               # teacher forcing for parity runs
            tokens = forced_tokens
        rec["tokens"] = list(tokens)
        self.trace.append(rec)
        self.postprocess(batch, tokens, is_prefill)
        return batch, is_prefill

    def generate(self, prompts: list[list[int]], temperature=0.0, max_tokens=64, ignore_eos=False) -> list[list[int]]:
        n = len(prompts)
        temps = temperature if isinstance(temperature, list) else [temperature] * n
        mts = max_tokens if isinstance(max_tokens, list) else [max_tokens] * n
        seqs = [self.add(p, t, m, ignore_eos) for p, t, m in zip(prompts, temps, mts)]
        while self.waiting or self.running:
            self.step()
        return [s["toks"][s["n_prompt"]:] for s in seqs]
