"""Paged-KV block accounting with ref-counted prefix caching — array-based.

Same observable behaviour as the reference's `BlockManager` (nano-vllm
engine/block_manager.py:26-120): identical allocation ORDER (FIFO free list), identical hash
chain (xxh64 over an 8-byte LE prefix hash + the block's int64 token bytes, :35-41), identical
reuse / revival / eviction rules — verified against the imported reference on random traces in
tests/test_host_logic.py (golden traces recorded from the imported reference) and
tests/test_oracle_vs_reference.py. What differs is the data structure, sized for the
~10^4-10^5 blocks that 288 GB of HBM gives one MI355X:
  * no per-block Python object: ref counts / hashes / token fingerprints live in flat lists;
  * the free list is an intrusive doubly-linked list over block ids, so reviving a cached block
    out of the middle of the free list is O(1) (the reference's deque.remove is O(n), :87);
  * can_allocate() hands its hash chain to the allocate() that follows (the reference re-hashes).
"""
from __future__ import annotations

import numpy as np
import xxhash

from .seq import Sequence

_NIL = -1


class BlockManager:

    def __init__(self, num_blocks: int, block_size: int):
        assert num_blocks > 0
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.ref_count = [0] * num_blocks
        self.block_hash = [-1] * num_blocks
        self.block_tokens: list[bytes | None] = [None] * num_blocks   # int64 bytes of the hashed tokens
        self.hash_to_block_id: dict[int, int] = {}
        # intrusive FIFO free list, initially 0,1,2,...
        self._next = list(range(1, num_blocks)) + [_NIL]
        self._prev = [_NIL] + list(range(0, num_blocks - 1))
        self._head, self._tail = 0, num_blocks - 1
        self._in_free = [True] * num_blocks
        self.num_free = num_blocks
        self._probe: tuple | None = None
        self._table_gen = 0           # allocation counter: every allocate() stamps the sequence (see allocate)

    # --- hashing (block_manager.py:35-41) -----------------------------------------------------
    @staticmethod
    def _token_bytes(token_ids) -> bytes:
        return np.asarray(token_ids, dtype=np.int64).tobytes()

    @classmethod
    def compute_hash(cls, token_ids, prefix: int = -1) -> int:
        h = xxhash.xxh64()
        if prefix != -1:
            h.update(prefix.to_bytes(8, "little"))
        h.update(token_ids if isinstance(token_ids, bytes) else cls._token_bytes(token_ids))
        return h.intdigest()

    # --- free list ------------------------------------------------------------------------------
    def _free_unlink(self, b: int) -> None:
        p, n = self._prev[b], self._next[b]
        if p != _NIL:
            self._next[p] = n
        else:
            self._head = n
        if n != _NIL:
            self._prev[n] = p
        else:
            self._tail = p
        self._in_free[b] = False
        self.num_free -= 1

    def _free_append(self, b: int) -> None:
        self._prev[b], self._next[b] = self._tail, _NIL
        if self._tail != _NIL:
            self._next[self._tail] = b
        else:
            self._head = b
        self._tail = b
        self._in_free[b] = True
        self.num_free += 1

    @property
    def free_block_ids(self) -> list[int]:
        out, b = [], self._head
        while b != _NIL:
            out.append(b)
            b = self._next[b]
        return out

    @property
    def used_block_ids(self) -> set[int]:
        return {b for b in range(self.num_blocks) if self.ref_count[b] > 0}

    # --- allocation (block_manager.py:43-56) ---------------------------------------------------
    def _allocate_block(self) -> int:
        b = self._head
        assert b != _NIL and self.ref_count[b] == 0
        self._free_unlink(b)
        h = self.block_hash[b]
        if h != -1 and self.hash_to_block_id.get(h) == b:   # evict the stale cache entry
            del self.hash_to_block_id[h]
        self.ref_count[b] = 1
        self.block_hash[b] = -1
        self.block_tokens[b] = None
        return b

    def _probe_prefix(self, seq: Sequence):
        """Walk the hash chain over all blocks but the last (block_manager.py:62-70)."""
        hashes, ids = [], []
        h = -1
        bs = self.block_size
        toks = seq.token_ids
        for i in range(seq.num_blocks - 1):
            tb = self._token_bytes(toks[i * bs: (i + 1) * bs])
            h = self.compute_hash(tb, h)
            b = self.hash_to_block_id.get(h, -1)
            if b == -1 or self.block_tokens[b] != tb:
                break
            hashes.append(h)
            ids.append(b)
        return hashes, ids

    def can_allocate(self, seq: Sequence) -> int:
        """Number of leading blocks served from the prefix cache, or -1 if the rest does not fit."""
        hashes, ids = self._probe_prefix(seq)
        need = seq.num_blocks - sum(1 for b in ids if self.ref_count[b] > 0)
        self._probe = (seq.seq_id, seq.num_tokens, ids)
        if self.num_free < need:
            return -1
        return len(ids)

    def allocate(self, seq: Sequence, num_cached_blocks: int) -> None:
        assert not seq.block_table
        if self._probe is not None and self._probe[0] == seq.seq_id and self._probe[1] == seq.num_tokens:
            ids = self._probe[2]
        else:
            ids = self._probe_prefix(seq)[1]
        self._probe = None
        assert num_cached_blocks <= len(ids)
        table = seq.block_table
        for b in ids[:num_cached_blocks]:
            if self.ref_count[b] > 0:
                self.ref_count[b] += 1
            else:                                   # revive a freed-but-still-hashed block (:85-88)
                self.ref_count[b] = 1
                self._free_unlink(b)
            table.append(b)
        for _ in range(num_cached_blocks, seq.num_blocks):
            table.append(self._allocate_block())
        seq.num_cached_tokens = num_cached_blocks * self.block_size
        # A preempted sequence that is allocated again can come back with the same NUMBER of blocks but other
        # block ids; anything that caches per-sequence table rows (engine/runner.py) keys them by this stamp.
        self._table_gen += 1
        seq.table_gen = self._table_gen

    def deallocate(self, seq: Sequence) -> None:
        """Free in reverse so a sequence's prefix blocks are recycled last (:94-101)."""
        for b in reversed(seq.block_table):
            self.ref_count[b] -= 1
            if self.ref_count[b] == 0:
                self._free_append(b)
        seq.num_cached_tokens = 0
        seq.block_table.clear()

    def can_append(self, seq: Sequence) -> bool:
        return self.num_free >= (1 if seq.num_tokens % self.block_size == 1 else 0)

    def may_append(self, seq: Sequence) -> None:
        if seq.num_tokens % self.block_size == 1:
            seq.block_table.append(self._allocate_block())

    def hash_blocks(self, seq: Sequence) -> None:
        """Register every block that became full in this step (:110-120)."""
        bs = self.block_size
        start = seq.num_cached_tokens // bs
        end = (seq.num_cached_tokens + seq.num_scheduled_tokens) // bs
        if start == end:
            return
        table = seq.block_table
        h = self.block_hash[table[start - 1]] if start > 0 else -1
        toks = seq.token_ids
        for i in range(start, end):
            b = table[i]
            tb = self._token_bytes(toks[i * bs: (i + 1) * bs])
            h = self.compute_hash(tb, h)
            self.block_hash[b] = h
            self.block_tokens[b] = tb
            self.hash_to_block_id[h] = b
