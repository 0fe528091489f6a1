"""Continuous-batching scheduler.

Policy identical to the reference (nano-vllm engine/scheduler.py:25-92; spec in SURVEY.md
Appendix A.1): a step is either all-prefill or all-decode; prefill has absolute priority and
admits head-of-line sequences FIFO under `max_num_seqs` / `max_num_batched_tokens`, chunking
only the first sequence of a batch; decode preempts from the tail (recompute) when the block
pool is exhausted. Equivalence with the imported reference is tested on random traces
(tests/test_host_logic.py: golden traces recorded from the imported reference + a live-reference test) — outputs and prefix-cache reuse depend on it.

Host-path changes that do not alter behaviour: finished sequences leave `running` through an
identity filter instead of deque.remove per sequence, and the running queue is a plain list
rotated in place.
"""
from __future__ import annotations

from collections import deque

from .kv_blocks import BlockManager
from .seq import Sequence, SequenceStatus


class Scheduler:

    def __init__(self, config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_size = config.kvcache_block_size
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    # ------------------------------------------------------------------------------------------
    def schedule(self) -> tuple[list[Sequence], bool]:
        batch = self._schedule_prefill()
        if batch:
            return batch, True
        return self._schedule_decode(), False

    def _schedule_prefill(self) -> list[Sequence]:
        bm, bs = self.block_manager, self.block_size
        batch: list[Sequence] = []
        budget = self.max_num_batched_tokens
        waiting, running = self.waiting, self.running
        while waiting and len(batch) < self.max_num_seqs and budget > 0:
            seq = waiting[0]
            fresh = not seq.block_table
            if fresh:
                cached_blocks = bm.can_allocate(seq)
                if cached_blocks < 0:
                    break                                    # pool exhausted: head-of-line blocks
                todo = seq.num_tokens - cached_blocks * bs
            else:                                            # continuation of a chunked prefill
                todo = seq.num_tokens - seq.num_cached_tokens
            if todo > budget and batch:
                break                                        # only the first sequence may be chunked
            if fresh:
                bm.allocate(seq, cached_blocks)              # blocks for the WHOLE sequence
            seq.num_scheduled_tokens = min(todo, budget)
            budget -= seq.num_scheduled_tokens
            if seq.num_cached_tokens + seq.num_scheduled_tokens == seq.num_tokens:
                seq.status = SequenceStatus.RUNNING
                waiting.popleft()
                running.append(seq)
            batch.append(seq)
        return batch

    def _schedule_decode(self) -> list[Sequence]:
        bm = self.block_manager
        running = self.running
        batch: list[Sequence] = []
        while running and len(batch) < self.max_num_seqs:
            seq = running.popleft()
            ok = True
            while not bm.can_append(seq):
                if running:
                    self.preempt(running.pop())              # evict from the tail
                else:
                    self.preempt(seq)                        # nothing left to evict but itself
                    ok = False
                    break
            if ok:
                seq.num_scheduled_tokens = 1
                seq.is_prefill = False
                bm.may_append(seq)
                batch.append(seq)
        assert batch
        running.extendleft(reversed(batch))                  # keep original order at the front
        return batch

    def preempt(self, seq: Sequence) -> None:
        seq.status = SequenceStatus.WAITING
        seq.is_prefill = True
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    # --- lookahead form of postprocess for decode steps: `postprocess_early` does everything `postprocess` does that
    # cannot depend on the sampled values — hash -> count -> append a PLACEHOLDER -> finish by max_tokens / deallocate —
    # in the same order, so the block manager goes through the reference's states; `fill_tokens` writes the values
    # once they are on the host and applies the one thing that DOES depend on them: the EOS test. A sequence that
    # turns out to have sampled EOS in step N was optimistically scheduled into step N+1 (already enqueued): it is
    # finished retroactively — its step-N+1 row computes a token nobody reads and stores one K/V row into a block
    # that is free again (harmless: every later owner writes a slot before reading it; hashed prefix blocks are full
    # blocks, which that row never touches). The engine uses the gap to schedule and stage step N+1 while the GPU
    # still runs step N.
    PLACEHOLDER = -1

    @staticmethod
    def can_lookahead(seqs: list[Sequence], is_prefill: bool) -> bool:
        return not is_prefill

    def postprocess_early(self, seqs: list[Sequence]) -> None:
        live = [s for s in seqs if s.status is not SequenceStatus.FINISHED]    # EOS found while this step was in flight
        self.postprocess(live, [self.PLACEHOLDER] * len(live), False, check_eos=False)

    def fill_tokens(self, seqs: list[Sequence], token_ids: list[int]) -> None:
        finished = False
        for seq, token_id in zip(seqs, token_ids):
            if seq.token_ids[-1] != self.PLACEHOLDER:
                continue                          # finished by EOS one step earlier: this row's result is discarded
            seq.token_ids[-1] = token_id          # the placeholder is still the last element: fill precedes the
            seq.last_token = token_id             # next step's early postprocess
            if not seq.ignore_eos and token_id == self.eos and seq.status is not SequenceStatus.FINISHED:
                seq.status = SequenceStatus.FINISHED
                if seq.block_table:               # (a preempted sequence holds no blocks)
                    self.block_manager.deallocate(seq)
                finished = True
        if finished:
            self.running = deque(s for s in self.running if s.status is not SequenceStatus.FINISHED)
            self.waiting = deque(s for s in self.waiting if s.status is not SequenceStatus.FINISHED)

    def postprocess(self, seqs: list[Sequence], token_ids: list[int], is_prefill: bool, check_eos: bool = True) -> None:
        bm = self.block_manager
        finished = False
        for seq, token_id in zip(seqs, token_ids):
            bm.hash_blocks(seq)
            seq.num_cached_tokens += seq.num_scheduled_tokens
            seq.num_scheduled_tokens = 0
            if is_prefill and seq.num_cached_tokens < seq.num_tokens:
                continue                                     # mid-prefill: the sampled token is discarded
            seq.append_token(token_id)
            if (check_eos and not seq.ignore_eos and token_id == self.eos) or \
                    seq.num_completion_tokens == seq.max_tokens:
                seq.status = SequenceStatus.FINISHED
                bm.deallocate(seq)
                finished = True
        if finished:
            self.running = deque(s for s in self.running if s.status is not SequenceStatus.FINISHED)
