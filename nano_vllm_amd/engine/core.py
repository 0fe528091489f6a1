"""The engine loop behind `LLM`: tokenizer + scheduler on rank 0, one ModelRunner per GPU.

Public surface identical to the reference's `LLMEngine` (nano-vllm engine/llm_engine.py:15-90):
`LLMEngine(model, **kwargs)` (unknown kwargs ignored, :18-19), `add_request`, `step` ->
(finished outputs, +prefill tokens | -decode batch), `is_finished`, `generate` (results in
submission order, dicts with "text" and "token_ids"), `exit`.
"""
from __future__ import annotations

import atexit
import os
from dataclasses import fields
from time import perf_counter

from ..api import Config, SamplingParams
from .sched import Scheduler
from .seq import Sequence


def _worker_main(config, rank, env):
    os.environ.update(env)                 # NVL_TP_* switches of the parent (spawn does not copy later changes)
    if os.environ.get("NVL_TP_SHARE_GPU") == "1" and "HSA_CU_MASK" not in os.environ:
        # functional mode, every rank on GPU 0: give each WORKER its own slice of the compute units (read when the HSA
        # runtime starts in this fresh process). Ranks whose spinning collective kernels share CUs starve each other on a
        # time-shared GPU (profiles/r05_tp2_cu_mask_experiment.json: 0 of 16 runs stall with disjoint slices, 4 of 6
        # without); rank 0's runtime is already up and keeps the whole chip.
        # (the CU count comes from the parent, which has a runtime up: NVL_TP_CU_COUNT in `env`; MI355X has 256)
        per = int(os.environ.get("NVL_TP_CU_COUNT", "256")) // config.tensor_parallel_size
        if per >= 1:
            os.environ["HSA_CU_MASK"] = f"0:{rank * per}-{(rank + 1) * per - 1}"
    from .runner import ModelRunner
    ModelRunner(config, rank)              # never returns until "exit" (runner.loop)


def _external_tp_group(size: int) -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() == size


class LLMEngine:

    def __init__(self, model, **kwargs):
        known = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in known})
        self.config = config
        Sequence.block_size = config.kvcache_block_size
        self.ps = []
        if config.tensor_parallel_size > 1 and not _external_tp_group(config.tensor_parallel_size):
            # one process per GPU, spawned here (llm_engine.py:24-30). When the caller already runs one process
            # per GPU under torch.distributed (torchrun: bench.py --tp N), ranks > 0 call `LLMEngine.worker`
            # instead and nothing is spawned.
            import torch.multiprocessing as mp
            ctx = mp.get_context("spawn")
            env = {k: v for k, v in os.environ.items() if k.startswith("NVL_")}
            if env.get("NVL_TP_SHARE_GPU") == "1":
                import torch
                env["NVL_TP_CU_COUNT"] = str(torch.cuda.get_device_properties(0).multi_processor_count)
            for rank in range(1, config.tensor_parallel_size):
                proc = ctx.Process(target=_worker_main, args=(config, rank, env))
                proc.start()
                self.ps.append(proc)
        from .runner import ModelRunner
        self.model_runner = ModelRunner(config, 0)                  # also fills config.num_kvcache_blocks
        from transformers import AutoTokenizer
        self.tokenizer = AutoTokenizer.from_pretrained(config.model, use_fast=True)
        config.eos = self.tokenizer.eos_token_id if self.tokenizer.eos_token_id is not None else -1
        self.scheduler = Scheduler(config)
        # decode lookahead (see _step_lookahead), at any TP degree: workers execute the staging images rank 0
        # posts, and every rank holds the sampled ids on the device. NVL_LOOKAHEAD=0 restores the strictly
        # serial loop for A/B measurements
        self._lookahead = os.environ.get("NVL_LOOKAHEAD", "1") != "0"
        self._unfilled = None              # sequences of an in-flight lookahead step whose token values are pending
        self._requests = 0                 # ordinal of the next request (the sampler's per-sequence key)
        self._exited = False
        atexit.register(self.exit)

    @classmethod
    def worker(cls, model, **kwargs) -> None:
        """Tensor-parallel worker entry point for externally launched ranks (torchrun): call it with the SAME
        arguments rank 0 passes to `LLM(...)` from every process whose torch.distributed rank is > 0. Returns
        when rank 0's engine exits."""
        import torch.distributed as dist
        known = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in known})
        assert dist.is_initialized() and dist.get_world_size() == config.tensor_parallel_size and dist.get_rank() > 0
        Sequence.block_size = config.kvcache_block_size
        from .runner import ModelRunner
        ModelRunner(config, dist.get_rank())

    def exit(self):
        if self._exited:
            return
        self._exited = True
        self.model_runner.call("exit")
        del self.model_runner
        for p in self.ps:
            p.join()

    def _checked_prompt(self, prompt: str | list[int], sampling_params: SamplingParams) -> list[int]:
        """Tokenise and validate one request; no side effects."""
        if isinstance(prompt, str):
            prompt = self.tokenizer.encode(prompt)
        assert len(prompt) > 0, "empty prompt"
        # the staging block tables and the RoPE table are sized by max_model_len: a request that would outgrow
        # them is refused here instead of failing mid-generation (stricter than the reference, which fails later)
        assert len(prompt) + sampling_params.max_tokens <= self.config.max_model_len, (
            f"prompt ({len(prompt)} tokens) + max_tokens ({sampling_params.max_tokens}) exceeds "
            f"max_model_len ({self.config.max_model_len})")
        return prompt

    def _new_sequence(self, prompt: list[int], sampling_params: SamplingParams) -> Sequence:
        seq = Sequence(prompt, sampling_params)
        seq.rng_key = self._requests
        self._requests += 1
        return seq

    def add_request(self, prompt: str | list[int], sampling_params: SamplingParams):
        self.scheduler.add(self._new_sequence(self._checked_prompt(prompt, sampling_params), sampling_params))

    def step(self):
        seqs, is_prefill = self.scheduler.schedule()
        num_tokens = sum(s.num_scheduled_tokens for s in seqs) if is_prefill else -len(seqs)
        token_ids = self.model_runner.call("run", seqs, is_prefill)
        self.scheduler.postprocess(seqs, token_ids, is_prefill)
        outputs = [(s.seq_id, s.completion_token_ids) for s in seqs if s.is_finished]
        return outputs, num_tokens

    def _collect(self):
        """Bring the oldest in-flight decode step's ids to the host and write them into its sequences."""
        seqs = self._unfilled
        self._unfilled = None
        if seqs is None:
            return []
        before = {s.seq_id for s in seqs if s.is_finished and s.token_ids[-1] != self.scheduler.PLACEHOLDER}
        self.scheduler.fill_tokens(seqs, self.model_runner.call("decode_end"))
        # (a sequence that had already finished by EOS in the step before was reported then)
        return [(s.seq_id, s.completion_token_ids) for s in seqs if s.is_finished and s.seq_id not in before]

    def _step_lookahead(self, pending):
        """One engine step for `generate()`. Same results as `step()`; on decode steps the host runs ahead of the
        GPU ("overlap schedule(N+1) with GPU(N)"): step N+1 is ENQUEUED before step N's ids
        have reached the host — its input ids are taken from step N's output on the device (nvl_feed_tokens) —
        then step N is collected, and postprocess (token values filled in afterwards), schedule and staging of
        step N+2 run while the GPU works. The GPU queue never drains between decode steps. With `ignore_eos` the
        scheduler and block manager go through exactly the serial loop's operations; a sequence that samples EOS
        is discovered one step late and finished retroactively (sched.Scheduler.fill_tokens): its own outputs are the
        serial loop's, only its last step's row was computed for nothing. At any temperature every request sees the
        serial loop's random numbers: the sampler's counter-based draw is keyed by (seed, request ordinal, token
        position) — `Sequence.rng_key`, staged per row as `rkey` — not by (step, batch row), so the row a
        retroactively finished sequence still occupies in step N+1 does not shift anybody else's draws
        (tests/test_engine_host.py proves the equality on a stand-in device; on the GPU the logits of the other rows
        can still differ in their last bf16 bit, as between any two batch compositions — tests/test_e2e_gpu.py). Finished sequences of a lookahead step
        are reported by the next call.
        `pending`: None, or (seqs, is_prefill, staged) scheduled by the previous call.
        Returns (finished outputs, num_tokens, pending for the next call)."""
        sched, runner = self.scheduler, self.model_runner
        if pending is None:
            outputs = self._collect()                    # nothing staged: the schedule may need real tokens
            if sched.is_finished():
                return outputs, 0, None
            seqs, is_prefill = sched.schedule()
            staged = False
        else:
            outputs = []
            seqs, is_prefill, staged = pending
        num_tokens = sum(s.num_scheduled_tokens for s in seqs) if is_prefill else -len(seqs)
        nxt = None
        if self._lookahead and sched.can_lookahead(seqs, is_prefill):
            runner.call("decode_begin", seqs, staged)
            outputs += self._collect()                   # the previous decode step, now that the GPU has work queued
            sched.postprocess_early(seqs)
            self._unfilled = seqs
            if not sched.is_finished():
                nseqs, nprefill = sched.schedule()
                if sched.can_lookahead(nseqs, nprefill):
                    runner.call("stage_next_decode", nseqs)
                    nxt = (nseqs, False, True)
                else:
                    nxt = (nseqs, nprefill, False)
        else:
            outputs += self._collect()
            token_ids = runner.call("run", seqs, is_prefill)
            sched.postprocess(seqs, token_ids, is_prefill)
            outputs += [(s.seq_id, s.completion_token_ids) for s in seqs if s.is_finished]
        return outputs, num_tokens, nxt

    def is_finished(self):
        return self.scheduler.is_finished()

    def reset_prefix_cache(self):
        """Forget every cached block (extension; engine must be idle). A second generate() over the same
        prompts otherwise serves their full blocks from the prefix cache — which is the point of the cache,
        but not what a cold-start measurement wants."""
        assert self.scheduler.is_finished(), "reset_prefix_cache() needs an idle engine"
        from .kv_blocks import BlockManager
        bm = self.scheduler.block_manager
        self.scheduler.block_manager = BlockManager(bm.num_blocks, bm.block_size)

    def generate(self, prompts, sampling_params, use_tqdm: bool = True):
        from tqdm.auto import tqdm
        pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True, disable=not use_tqdm)
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        # validate EVERY request before enqueueing ANY: a refused request must not leave its predecessors queued
        checked = [self._checked_prompt(prompt, sp) for prompt, sp in zip(prompts, sampling_params)]
        for prompt, sp in zip(checked, sampling_params):
            self.scheduler.add(self._new_sequence(prompt, sp))
        done: dict[int, list[int]] = {}
        prefill_tps = decode_tps = 0.0
        pending = None                      # batch already scheduled (and maybe staged) by the lookahead
        while not self.is_finished() or pending is not None or self._unfilled is not None:
            t0 = perf_counter()
            finished, num_tokens, pending = self._step_lookahead(pending)
            dt = perf_counter() - t0
            if num_tokens > 0:
                prefill_tps = num_tokens / dt
            else:
                decode_tps = -num_tokens / dt
            if use_tqdm:
                pbar.set_postfix({"Prefill": f"{int(prefill_tps)}tok/s", "Decode": f"{int(decode_tps)}tok/s"})
            for seq_id, toks in finished:
                done[seq_id] = toks
                pbar.update(1)
        pbar.close()
        ordered = [done[k] for k in sorted(done)]
        return [{"text": self.tokenizer.decode(toks), "token_ids": toks} for toks in ordered]


class LLM(LLMEngine):
    """The user-facing name (`from nanovllm import LLM`); behaviour is entirely LLMEngine's."""
