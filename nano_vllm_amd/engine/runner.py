"""One process per GPU: owns the model shard, the paged KV cache, the step staging buffers and
the captured decode hipGraphs.

Role of the reference's `ModelRunner` (nano-vllm engine/model_runner.py:15-257; batch layout
spec in SURVEY.md Appendix A.3), re-designed around the MI355X host/device boundary:

  * ONE pinned staging block + ONE device block with an identical fixed layout hold every
    per-step input (token ids, positions, slot mapping, context lengths, temperatures, RNG
    offset, block tables). A step is: fill the pinned block in place with numpy (no per-step
    tensor allocation, vs the reference's 5-6 `torch.tensor(..., pin_memory=True).cuda()`,
    model_runner.py:126,164-168,182-185,192), one async H2D copy, one graph replay, one D2H of
    the sampled ids.
  * The device block IS the static input of the captured graphs, so eager and graph paths
    share buffers and there are no per-replay staging copies (model_runner.py:204-210).
  * Block-table rows are rewritten only when a row's (sequence, #blocks) changed.
  * The whole decode step — 28 layers, lm_head and the sampler — is one hipGraph per batch
    bucket at TP=1 (the reference captures the layers only and runs lm_head + sampler eagerly).
  * KV cache layout [L, 2, num_blocks, Hkv, block, 128] (layer-major, head-major blocks), zero-initialised;
    sized by the reference's formula (model_runner.py:103-115).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..api import Config, model_geometry
from ..attn_meta import reset_context, set_context
from ..layers import Sampler
from ..qwen3 import Qwen3ForCausalLM
from ..weights import init_dummy_weights, load_model
from .seq import Sequence

_SHM_NAME = "nanovllm_amd"
_DIST_PORT = 2333        # same rendezvous port as the reference (model_runner.py:26)


def _align(n: int, a: int = 16) -> int:
    return (n + a - 1) // a * a


class _Stage:
    """Fixed-layout staging block: pinned host copy + device copy + typed views of both."""

    def __init__(self, fields: list[tuple[str, np.dtype, tuple]], device, host_copies: int = 1):
        off = 0
        layout = []
        for name, dt, shape in fields:
            nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
            layout.append((name, np.dtype(dt), shape, off, nbytes))
            off = _align(off + nbytes)
        self.nbytes = off
        pin = device.type == "cuda"
        # `host_copies` pinned host images (the decode stage keeps two: step N+1 is staged while the upload of
        # step N may still be queued behind the GPU work of step N-1); `np` / `host` always name the current one
        self.hosts = [torch.zeros(off, dtype=torch.uint8, device="cpu", pin_memory=pin) for _ in range(host_copies)]
        self.dev = torch.zeros(off, dtype=torch.uint8, device=device)
        self.nps: list[dict[str, np.ndarray]] = []
        self.t: dict[str, torch.Tensor] = {}
        tdt = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32,
               np.dtype(np.uint64): torch.int64}
        for h in self.hosts:
            hnp = h.numpy()
            self.nps.append({name: hnp[o:o + nb].view(dt).reshape(shape) for name, dt, shape, o, nb in layout})
        for name, dt, shape, o, nb in layout:
            self.t[name] = self.dev[o:o + nb].view(tdt[dt]).view(shape)
        self.uploaded = [torch.cuda.Event() if pin else None for _ in self.hosts]
        self.cur = 0

    @property
    def np(self) -> dict[str, np.ndarray]:
        return self.nps[self.cur]

    @property
    def host(self) -> torch.Tensor:
        return self.hosts[self.cur]

    def flip(self) -> None:
        """Switch to the other pinned image and make sure its last upload has been consumed."""
        self.cur = (self.cur + 1) % len(self.hosts)
        ev = self.uploaded[self.cur]
        if ev is not None:
            ev.synchronize()

    def upload(self, nbytes: int | None = None) -> None:
        n = self.nbytes if nbytes is None else nbytes
        self.dev[:n].copy_(self.host[:n], non_blocking=True)
        if self.uploaded[self.cur] is not None:
            self.uploaded[self.cur].record()


class ModelRunner:

    def __init__(self, config: Config, rank: int, event):
        self.config = config
        hf = config.hf_config
        self.block_size = config.kvcache_block_size
        self.enforce_eager = config.enforce_eager
        self.world_size = config.tensor_parallel_size
        self.rank = rank
        self.event = event
        assert torch.cuda.is_available(), "nano_vllm_amd needs a HIP device (no CPU fallback for the hot path)"
        ops.load_library()

        if self.world_size > 1:
            dist.init_process_group("nccl", f"tcp://127.0.0.1:{_DIST_PORT}", world_size=self.world_size, rank=rank)
        from .. import tp
        tp.init(rank if self.world_size > 1 else 0, self.world_size)
        torch.cuda.set_device(rank)
        self.device = torch.device("cuda", rank)
        self.geo = model_geometry(hf, self.world_size)
        dtype = self.geo["dtype"] or torch.bfloat16
        if isinstance(dtype, str):
            dtype = getattr(torch, dtype)
        assert dtype == torch.bfloat16, f"libnvl kernels are bf16 (config dtype {dtype})"
        prev_dtype = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        torch.set_default_device(self.device)
        try:
            self.model = Qwen3ForCausalLM(hf)
            if config.dummy_weights:
                init_dummy_weights(self.model, hf, config.seed)
            else:
                load_model(self.model, config.model)
            self.sampler = Sampler(seed=config.seed)
            # decode micro-batching (see _forward_decode): second chain's stream, sampler and workspace
            self.microbatches = int(os.environ.get("NVL_MICROBATCHES", "1")) if self.world_size == 1 else 1
            self.side_stream = torch.cuda.Stream(device=self.device) if self.microbatches > 1 else None
            self.sampler_b = Sampler(seed=config.seed + 0x9E3779B9)
            self._alloc_stages()
            self.warmup_model()
            self.allocate_kv_cache()
            self.graphs: dict[int, torch.cuda.CUDAGraph] = {}
            if not self.enforce_eager:
                self.capture_graphs()
        finally:
            torch.set_default_device("cpu")
            torch.set_default_dtype(prev_dtype)

        if self.world_size > 1:
            from multiprocessing.shared_memory import SharedMemory
            if rank == 0:
                try:
                    SharedMemory(name=_SHM_NAME).unlink()       # stale segment from a crashed run
                except FileNotFoundError:
                    pass
                self.shm = SharedMemory(name=_SHM_NAME, create=True, size=2 ** 20)
                dist.barrier()
            else:
                dist.barrier()
                self.shm = SharedMemory(name=_SHM_NAME)
                self.loop()

    # ------------------------------------------------------------------ lifecycle / TP RPC
    def exit(self):
        if self.world_size > 1:
            self.shm.close()
            dist.barrier()
            if self.rank == 0:
                self.shm.unlink()
        self.graphs = {}
        torch.cuda.synchronize()
        if self.world_size > 1:
            dist.destroy_process_group()
            from .. import tp
            tp.init(0, 1)

    def loop(self):
        while True:
            method, args = self._read_shm()
            self.call(method, *args)
            if method == "exit":
                break

    def _read_shm(self):
        self.event.wait()
        n = int.from_bytes(self.shm.buf[0:4], "little")
        method, *args = pickle.loads(self.shm.buf[4:n + 4])
        self.event.clear()
        return method, args

    def _write_shm(self, method, *args):
        data = pickle.dumps([method, *args])
        n = len(data)
        assert n + 4 <= self.shm.size, "step message exceeds the 1 MiB control channel"
        self.shm.buf[0:4] = n.to_bytes(4, "little")
        self.shm.buf[4:n + 4] = data
        for ev in self.event:
            ev.set()

    def call(self, method, *args):
        if self.world_size > 1 and self.rank == 0:
            self._write_shm(method, *args)
        return getattr(self, method)(*args)

    # ------------------------------------------------------------------ buffers
    def _alloc_stages(self):
        cfg = self.config
        self.max_bs = cfg.max_num_seqs
        self.max_blocks = -(-cfg.max_model_len // self.block_size)
        mb, w = self.max_bs, self.max_blocks
        self.dstage = _Stage([
            ("ids", np.int64, (mb,)), ("pos", np.int64, (mb,)), ("rng", np.uint64, (2,)),
            ("slots", np.int32, (mb,)), ("ctx", np.int32, (mb,)), ("temps", np.float32, (mb,)),
            ("src", np.int32, (mb,)), ("bt", np.int32, (mb, w)),
        ], self.device, host_copies=2)
        for image in self.dstage.nps:
            image["slots"][:] = -1
            image["bt"][:] = -1
            image["src"][:] = -1
        self.dstage.upload()
        # per pinned image: (seq id, #blocks) per decode row whose block-table row is current, rows in use
        self._row_keys = [np.full((mb, 2), -1, dtype=np.int64) for _ in self.dstage.nps]
        self._dirty = [0 for _ in self.dstage.nps]
        nt = cfg.max_num_batched_tokens
        ns = min(cfg.max_num_seqs, nt)
        self.pstage = _Stage([
            ("ids", np.int64, (nt,)), ("pos", np.int64, (nt,)), ("rng", np.uint64, (2,)),
            ("slots", np.int32, (nt,)), ("cu_q", np.int32, (ns + 1,)), ("cu_k", np.int32, (ns + 1,)),
            ("temps", np.float32, (ns,)), ("bt", np.int32, (ns, w)),
        ], self.device)
        self.tokens_dev = torch.zeros(max(mb, ns), dtype=torch.int64, device=self.device)
        self.tokens_host = torch.zeros(max(mb, ns), dtype=torch.int64, device="cpu", pin_memory=True)
        self.tokens_host_b = torch.zeros(max(mb, ns), dtype=torch.int64, device="cpu", pin_memory=True)
        ws_bytes = ops.paged_attn_decode_workspace_bytes(mb, self.geo["heads"], cfg.max_model_len)
        # zeroed once: the kernel's arrival counters live in it and are left at zero by every launch
        self.decode_ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=self.device)
        self.decode_ws_b = torch.zeros(ws_bytes, dtype=torch.uint8, device=self.device)
        self.hidden_out = (torch.zeros(mb, self.geo["hidden"], dtype=torch.bfloat16, device=self.device)
                           if self.world_size > 1 else None)
        self.step_count = 0
        self._inflight: list = []            # decode steps enqueued and not yet collected (at most two)
        self._flight_parity = 0
        self._step_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._last_rows: dict = {}

    # ------------------------------------------------------------------ warm-up + KV cache
    def warmup_model(self):
        """One worst-case prefill without a cache, to measure peak activation memory
        (model_runner.py:91-101)."""
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        cfg = self.config
        seq_len = min(cfg.max_num_batched_tokens, cfg.max_model_len)
        num_seqs = min(cfg.max_num_batched_tokens // seq_len, cfg.max_num_seqs)
        seqs = [Sequence([0] * seq_len) for _ in range(num_seqs)]
        for s in seqs:
            s.num_scheduled_tokens = seq_len
        self.run(seqs, True)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def allocate_kv_cache(self):
        cfg, geo = self.config, self.geo
        free, total = torch.cuda.mem_get_info()
        used = total - free
        stats = torch.cuda.memory_stats()
        peak, current = stats["allocated_bytes.all.peak"], stats["allocated_bytes.all.current"]
        block_bytes = 2 * geo["layers"] * self.block_size * geo["kv_heads"] * geo["head_dim"] * 2
        if cfg.num_kvcache_blocks <= 0:
            cfg.num_kvcache_blocks = int(total * cfg.gpu_memory_utilization - used - peak + current) // block_bytes
        assert cfg.num_kvcache_blocks > 0, "no memory left for the KV cache"
        # zero-filled: masked tail rows of a block are multiplied by P == 0 in the decode kernel
        # Layer-major: a layer's K and V pools sit next to each other (the reference's [2, L, ...] puts them
        # 137 GB apart at this pool size, which costs the decode kernel 2.7 % of its bandwidth on MI355X —
        # tools/attn_replay.py --cache-blocks 9377 [--layer-major]). `kv_cache` keeps the [2, L, ...] indexing
        # as a transposed view.
        self._kv_storage = torch.zeros(geo["layers"], 2, cfg.num_kvcache_blocks, geo["kv_heads"], self.block_size,
                                       geo["head_dim"], dtype=torch.bfloat16, device=self.device)
        self.kv_cache = self._kv_storage.transpose(0, 1)
        layer = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = self.kv_cache[0, layer]
                module.v_cache = self.kv_cache[1, layer]
                layer += 1

    # ------------------------------------------------------------------ batch preparation (host)
    def prepare_prefill(self, seqs: list[Sequence]) -> dict:
        """Fill the prefill staging block (semantics of model_runner.py:129-170)."""
        st, bs = self.pstage.np, self.block_size
        n = 0
        cu_q, cu_k = st["cu_q"], st["cu_k"]
        cu_q[0] = cu_k[0] = 0
        max_q = max_k = 0
        have_slots = True
        for i, seq in enumerate(seqs):
            start = seq.num_cached_tokens
            lq = seq.num_scheduled_tokens
            end = start + lq
            st["ids"][n:n + lq] = seq.token_ids[start:end]
            pos = np.arange(start, end, dtype=np.int64)
            st["pos"][n:n + lq] = pos
            cu_q[i + 1] = cu_q[i] + lq
            cu_k[i + 1] = cu_k[i] + end
            max_q, max_k = max(max_q, lq), max(max_k, end)
            if seq.block_table:
                table = np.asarray(seq.block_table, dtype=np.int64)
                st["slots"][n:n + lq] = table[pos // bs] * bs + pos % bs
            else:                                              # warm-up: no cache, nothing to store
                have_slots = False
            st["temps"][i] = seq.temperature
            n += lq
        ns = len(seqs)
        paged = int(cu_k[ns]) > int(cu_q[ns])                  # some K/V must come from the cache
        if paged:
            bt = st["bt"]
            for i, seq in enumerate(seqs):
                t = seq.block_table
                bt[i, :len(t)] = t
                bt[i, len(t):] = -1
        return dict(n=n, ns=ns, max_q=max_q, max_k=max_k, paged=paged, have_slots=have_slots)

    def prepare_decode(self, seqs: list[Sequence], prev_rows: dict | None = None) -> int:
        """Fill the current pinned decode image (semantics of model_runner.py:172-188); returns bs.
        `prev_rows` (lookahead): seq_id -> row of the sequence in the decode step that is still in flight; the
        input ids are then NOT staged — the graph's first node copies them from that step's sampled ids on the
        device (nvl_feed_tokens) — and `src` carries the rows."""
        st, bs = self.dstage.np, self.block_size
        n = len(seqs)
        lens = np.fromiter((s.num_tokens for s in seqs), dtype=np.int64, count=n)
        if prev_rows is None:
            st["ids"][:n] = [s.last_token for s in seqs]
            st["src"][:n] = -1
        else:
            st["src"][:n] = [prev_rows[s.seq_id] for s in seqs]
        st["pos"][:n] = lens - 1
        st["ctx"][:n] = lens
        last_blk = np.fromiter((s.block_table[-1] for s in seqs), dtype=np.int64, count=n)
        st["slots"][:n] = last_blk * bs + (lens - 1) % bs
        st["temps"][:n] = [s.temperature for s in seqs]
        # block tables: rewrite a row only when its (sequence, #blocks) changed in THIS image
        key = self._row_keys[self.dstage.cur]
        ids = np.fromiter((s.seq_id for s in seqs), dtype=np.int64, count=n)
        nblk = np.fromiter((len(s.block_table) for s in seqs), dtype=np.int64, count=n)
        stale = np.nonzero((key[:n, 0] != ids) | (key[:n, 1] != nblk) | (ids < 0))[0]
        bt = st["bt"]
        for i in stale:
            t = seqs[i].block_table
            bt[i, :len(t)] = t
            bt[i, len(t):] = -1
        key[:n, 0], key[:n, 1] = ids, nblk
        # neutralise rows used by a previous, larger batch (graph padding: slot -1, context 0)
        dirty = self._dirty[self.dstage.cur]
        if dirty > n:
            st["slots"][n:dirty] = -1
            st["ctx"][n:dirty] = 0
            key[n:dirty] = -1
        self._dirty[self.dstage.cur] = n
        return n

    # ------------------------------------------------------------------ forward
    def _next_rng(self, st: _Stage) -> None:
        self.step_count += 1
        st.np["rng"][0] = self.step_count

    def _decode_rows(self, r0: int, r1: int, ws, sampler):
        """Decode forward for rows [r0, r1) of the static device buffers, on the current stream."""
        t = self.dstage.t
        set_context(False, slot_mapping=t["slots"][r0:r1], context_lens=t["ctx"][r0:r1],
                    block_tables=t["bt"][r0:r1], decode_workspace=ws, max_context=self.config.max_model_len)
        hidden = self.model(t["ids"][r0:r1], t["pos"][r0:r1])
        if self.world_size == 1:
            logits = self.model.compute_logits(hidden)
            sampler(logits, t["temps"][r0:r1], out=self.tokens_dev[r0:r1], offset_dev=t["rng"][:1])
        else:
            self.hidden_out[r0:r1].copy_(hidden)
        reset_context()

    @torch.inference_mode()
    def _forward_decode(self, bs: int):
        """Decode forward on the static device buffers (captured per bucket, or run eagerly).
        TP=1: layers + lm_head + sampler. TP>1: layers only (the logits gather to rank 0 and the
        sampler run eagerly in `_decode_tail`, as in the reference, model_runner.py:212,218).

        Micro-batching (NVL_MICROBATCHES=2, off by default; TP=1, bs >= 32): sequences are independent, so
        the batch can be cut into two half-batches whose layer chains are forked onto two HIP streams (two
        parallel branches of the same captured hipGraph), hoping that one chain's HBM-bound attention overlaps
        the other's short latency-bound kernels. Measured on MI355X / ROCm 7.2 the branches do not overlap
        usefully (27-30 k vs 31.7 k tok/s) while every small kernel and one pass over the weights are paid
        twice; kept as a tested option (tests/test_e2e_gpu.py) for stacks where graph branches do run
        concurrently."""
        # input ids of sequences that were in the previous decode step come straight from its sampled ids
        # (for the whole batch, BEFORE any chain of this step can overwrite tokens_dev)
        t = self.dstage.t
        ops.feed_tokens(t["ids"][:bs], t["src"][:bs], self.tokens_dev)
        if self.microbatches > 1 and bs >= 32 and self.world_size == 1:
            h = bs // 2
            main = torch.cuda.current_stream()
            side = self.side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._decode_rows(h, bs, self.decode_ws_b, self.sampler_b)
            self._decode_rows(0, h, self.decode_ws, self.sampler)
            main.wait_stream(side)
        else:
            self._decode_rows(0, bs, self.decode_ws, self.sampler)

    @torch.inference_mode()
    def _decode_tail(self, bs: int):
        t = self.dstage.t
        set_context(False)
        logits = self.model.compute_logits(self.hidden_out[:bs])
        if self.rank == 0:
            self.sampler(logits, t["temps"][:bs], out=self.tokens_dev[:bs], offset_dev=t["rng"][:1])
        reset_context()

    @torch.inference_mode()
    def decode_begin(self, seqs: list[Sequence], staged: bool = False) -> int:
        """Enqueue one decode step (H2D of the staging image, graph replay or eager forward, D2H of the
        sampled ids) and return without waiting; up to two steps may be in flight. `staged`: the image was
        already filled by `stage_next_decode` (lookahead)."""
        if not staged:
            self.dstage.flip()
            self.prepare_decode(seqs)
        n = len(seqs)
        self._next_rng(self.dstage)
        self.dstage.upload()
        bucket = next((b for b in self.graph_bs if b >= n), None) if self.graphs else None
        if bucket is not None:
            self.graphs[bucket].replay()
        else:
            self._forward_decode(n)
        if self.world_size > 1:
            self._decode_tail(n)
        host = self.tokens_host if self._flight_parity == 0 else self.tokens_host_b
        done = self._step_done[self._flight_parity]
        if self.rank == 0:
            host[:n].copy_(self.tokens_dev[:n], non_blocking=True)
        done.record()
        self._inflight.append((n, host, done))
        self._flight_parity ^= 1
        self._last_rows = {s.seq_id: i for i, s in enumerate(seqs)}
        return n

    def decode_end(self) -> list[int] | None:
        """Wait for the OLDEST step enqueued by `decode_begin` and return its sampled ids (rank 0)."""
        n, host, done = self._inflight.pop(0)
        done.synchronize()
        return host[:n].tolist() if self.rank == 0 else None

    def stage_next_decode(self, seqs: list[Sequence]) -> None:
        """Lookahead: fill the other pinned image for the NEXT decode step while the step just enqueued runs
        (everything but the input ids, which the graph takes from that step's output on the device)."""
        self.dstage.flip()
        self.prepare_decode(seqs, self._last_rows)

    def _run_decode(self, seqs: list[Sequence]) -> list[int] | None:
        self.decode_begin(seqs)
        return self.decode_end()

    @torch.inference_mode()
    def _run_prefill(self, seqs: list[Sequence]) -> list[int] | None:
        info = self.prepare_prefill(seqs)
        self._next_rng(self.pstage)
        self.pstage.upload()
        t = self.pstage.t
        n, ns = info["n"], info["ns"]
        set_context(True, t["cu_q"][:ns + 1], t["cu_k"][:ns + 1], info["max_q"], info["max_k"],
                    t["slots"][:n] if info["have_slots"] else None, None, t["bt"][:ns] if info["paged"] else None)
        hidden = self.model(t["ids"][:n], t["pos"][:n])
        logits = self.model.compute_logits(hidden)
        reset_context()
        if self.rank != 0:
            return None
        self.sampler(logits, t["temps"][:ns], out=self.tokens_dev[:ns], offset_dev=t["rng"][:1])
        self.tokens_host[:ns].copy_(self.tokens_dev[:ns], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.tokens_host[:ns].tolist()

    def run(self, seqs: list[Sequence], is_prefill: bool) -> list[int] | None:
        return self._run_prefill(seqs) if is_prefill else self._run_decode(seqs)

    # ------------------------------------------------------------------ hipGraph capture
    @torch.inference_mode()
    def capture_graphs(self):
        """One graph per batch bucket, largest first so the pool is sized once
        (buckets as model_runner.py:234)."""
        max_bs = min(self.max_bs, 512)
        self.graph_bs = [b for b in (1, 2, 4, 8) if b <= max_bs] + list(range(16, max_bs + 1, 16))
        pool = None
        # neutral inputs: every row is padding (slot -1, context 0)
        for bs in reversed(self.graph_bs):
            graph = torch.cuda.CUDAGraph()
            self._forward_decode(bs)                    # warm-up (allocator, hipBLASLt heuristics)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool):
                self._forward_decode(bs)
            if pool is None:
                pool = graph.pool()
            self.graphs[bs] = graph
        torch.cuda.synchronize()
        self.graph_pool = pool
