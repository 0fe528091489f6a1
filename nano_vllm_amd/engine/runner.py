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
  * Tensor parallelism (model_runner.py:26,41-89): rank 0 prepares a step ONCE and ships the filled staging
    image — a fixed binary layout, not pickled Sequence objects — to the workers through a ring of slots in
    POSIX shared memory (`_Channel`); workers copy it into their own pinned block, upload and launch. Every
    rank samples its vocabulary shard and the 8-byte-per-row winners are exchanged over xGMI, so every rank
    holds the sampled ids on the device and the decode lookahead works unchanged at TP > 1.
"""
from __future__ import annotations

import os
import pickle
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..api import Config, model_geometry
from ..attn_meta import get_context, reset_context, set_context
from ..layers import Sampler
from ..qwen3 import Qwen3ForCausalLM
from ..weights import init_dummy_weights, load_model
from .seq import Sequence

_SHM_NAME = "nanovllm_amd"
_DIST_PORT = 2333        # same rendezvous port as the reference (model_runner.py:26); NVL_TP_PORT overrides


def _tp_port() -> int:
    return int(os.environ.get("NVL_TP_PORT", os.environ.get("MASTER_PORT", _DIST_PORT)))


class _Channel:
    """Rank 0 -> workers control channel: a ring of message slots in POSIX shared memory.

    Role of the reference's SharedMemory + Event + pickle protocol (model_runner.py:41-48,61-89), with two
    differences: messages are a fixed binary header + raw staging-image bytes (no per-step pickling of
    Sequence objects — SURVEY.md §8f rank 1), and it is a RING with per-worker read cursors instead of one
    slot, so rank 0 can post decode step N+1 before a worker has picked up step N (decode lookahead).
    Cursors are 8-byte aligned words written by exactly one process each; readers poll them (no mp.Event,
    so workers started by torchrun rather than forked/spawned by rank 0 can attach by name)."""
    SLOTS = 4
    HDR = 128                      # head cursor + up to 8 tail cursors
    MSG_HDR = 128                  # op, payload bytes, 8 int64 arguments

    OP_EXIT, OP_PREFILL, OP_DECODE, OP_CALL = 0, 1, 2, 3

    def __init__(self, name: str, slot_payload: int, world: int, create: bool):
        from multiprocessing.shared_memory import SharedMemory
        self.world = world
        self.slot_bytes = _align(self.MSG_HDR + slot_payload, 64)
        size = self.HDR + self.SLOTS * self.slot_bytes
        if create:
            try:
                SharedMemory(name=name).unlink()           # stale segment from a crashed run
            except FileNotFoundError:
                pass
            self.shm = SharedMemory(name=name, create=True, size=size)
            self.shm.buf[:self.HDR] = bytes(self.HDR)
        else:
            deadline = time.time() + 120
            while True:
                try:
                    self.shm = SharedMemory(name=name)
                    try:                                   # attached, not owned: rank 0 unlinks it — keep Python's
                        from multiprocessing import resource_tracker      # tracker from "cleaning up" a segment that
                        resource_tracker.unregister(self.shm._name, "shared_memory")   # is not this process' to remove
                    except Exception:  # noqa: BLE001
                        pass
                    break
                except FileNotFoundError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.01)
        self.owner = create
        self.cur = np.frombuffer(self.shm.buf, dtype=np.int64, count=self.HDR // 8)   # [0] head, [1 + w] tails
        self.buf = np.frombuffer(self.shm.buf, dtype=np.uint8)
        self.sent = 0
        self.seen = 0

    def _slot(self, k: int):
        off = self.HDR + (k % self.SLOTS) * self.slot_bytes
        hdr = self.buf[off:off + self.MSG_HDR].view(np.int64)
        return hdr, self.buf[off + self.MSG_HDR:off + self.slot_bytes]

    # ---- rank 0 ----
    def send(self, op: int, args=(), payload: np.ndarray | bytes | None = None) -> None:
        tails = self.cur[1:self.world]
        spins = 0
        while self.sent - int(tails.min()) >= self.SLOTS:    # every slot still unread by some worker
            spins += 1
            if spins > 200:
                time.sleep(0.00005)
        hdr, body = self._slot(self.sent)
        n = 0
        if payload is not None:
            src = np.frombuffer(payload, dtype=np.uint8) if isinstance(payload, (bytes, bytearray)) else payload
            n = src.size
            assert n <= body.size, "step message exceeds the control channel slot"
            body[:n] = src
        hdr[0], hdr[1] = op, n
        for i, a in enumerate(args):
            hdr[2 + i] = a
        self.sent += 1
        self.cur[0] = self.sent                                # publish (x86 keeps the store order)

    # ---- workers ----
    def recv(self):
        """Block until the next message; returns (op, args[8], payload view). Call `ack()` once the payload
        has been consumed."""
        spins = 0
        while int(self.cur[0]) <= self.seen:
            spins += 1
            if spins > 20000:
                time.sleep(0.0001)
        hdr, body = self._slot(self.seen)
        return int(hdr[0]), hdr[2:10], body[:int(hdr[1])]

    def ack(self, rank: int) -> None:
        self.seen += 1
        self.cur[rank] = self.seen                             # tail of worker `rank` lives at cur[rank], rank >= 1

    def close(self) -> None:
        self.cur = self.buf = None
        try:
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except (BufferError, FileNotFoundError):
            pass


def _align(n: int, a: int = 16) -> int:
    return (n + a - 1) // a * a


MAX_PREFIX_GROUPS = 4          # shared prefixes one decode step shares at most (group slots per pack of the attention pass)


def shared_prefix_group(bt: np.ndarray, lens: np.ndarray, block_size: int, max_groups: int = MAX_PREFIX_GROUPS):
    """(k, member): the rows of a decode batch that start with the same KV blocks, and how many. member[row] = 0 (not a
    member) or a GROUP id 1, 2, ...: rows of one group have the same FIRST block id — what the prefix cache hands out when
    requests start with the same tokens (block_manager.py:58-82: a cache hit appends the block id of the request that
    registered it; requests prefilled in the step that first computed the prefix hold private copies, :110-120, and are
    not members). Group 1 is the largest group and sets k = the number of leading columns of bt on which all its rows
    agree; every further group of >= 2 rows (largest first, `max_groups` in all: two system prompts in one batch) joins
    if its rows agree on their first k columns as well — the attention pass then covers k blocks of EVERY group. Only
    blocks that lie completely before the newest token of the shortest member count (k <= (min(len) - 1) // block_size):
    the block a member is still writing is never shared. (0, None): nothing to share."""
    n = len(lens)
    if n < 2:
        return 0, None
    first = bt[:n, 0]
    vals, counts = np.unique(first, return_counts=True)
    if len(vals) == n:                                          # (the common case: all different)
        return 0, None
    order = np.argsort(-counts, kind="stable")
    j = int(order[0])
    if counts[j] < 2 or vals[j] < 0:
        return 0, None
    rows = np.nonzero(first == vals[j])[0]
    cap = min(int((int(lens[rows].min()) - 1) // block_size), bt.shape[1])
    if cap <= 0:
        return 0, None
    same = (bt[rows, :cap] == bt[rows[0], :cap]).all(axis=0)
    k = cap if same.all() else int(np.argmin(same))
    if k <= 0:
        return 0, None
    member = np.zeros(n, dtype=np.int32)
    member[rows] = 1
    gid = 1
    for j in order[1:]:
        if gid >= max_groups or counts[j] < 2:
            break
        if vals[j] < 0:
            continue
        rows = np.nonzero(first == vals[j])[0]
        if (int(lens[rows].min()) - 1) // block_size < k or not (bt[rows, :k] == bt[rows[0], :k]).all():
            continue                                            # shorter than, or not agreeing on, the k blocks the pass covers
        gid += 1
        member[rows] = gid
    return k, member


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

    def __init__(self, config: Config, rank: int, event=None):
        """`event` is accepted for signature compatibility with the reference (model_runner.py:17) and unused:
        the control channel is polled."""
        self.config = config
        hf = config.hf_config
        self.block_size = config.kvcache_block_size
        self.enforce_eager = config.enforce_eager
        self.world_size = config.tensor_parallel_size
        self.rank = rank
        assert torch.cuda.is_available(), "nano_vllm_amd needs a HIP device (no CPU fallback for the hot path)"
        ops.load_library()

        # Devices: rank r drives GPU r, as the reference does (model_runner.py:27). NVL_TP_SHARE_GPU=1 puts every
        # rank on GPU 0 — the functional test of the TP engine on a one-GPU box (with NVL_TP_BACKEND=gloo:
        # RCCL refuses two ranks on one device).
        share_gpu = os.environ.get("NVL_TP_SHARE_GPU") == "1"
        dev_index = 0 if (share_gpu or self.world_size == 1) else int(os.environ.get("LOCAL_RANK", rank))
        if self.world_size == 1:
            dev_index = torch.cuda.current_device()
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        self._own_pg = False
        if self.world_size > 1:
            if dist.is_initialized():
                # launched externally (torchrun: bench.py --tp N): the default group IS the TP group
                assert dist.get_world_size() == self.world_size and dist.get_rank() == rank, \
                    "an existing torch.distributed group must be the tensor-parallel group"
            else:
                backend = os.environ.get("NVL_TP_BACKEND", "nccl")
                kw = dict(device_id=self.device) if backend == "nccl" else {}
                dist.init_process_group(backend, f"tcp://127.0.0.1:{_tp_port()}", world_size=self.world_size,
                                        rank=rank, **kw)
                self._own_pg = True
        try:
            self._init_rest(config, hf, rank)
        except BaseException:
            # do not leave a half-built tensor-parallel group behind (the next engine of this process would adopt it)
            if self._own_pg and dist.is_initialized():
                dist.destroy_process_group()
            raise

    def _init_rest(self, config: Config, hf, rank: int):
        from .. import tp
        tp.init(rank if self.world_size > 1 else 0, self.world_size)
        self.geo = model_geometry(hf, self.world_size)
        dtype = self.geo["dtype"] or torch.bfloat16
        if isinstance(dtype, str):
            dtype = getattr(torch, dtype)
        assert dtype == torch.bfloat16, f"libnvl kernels are bf16 (config dtype {dtype})"
        self.p2p = False
        if self.world_size > 1:
            # before the memory measurements below, so the comm buffers are accounted for
            self.p2p = tp.init_p2p(min(config.max_num_seqs, 512), self.geo["hidden"], self.device)
            if not tp.capturable():
                self.enforce_eager = True          # gloo: collectives go through the host, nothing to capture
        prev_dtype = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        torch.set_default_device(self.device)
        try:
            self.model = Qwen3ForCausalLM(hf)
            if config.dummy_weights:
                init_dummy_weights(self.model, hf, config.seed)
            else:
                load_model(self.model, config.model)
            # tile-packed copies of the deep-K projections for the decode GEMM (before the KV pool is sized from what
            # is left: a second copy of those weights is the price of a 4x cheaper weight stream per CU)
            self.packed_weight_bytes, self.packed_weight_skipped = self._pack_weights()
            self.sampler = Sampler(seed=config.seed, max_rows=config.max_num_seqs)
            self._alloc_stages()
            self.warmup_model()
            self.allocate_kv_cache()
            self.graphs: dict[int, torch.cuda.CUDAGraph] = {}
            if not self.enforce_eager:
                self.capture_graphs()
        finally:
            torch.set_default_device("cpu")
            torch.set_default_dtype(prev_dtype)

        self.chan = None
        if self.world_size > 1:
            payload = max(self.dstage.nbytes, self.pstage.nbytes)
            name = f"{_SHM_NAME}_{_tp_port()}"
            if rank == 0:
                self.chan = _Channel(name, payload, self.world_size, create=True)
                dist.barrier()
            else:
                dist.barrier()
                self.chan = _Channel(name, payload, self.world_size, create=False)
                self.loop()

    def _pack_weights(self) -> tuple[int, int]:
        """Tile-packed second copies of the decode GEMMs' weights, under a BUDGET: they are allocated before the KV
        pool is sized, so every packed byte is a KV-cache byte less (Qwen3-32B at TP = 1: 64 GB of copies). Budget =
        min(NVL_PACKED_BUDGET_FRAC (default 0.25) x device memory, half of what is free once the weights are loaded);
        projections are packed in model order until the next one no longer fits, the rest keep the row-major weight
        stream (slower decode GEMM, same results bit for bit) — on a 192 GB part Qwen3-32B packs its first ~48 GB and
        keeps its KV pool. Returns (bytes packed, projections left unpacked for lack of budget)."""
        from ..layers import LinearBase, ParallelLMHead
        free, total = torch.cuda.mem_get_info(self.device)
        budget = min(float(os.environ.get("NVL_PACKED_BUDGET_FRAC", "0.25")) * total, 0.5 * free)
        used = skipped = 0
        for m in self.model.modules():
            if isinstance(m, (LinearBase, ParallelLMHead)):
                got = m.pack_for_decode(budget - used)
                if got < 0:
                    skipped += 1
                else:
                    used += got
        return used, skipped

    # ------------------------------------------------------------------ lifecycle / TP control channel
    def exit(self):
        if self.world_size > 1 and self.rank == 0 and self.chan is not None:
            self.chan.send(_Channel.OP_EXIT)
        self.graphs = {}
        self.graphs_px = {}
        torch.cuda.synchronize()
        if self.world_size > 1:
            from .. import tp
            problem = None
            if self.p2p:
                try:
                    tp.comm().status()        # did any P2P collective ever give up waiting for a peer?
                except ops.NvlError as ex:
                    problem = ex
            dist.barrier()
            if self.chan is not None:
                self.chan.close()
                self.chan = None
            tp.shutdown()
            if self._own_pg:
                dist.destroy_process_group()
            if problem is not None:           # after the clean-up, so the other ranks are not left in a barrier
                raise problem

    def loop(self):
        """Worker ranks: execute the steps rank 0 posts (model_runner.py:61-74)."""
        ch = self.chan
        while True:
            op, args, body = ch.recv()
            if op == _Channel.OP_EXIT:
                ch.ack(self.rank)
                del args, body                 # views into the shared segment: it cannot be unmapped while they live
                self.exit()
                break
            if op == _Channel.OP_DECODE:
                n = int(args[0])
                self.dstage.flip()
                self.dstage.host.numpy()[:body.size] = body
                ch.ack(self.rank)
                self._launch_decode(n)
                self._inflight.clear()             # workers never collect: their flight list is bookkeeping only
            elif op == _Channel.OP_PREFILL:
                info = dict(n=int(args[0]), ns=int(args[1]), max_q=int(args[2]), max_k=int(args[3]),
                            paged=bool(args[4]), have_slots=bool(args[5]))
                self.pstage.uploaded[0].synchronize()          # the previous prefill's upload has been consumed
                self.pstage.host.numpy()[:body.size] = body
                ch.ack(self.rank)
                self._launch_prefill(info)
            else:
                method, *margs = pickle.loads(body.tobytes())
                ch.ack(self.rank)
                getattr(self, method)(*margs)

    def call(self, method, *args):
        """Rank 0's entry point (model_runner.py:84-89). Steps are broadcast to the workers by the methods
        themselves (as staging images); anything else goes through `call_all`."""
        return getattr(self, method)(*args)

    def call_all(self, method, *args):
        """Run `method(*args)` on every rank (rare control calls; pickled)."""
        if self.world_size > 1 and self.rank == 0:
            self.chan.send(_Channel.OP_CALL, payload=pickle.dumps([method, *args]))
        return getattr(self, method)(*args)

    # ------------------------------------------------------------------ buffers
    def _alloc_stages(self):
        cfg = self.config
        self.max_bs = cfg.max_num_seqs
        self.max_blocks = -(-cfg.max_model_len // self.block_size)
        mb, w = self.max_bs, self.max_blocks
        self.dstage = _Stage([
            ("ids", np.int64, (mb,)), ("pos", np.int64, (mb,)), ("rng", np.uint64, (2,)), ("rkey", np.int64, (mb,)),
            ("slots", np.int32, (mb,)), ("ctx", np.int32, (mb,)), ("temps", np.float32, (mb,)),
            ("src", np.int32, (mb,)), ("shp", np.int32, (1 + mb,)), ("bt", np.int32, (mb, w)),
        ], self.device, host_copies=2)
        for image in self.dstage.nps:
            image["slots"][:] = -1
            image["bt"][:] = -1
            image["src"][:] = -1
        self.dstage.upload()
        # per pinned image: (seq id, #blocks, allocation stamp) per decode row whose block-table row is current
        self._row_keys = [np.full((mb, 3), -1, dtype=np.int64) for _ in self.dstage.nps]
        self._dirty = [0 for _ in self.dstage.nps]
        nt = cfg.max_num_batched_tokens
        ns = min(cfg.max_num_seqs, nt)
        self.pstage = _Stage([
            ("ids", np.int64, (nt,)), ("pos", np.int64, (nt,)), ("rng", np.uint64, (2,)), ("rkey", np.int64, (ns,)),
            ("slots", np.int32, (nt,)), ("cu_q", np.int32, (ns + 1,)), ("cu_k", np.int32, (ns + 1,)),
            ("temps", np.float32, (ns,)), ("bt", np.int32, (ns, w)),
        ], self.device)
        self.step_count = 0
        self.share_prefix = False            # (decided below, once the device side exists)
        self._seen_cached_kv = False         # has any prefill step attended to K/V it did not compute itself?
        self._inflight: list = []            # decode steps enqueued and not yet collected (at most two)
        self._flight_parity = 0
        self._last_rows: dict = {}
        if self.device.type != "cuda":       # host-only instance (tests of the staging logic): no device buffers
            return
        self.tokens_dev = torch.zeros(max(mb, ns), dtype=torch.int64, device=self.device)
        self.tokens_host = torch.zeros(max(mb, ns), dtype=torch.int64, device="cpu", pin_memory=True)
        self.tokens_host_b = torch.zeros(max(mb, ns), dtype=torch.int64, device="cpu", pin_memory=True)
        ws_bytes = ops.paged_attn_decode_workspace_bytes(mb, self.geo["heads"], cfg.max_model_len)
        # zeroed once: the kernel's arrival counters live in it and are left at zero by every launch
        self.decode_ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=self.device)
        # per-step work plan of the decode attention launches (ops.decode_plan): made once per step by the first
        # node of the decode graph, read by every layer's attention launch. NVL_DECODE_PLAN=0: every launch derives
        # its own schedule again (A/B measurements)
        self.use_plan = os.environ.get("NVL_DECODE_PLAN", "1") != "0"
        self.decode_plan = torch.zeros(ops.decode_plan_bytes(), dtype=torch.uint8, device=self.device)
        # shared-prefix attention pass (include/nvl.h, nvl_decode_plan): a decode step in which a group of sequences starts
        # with the same KV blocks (prefix-cache hits on one system prompt) reads them once per pack of 16 / G rows. The packs
        # take workgroups away from the stream-K grid of the same launch (round 5: a launch of their own per layer), so a step
        # takes the pass only when the K/V bytes it saves are worth that (prepare_decode);
        # the graph of a bucket WITH the pass is captured the first time a step of that bucket wants it.
        # NVL_SHARED_PREFIX=0 switches it off; NVL_SHARED_PREFIX_MIN_MB sets the threshold (saved MB per layer). The
        # default sits under the smallest saving measured to pay in the final form (two items per pack workgroup, bf16 qkv
        # on shared-prefix steps — profiles/r06_shared_prefix_engine_check.json: config 3's workload on the 8B shapes at
        # 96 / 128 / 160 / 256 sequences +3.5 / +5.3 / +5.6 / +6.8 % tok/s with the pass forced on, 111 MB saved at 96;
        # r06_shared_prefix_items_per_wg.json: 0.6B shapes +9 % at 96 sequences = 130 MB) and above the ones measured to lose
        # (17-45 MB: one-kv-head rank shapes, 48 sequences): the packs take workgroups from the stream-K grid and their
        # walk is a latency chain, and at small batches L2 / Infinity Cache already absorb most of the repeated reads.
        # The opt-in fp8 KV cache takes the pass under the same rule since the packs moved into the stream-K launch
        # (256 sequences: 0.6B shapes 69.4 -> 75.7 k tok/s, 8B 19.97 -> 20.55 k; with round 5's separate launch it lost).
        self.share_prefix = (self.use_plan and os.environ.get("NVL_SHARED_PREFIX", "1") != "0"
                             and ops.decode_attention_shares_prefixes(self.geo["heads"], self.geo["kv_heads"],
                                                                      self.block_size))
        self.share_prefix_min_bytes = float(os.environ.get("NVL_SHARED_PREFIX_MIN_MB", "100")) * 1e6
        self.decode_plan_px = torch.zeros(ops.decode_plan_bytes(), dtype=torch.uint8, device=self.device)
        self.graphs_px: dict[tuple, torch.cuda.CUDAGraph] = {}      # (bucket, group slots) -> graph with the pass
        self.prefix_steps = 0                # decode steps that ran the shared-prefix pass (reporting)
        self.prefix_multi_steps = 0          # ... with more than one shared prefix in the batch
        self._step_done = [torch.cuda.Event(), torch.cuda.Event()]
        # TP with the xGMI P2P collectives: every spin of those kernels is bounded, and a timeout is LATCHED in the shared
        # flag region while the step carries on with an invalid sum. The last node of every step (prefill and decode,
        # eager and captured) ORs the ranks' latches into `comm_status_dev`; rank 0 takes the word to the host with the
        # step's ids and raises from step() / generate() (exit() keeps its own check as the backstop).
        self.comm_status_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.comm_status_host = torch.zeros(2, dtype=torch.int32, device="cpu", pin_memory=True)

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
        self._launch_prefill(self.prepare_prefill(seqs))       # every rank warms up on its own (same shapes)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def allocate_kv_cache(self):
        cfg, geo = self.config, self.geo
        free, total = torch.cuda.mem_get_info()
        used = total - free
        stats = torch.cuda.memory_stats()
        peak, current = stats["allocated_bytes.all.peak"], stats["allocated_bytes.all.current"]
        fp8 = cfg.kv_cache_dtype == "fp8"
        block_bytes = 2 * geo["layers"] * self.block_size * geo["kv_heads"] * geo["head_dim"] * (1 if fp8 else 2)
        if cfg.num_kvcache_blocks <= 0:
            assert not (self.world_size > 1 and os.environ.get("NVL_TP_SHARE_GPU") == "1"), \
                "ranks sharing one GPU cannot size the KV cache from free memory: pass num_kvcache_blocks"
            cfg.num_kvcache_blocks = int(total * cfg.gpu_memory_utilization - used - peak + current) // block_bytes
        if self.world_size > 1:
            # the scheduler (rank 0) must not hand out a block some rank does not have
            from .. import tp
            n = torch.tensor([cfg.num_kvcache_blocks], dtype=torch.int64, device="cpu")
            tp.group_all_reduce(n, op=dist.ReduceOp.MIN)
            cfg.num_kvcache_blocks = int(n.item())
        assert cfg.num_kvcache_blocks > 0, "no memory left for the KV cache"
        # zero-filled: masked tail rows of a block are multiplied by P == 0 in the decode kernel
        # Layer-major: a layer's K and V pools sit next to each other (the reference's [2, L, ...] puts them
        # 137 GB apart at this pool size, which costs the decode kernel 2.7 % of its bandwidth on MI355X —
        # tools/attn_replay.py --cache-blocks 9377 [--layer-major]). `kv_cache` keeps the [2, L, ...] indexing
        # as a transposed view.
        self._kv_storage = torch.zeros(geo["layers"], 2, cfg.num_kvcache_blocks, geo["kv_heads"], self.block_size,
                                       geo["head_dim"], dtype=torch.uint8 if fp8 else torch.bfloat16,
                                       device=self.device)
        if fp8:
            self._kv_storage = self._kv_storage.view(torch.float8_e4m3fn)      # zero bytes are +0.0 in e4m3 too
        self.kv_cache = self._kv_storage.transpose(0, 1)
        layer = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = self.kv_cache[0, layer]
                module.v_cache = self.kv_cache[1, layer]
                layer += 1

    # ------------------------------------------------------------------ batch preparation (host, rank 0)
    def prepare_prefill(self, seqs: list[Sequence]) -> dict:
        """Fill the prefill staging block (semantics of model_runner.py:129-170)."""
        ev = self.pstage.uploaded[0]
        if ev is not None:
            ev.synchronize()                                   # the previous prefill's upload has been consumed
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
            st["rkey"][i] = seq.rng_key | (end << 32)          # the draw of the token that will sit at position `end`
            n += lq
        ns = len(seqs)
        paged = int(cu_k[ns]) > int(cu_q[ns])                  # some K/V must come from the cache
        if paged:
            self._seen_cached_kv = True                        # (prefix-cache hits or chunked prefill: blocks MAY be shared)
            bt = st["bt"]
            for i, seq in enumerate(seqs):
                t = seq.block_table
                bt[i, :len(t)] = t
                bt[i, len(t):] = -1
        return dict(n=n, ns=ns, max_q=max_q, max_k=max_k, paged=paged, have_slots=have_slots)

    def prepare_decode(self, seqs: list[Sequence], prev_rows: dict | None = None) -> int:
        """Fill the current pinned decode image (semantics of model_runner.py:172-188); returns bs.
        `prev_rows` (lookahead): seq_id -> row of the sequence in the decode step that is still in flight; the
        input ids of those sequences are NOT staged — the graph's first node copies them from that step's
        sampled ids on the device (nvl_feed_tokens) — and `src` carries the rows. A sequence that was not part
        of the in-flight step (more running sequences than max_num_seqs: one finishes, the next one moves
        up) gets src = -1 and its id staged: its last token came from an earlier, already collected step."""
        st, bs = self.dstage.np, self.block_size
        n = len(seqs)
        lens = np.fromiter((s.num_tokens for s in seqs), dtype=np.int64, count=n)
        if prev_rows is None:
            st["ids"][:n] = [s.last_token for s in seqs]
            st["src"][:n] = -1
        else:
            src = np.fromiter((prev_rows.get(s.seq_id, -1) for s in seqs), dtype=np.int32, count=n)
            st["src"][:n] = src
            for i in np.nonzero(src < 0)[0]:
                st["ids"][i] = seqs[i].last_token
        st["pos"][:n] = lens - 1
        st["ctx"][:n] = lens
        # sampler key per row: (request ordinal, position of the token being drawn) — independent of the batch row
        st["rkey"][:n] = np.fromiter((s.rng_key for s in seqs), dtype=np.int64, count=n) | (lens << 32)
        last_blk = np.fromiter((s.block_table[-1] for s in seqs), dtype=np.int64, count=n)
        st["slots"][:n] = last_blk * bs + (lens - 1) % bs
        st["temps"][:n] = [s.temperature for s in seqs]
        # block tables: rewrite a row only when its (sequence, #blocks, allocation stamp) changed in THIS image.
        # The stamp matters: a preempted sequence that is prefilled again may come back to the same row with the
        # same number of blocks but other block ids (kv_blocks.BlockManager.allocate).
        key = self._row_keys[self.dstage.cur]
        ids = np.fromiter((s.seq_id for s in seqs), dtype=np.int64, count=n)
        nblk = np.fromiter((len(s.block_table) for s in seqs), dtype=np.int64, count=n)
        gen = np.fromiter((s.table_gen for s in seqs), dtype=np.int64, count=n)
        stale = np.nonzero((key[:n, 0] != ids) | (key[:n, 1] != nblk) | (key[:n, 2] != gen) | (ids < 0))[0]
        bt = st["bt"]
        for i in stale:
            t = seqs[i].block_table
            bt[i, :len(t)] = t
            bt[i, len(t):] = -1
        key[:n, 0], key[:n, 1], key[:n, 2] = ids, nblk, gen
        # shared-prefix pass: [0] = leading blocks the member rows have in common (0: plain step), [1 + row] = member
        # (two rows can only hold the same block when a prefill step took K/V from the cache: a run that never did —
        #  the headline workload — skips the search)
        k, member = self._prefix_group_worth_a_pass(bt, lens, n) if (self.share_prefix and self._seen_cached_kv) else (0, None)
        st["shp"][0] = k
        if k > 0:
            st["shp"][1:1 + n] = member                     # group ids (0: not a member)
        # neutralise rows used by a previous, larger batch (graph padding: slot -1, context 0)
        dirty = self._dirty[self.dstage.cur]
        if dirty > n:
            st["slots"][n:dirty] = -1
            st["ctx"][n:dirty] = 0
            key[n:dirty] = -1
        self._dirty[self.dstage.cur] = n
        return n

    def _prefix_group_worth_a_pass(self, bt: np.ndarray, lens: np.ndarray, n: int):
        """shared_prefix_group of the batch, or (0, None) when the K/V bytes the shared-prefix pass would save per layer
        (the blocks are read once per pack of 16 / G sequences instead of once per member) do not pay for its launch."""
        k, member = shared_prefix_group(bt[:n], lens, self.block_size)
        if k == 0:
            return 0, None
        pack = 16 // (self.geo["heads"] // self.geo["kv_heads"])
        esize = 1 if self.config.kv_cache_dtype == "fp8" else 2
        m = int((member > 0).sum())
        # the pass reads a group's prefix once per pack of CONSECUTIVE rows that holds a member of it (the other rows ride
        # along as zero columns): members scattered among other rows share less than m / pack packs
        idx = np.nonzero(member)[0]
        npacks = len(np.unique((idx // pack) * (MAX_PREFIX_GROUPS + 1) + member[idx]))
        saved = k * self.block_size * (m - npacks) * self.geo["kv_heads"] * 2 * 128 * esize
        return (k, member) if saved >= self.share_prefix_min_bytes else (0, None)

    # ------------------------------------------------------------------ forward
    def _next_rng(self, st: _Stage) -> None:
        """`rng[0]` is the stream offset added to every row's draw: constant 0 — the per-row keys (`rkey`: request
        ordinal | position << 32) already make every (sequence, position) draw unique, and a step counter here would
        tie a request's tokens to how many steps other requests took before it."""
        self.step_count += 1
        st.np["rng"][0] = 0

    def _sample(self, hidden, temps, out, rng, sampler, rkey=None):
        """lm_head + sampler on the current stream; TP > 1: every rank samples its vocabulary shard and the
        per-row winners are merged on every rank (no [B, V] gather, embed_head.py:62-65)."""
        col0 = self.rank * self.geo["vocab_per_rank"]
        if self.world_size == 1:
            sampler(self.model.compute_logits(hidden), temps, out=out, offset_dev=rng, row_keys=rkey)
        else:
            logits = self.model.compute_logits_shard(hidden)
            sampler.forward_shard(logits, temps, col0, out, offset_dev=rng, row_keys=rkey)

    def _decode_rows(self, r0: int, r1: int, ws, sampler, plan=None, prefix: int = 0):
        """Decode forward for rows [r0, r1) of the static device buffers, on the current stream. `prefix` > 0: the plan
        carries the step's shared-prefix groups (staged as `shp`), the attention launches run the shared pass with that
        many group slots per pack (1: one shared prefix in the step; MAX_PREFIX_GROUPS: several)."""
        t = self.dstage.t
        assert not prefix or r0 == 0           # (the member flags are staged per row of the whole batch)
        if plan is not None:
            ops.decode_plan(t["ctx"][r0:r1], self.geo["heads"], self.geo["kv_heads"], self.config.max_model_len, plan,
                            shared_prefix=t["shp"] if prefix else None, block_size=self.block_size,
                            prefix_groups=max(prefix, 1))
        set_context(False, slot_mapping=t["slots"][r0:r1], context_lens=t["ctx"][r0:r1],
                    block_tables=t["bt"][r0:r1], decode_workspace=ws, max_context=self.config.max_model_len,
                    decode_plan=plan, shared_prefix=bool(prefix) and plan is not None)
        hidden = self.model(t["ids"][r0:r1], t["pos"][r0:r1])
        self._sample(hidden, t["temps"][r0:r1], self.tokens_dev[r0:r1], t["rng"][:1], sampler, t["rkey"][r0:r1])
        reset_context()
        self._collect_comm_status()

    def _collect_comm_status(self) -> None:
        """Last node of a step (every rank; captured with the decode graphs): did any rank's P2P collective time out?"""
        if self.p2p:
            from .. import tp
            tp.comm().status_async(self.comm_status_dev)

    def _raise_if_collective_timed_out(self, word: int, what: str) -> None:
        if word:
            raise ops.NvlError(f"a tensor-parallel P2P collective gave up waiting for a peer during {what}: the step's sums, and "
                               "every token sampled from them, are invalid (nvl_allreduce_status). NVL_TP_P2P=0 runs the "
                               "collectives over the process group instead")

    @torch.inference_mode()
    def _forward_decode(self, bs: int, prefix: int = 0):
        """Decode forward on the static device buffers (captured per bucket, or run eagerly): layers + lm_head
        + sampler, at any TP degree (the reference captures the layers only and runs lm_head, the logits
        gather and the sampler eagerly, model_runner.py:212,218). (A micro-batched form — two half-batch chains on
        two branches of the captured graph — was measured slower in rounds 1-3 and is gone: DESIGN.md, measured
        negatives.)"""
        # input ids of sequences that were in the previous decode step come straight from its sampled ids
        t = self.dstage.t
        ops.feed_tokens(t["ids"][:bs], t["src"][:bs], self.tokens_dev)
        plan = (self.decode_plan_px if prefix else self.decode_plan) if self.use_plan else None
        self._decode_rows(0, bs, self.decode_ws, self.sampler, plan, prefix)

    @torch.inference_mode()
    def _launch_decode(self, n: int) -> None:
        """Every rank: upload the current image, run the step, start the D2H of the ids (rank 0)."""
        self.dstage.upload()
        bucket = next((b for b in self.graph_bs if b >= n), None) if self.graphs else None
        # (every rank reads the same image: the group count is the largest group id among the staged flags)
        prefix = 0
        if self.share_prefix and int(self.dstage.np["shp"][0]) > 0:
            prefix = 1 if int(self.dstage.np["shp"][1:1 + n].max()) <= 1 else MAX_PREFIX_GROUPS
        self.prefix_steps += int(prefix > 0)
        self.prefix_multi_steps += int(prefix > 1)
        key = (bucket, prefix)
        if bucket is not None and prefix and key not in self.graphs_px and not self._capture_prefix_graph(bucket, prefix):
            # the capture failed (single rank only; _capture_prefix_graph raises under TP): this step replays the bucket's
            # plain graph, whose plan is built without the staged groups
            self.prefix_steps -= 1
            self.prefix_multi_steps -= int(prefix > 1)
            prefix = 0
        if bucket is not None and prefix:
            self.graphs_px[key].replay()
        elif bucket is not None:
            self.graphs[bucket].replay()
        else:
            self._forward_decode(n, prefix)
        host = self.tokens_host if self._flight_parity == 0 else self.tokens_host_b
        done = self._step_done[self._flight_parity]
        if self.rank == 0:
            host[:n].copy_(self.tokens_dev[:n], non_blocking=True)
            if self.p2p:
                self.comm_status_host[self._flight_parity:self._flight_parity + 1].copy_(self.comm_status_dev, non_blocking=True)
        done.record()
        self._inflight.append((n, host, done, self._flight_parity))
        self._flight_parity ^= 1

    def decode_begin(self, seqs: list[Sequence], staged: bool = False) -> int:
        """Enqueue one decode step on every rank (H2D of the staging image, graph replay or eager forward,
        D2H of the sampled ids) and return without waiting; up to two steps may be in flight. `staged`: the
        image was already filled by `stage_next_decode` (lookahead)."""
        if not staged:
            self.dstage.flip()
            self.prepare_decode(seqs)
        n = len(seqs)
        self._next_rng(self.dstage)
        if self.chan is not None:
            self.chan.send(_Channel.OP_DECODE, (n,), self.dstage.host.numpy())
        self._launch_decode(n)
        self._last_rows = {s.seq_id: i for i, s in enumerate(seqs)}
        return n

    def decode_end(self) -> list[int] | None:
        """Wait for the OLDEST step enqueued by `decode_begin` and return its sampled ids."""
        n, host, done, parity = self._inflight.pop(0)
        done.synchronize()
        if self.p2p and self.rank == 0:
            self._raise_if_collective_timed_out(int(self.comm_status_host[parity]), "a decode step")
        return host[:n].tolist()

    def stage_next_decode(self, seqs: list[Sequence]) -> None:
        """Lookahead: fill the other pinned image for the NEXT decode step while the step just enqueued runs
        (everything but the input ids, which the graph takes from that step's output on the device)."""
        self.dstage.flip()
        self.prepare_decode(seqs, self._last_rows)

    def _run_decode(self, seqs: list[Sequence]) -> list[int] | None:
        self.decode_begin(seqs)
        return self.decode_end()

    @torch.inference_mode()
    def _launch_prefill(self, info: dict) -> None:
        """Every rank: upload the prefill image and run the step (ids of each sequence's last token land in
        tokens_dev[:ns] on every rank)."""
        self.pstage.upload()
        t = self.pstage.t
        n, ns = info["n"], info["ns"]
        set_context(True, t["cu_q"][:ns + 1], t["cu_k"][:ns + 1], info["max_q"], info["max_k"],
                    t["slots"][:n] if info["have_slots"] else None, None, t["bt"][:ns] if info["paged"] else None)
        hidden = self.model(t["ids"][:n], t["pos"][:n])
        self._sample(hidden, t["temps"][:ns], self.tokens_dev[:ns], t["rng"][:1], self.sampler, t["rkey"][:ns])
        reset_context()
        self._collect_comm_status()

    def _run_prefill(self, seqs: list[Sequence]) -> list[int] | None:
        info = self.prepare_prefill(seqs)
        self._next_rng(self.pstage)
        if self.chan is not None:
            self.chan.send(_Channel.OP_PREFILL, (info["n"], info["ns"], info["max_q"], info["max_k"],
                                                 int(info["paged"]), int(info["have_slots"])),
                           self.pstage.host.numpy())
        self._launch_prefill(info)
        ns = info["ns"]
        self.tokens_host[:ns].copy_(self.tokens_dev[:ns], non_blocking=True)
        if self.p2p:
            self.comm_status_host[:1].copy_(self.comm_status_dev, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        if self.p2p:
            self._raise_if_collective_timed_out(int(self.comm_status_host[0]), "a prefill step")
        return self.tokens_host[:ns].tolist()

    def run(self, seqs: list[Sequence], is_prefill: bool) -> list[int] | None:
        return self._run_prefill(seqs) if is_prefill else self._run_decode(seqs)

    # ------------------------------------------------------------------ hipGraph capture
    @torch.inference_mode()
    def capture_graphs(self):
        """One graph per batch bucket, largest first so the pool is sized once
        (buckets as model_runner.py:234). TP > 1: every rank captures the same sequence of launches, including
        the xGMI collectives (enqueue-only kernels whose epochs live in device memory)."""
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
        self.graph_pool = pool
        if self.share_prefix:
            # The graph of the LARGEST bucket with the shared-prefix pass is captured here, warm, on the neutral inputs (every
            # row is padding: the pass finds nothing to do) — a failure of that path surfaces at start-up, and the pool holds
            # what a prefix graph needs before the serving path asks for one; the other buckets' prefix graphs are captured
            # when a step first wants them (most workloads never do). The kernels' LDS reservations are made by the launcher
            # of the plain kernel (attn_decode.hip), i.e. during the eager warm-ups above, never inside a capture.
            self._forward_decode(self.graph_bs[-1], prefix=1)
            torch.cuda.synchronize()
            self._capture_prefix_graph(self.graph_bs[-1], 1)
        torch.cuda.synchronize()
        from .. import layers
        layers.release_tuning_scratch()                 # the decode-GEMM choices of every bucket are made by now

    @torch.inference_mode()
    def _capture_prefix_graph(self, bs: int, groups: int = 1) -> bool:
        """The decode graph of bucket `bs` WITH the shared-prefix attention pass, captured the first time a step wants
        it (most workloads never do; the largest bucket's is captured at start-up). No warm-up run on the serving path:
        the static buffers hold a real step's inputs by now, every choice the launches make was made when the bucket's
        plain graph was captured, and the allocations come out of the plain graphs' pool. Every rank captures at the same
        step (the decision is read from the staged image). Returns False when the capture failed on a single-rank
        engine (the caller runs the step plain, and the pass stays off for this bucket); under TP a failure is raised —
        the ranks could not agree on a fallback without another exchange."""
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, self.graph_pool):
                self._forward_decode(bs, prefix=groups)
        except Exception as e:      # OOM of the pool's headroom, a launch refused inside the capture
            if self.world_size > 1:
                raise
            import warnings
            warnings.warn(f"shared-prefix graph of bucket {bs} could not be captured ({e!r}); the pass is off")
            self.share_prefix = False
            return False
        self.graphs_px[(bs, groups)] = graph
        return True
