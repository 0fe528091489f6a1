"""Tensor-parallel plumbing: one process per GPU.

The reference all-reduces inline on the compute stream after every row-parallel GEMM and gathers the
full logits on rank 0 (nano-vllm layers/linear.py:153-156, embed_head.py:41,62-65), all through NCCL.
On MI355X the 8 GPUs are fully connected by point-to-point xGMI links, so the collectives are split by
size:

  * decode-sized messages (<= a few MB: [batch, hidden] bf16, 129 of them per Qwen3-32B step) go through
    the hand-written P2P kernels of csrc/comm.hip (`ops.P2PComm`): every rank reads its peers' buffers
    directly, the all-reduce is fused with the residual-add + RMSNorm that always follows it, and the
    logits gather is replaced by an 8-bytes-per-row exchange of each vocabulary shard's sampling winner.
    Enqueue-only => captured inside the decode hipGraph.
  * prefill-sized all-reduces (tens of MB, bandwidth-bound) go through RCCL (`torch.distributed` backend
    "nccl"), chunk-pipelined against the next GEMM on a side HIP stream (`linear_allreduce`).

The P2P path is verified when the group is set up (`init_p2p`): a comparison with the process group's own
all-reduce, then a randomised stress run (>= 1000 back-to-back collectives of changing shapes with exactly
predictable results, eager and replayed from a hipGraph, with the producer GEMM writing into the shared
buffer, under concurrent HBM traffic) of the FENCED hand-off and of the fence-free "lean" one. The lean form is
selected only when both runs are clean on the topology at hand; otherwise the engine stays fenced, and if the
fenced form fails too (or IPC mapping is unavailable) it falls back to the process group for everything and
says so. `NVL_TP_P2P=0` forces that fallback; `NVL_TP_P2P_HANDOFF=fenced` pins the fenced form (auto, the default, and lean both still require the stress run to pass before the lean form is used). With backend "gloo" (the functional test
of the TP engine on a single GPU: RCCL refuses two ranks on one device) the group collectives are staged
through host memory where gloo has no device implementation.
"""
from __future__ import annotations

import os
import warnings

import torch
import torch.distributed as dist

_OVERLAP_MIN_TOKENS = 2048     # below this the all-reduce is latency-bound: no chunking
_OVERLAP_CHUNKS = 4

_side_stream: torch.cuda.Stream | None = None

# The tensor-parallel group is EXPLICIT state set by the engine (`init`), never inferred from
# torch.distributed's default group: a caller may have its own process group for something else —
# bench.py --gpus N runs N independent TP=1 replicas under one data-parallel group, and those engines
# must not shard their weights across it.
_rank, _size = 0, 1
_backend = ""
_comm = None                   # ops.P2PComm | None
_group_of_one = False          # a 1-rank group whose collectives ARE issued (see init)
_handoff_report = ""           # what init_p2p decided and why (bench.py prints it)


def init(rank: int, size: int, issue_collectives: bool = False) -> None:
    """Declare this process' place in the engine's tensor-parallel group (size 1 = no TP). With
    size > 1 the default torch.distributed group must be that group (engine/runner.py creates it).
    `issue_collectives` with size 1: treat the (initialised) 1-rank default group as a real group — every
    collective below is issued to the backend instead of being short-cut. Mathematically the identity; it is
    how the RCCL code paths (side-stream all-reduce overlap, graph capture of a group collective) are executed
    on a machine with a single GPU (tests/test_tp_gpu.py)."""
    global _rank, _size, _backend, _comm, _group_of_one
    assert 0 <= rank < size
    assert size == 1 or (dist.is_initialized() and dist.get_world_size() == size), "TP group not initialised"
    if _comm is not None:
        _comm.close()
        _comm = None
    _rank, _size = rank, size
    _group_of_one = bool(issue_collectives and size == 1)
    if _group_of_one:
        assert dist.is_initialized() and dist.get_world_size() == 1, "issue_collectives needs a 1-rank default group"
    _backend = dist.get_backend() if (size > 1 or _group_of_one) else ""


def _active() -> bool:
    """Are collectives issued (more than one rank, or a 1-rank group in issue_collectives mode)?"""
    return _size > 1 or _group_of_one


def handoff_report() -> str:
    """Which P2P hand-off flavour this engine runs and how it was chosen ("" without P2P collectives)."""
    return _handoff_report


def world() -> tuple[int, int]:
    """(rank, world_size) of the engine's tensor-parallel group; (0, 1) without TP."""
    return _rank, _size


def comm():
    return _comm


def capturable() -> bool:
    """Can a decode step's collectives be recorded into a hipGraph?"""
    return (not _active()) or _comm is not None or _backend == "nccl"


# ------------------------------------------------------------------------------------------------
def _exchange_bytes(blob: bytes) -> list[bytes]:
    out = [None] * _size
    dist.all_gather_object(out, blob)
    return out


def init_p2p(max_rows: int, hidden: int, device: torch.device) -> bool:
    """Collective. Set up the xGMI P2P collectives for messages up to [max_rows, hidden] bf16 and verify
    them against the process group. Returns whether they are in use."""
    global _comm, _handoff_report
    if _size == 1 or os.environ.get("NVL_TP_P2P", "1") == "0":
        return False
    from . import ops
    ok, why = True, ""
    try:
        c = ops.P2PComm(_rank, _size, max(max_rows * hidden * 2, 1 << 16), _exchange_bytes, dist.barrier)
    except Exception as ex:  # noqa: BLE001 - any failure here means "no IPC on this system": fall back loudly
        c, ok, why = None, False, repr(ex)
    # every rank must take the same decision
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu")
    group_all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if c is not None:
            c.close()
        warnings.warn(f"nano_vllm_amd: xGMI P2P collectives unavailable ({why or 'a peer failed'}); using {_backend}")
        return False
    # self-check on random data against the process group's all-reduce (rows chosen to cover the one-shot and
    # the two-shot kernels, with and without the fused norm)
    gen = torch.Generator(device="cpu").manual_seed(1234 + _rank)
    good = True
    for rows in (1, min(max_rows, 37)):
        if not c.fits(rows, hidden):
            continue
        x = torch.randn(rows, hidden, generator=gen).to(torch.bfloat16).to(device)
        ref = x.float()
        group_all_reduce(ref)
        got = c.all_reduce(x.clone())
        res = torch.randn(rows, hidden, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16).to(device)
        w = torch.ones(hidden, dtype=torch.bfloat16, device=device)
        res2 = res.clone()
        y = c.all_reduce_add_rmsnorm(x.clone(), res2, w, 1e-6)
        torch.cuda.synchronize()
        try:
            c.status()
        except Exception:  # noqa: BLE001
            good = False
        s = ref.to(torch.bfloat16).float() + res.float()
        yref = s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6)
        tol = 2e-2 * float(ref.abs().max()) + 1e-3
        good = good and bool((got.float() - ref).abs().max() <= tol) and bool((y.float() - yref).abs().max() <= 0.05)
        good = good and bool((res2.float() - s).abs().max() <= tol)
    want = os.environ.get("NVL_TP_P2P_HANDOFF", "auto")
    assert want in ("auto", "fenced", "lean"), "NVL_TP_P2P_HANDOFF must be auto, fenced or lean"
    epochs = int(os.environ.get("NVL_TP_P2P_STRESS_EPOCHS", "1000"))
    good = _agree(good)                                   # (every rank takes the same branch from here on)
    if good:
        good = _agree(_stress(c, "fenced", max_rows, hidden, device, epochs))
    if not good:
        c.close()
        _handoff_report = f"P2P collectives failed their self-check: {_backend} process group"
        warnings.warn(f"nano_vllm_amd: xGMI P2P collectives FAILED their self-check; using {_backend}")
        return False
    if want == "fenced":
        _handoff_report = "fenced (NVL_TP_P2P_HANDOFF=fenced)"
    elif _agree(_stress(c, "lean", max_rows, hidden, device, epochs)):
        c.set_handoff("lean")
        _handoff_report = (f"lean ({'NVL_TP_P2P_HANDOFF=lean, ' if want == 'lean' else ''}"
                           f"stress self-check of both flavours passed: {epochs} randomised epochs each)")
    else:
        c.set_handoff("fenced")
        _handoff_report = "fenced (the lean hand-off FAILED its stress self-check on this topology)"
        warnings.warn("nano_vllm_amd: the fence-free P2P hand-off failed its stress self-check here; staying fenced")
    dist.barrier()
    _comm = c
    return True


def _agree(ok: bool) -> bool:
    """Collective AND over the ranks (every rank must take the same decision)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu")
    group_all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def _stress(c, flavour: str, max_rows: int, hidden: int, device: torch.device, epochs: int) -> bool:
    """Collective. `epochs` back-to-back P2P collectives under `flavour` with exactly predictable results:
    rank r contributes small integers f(r, epoch, element) (exact in bf16, sums exact in fp32), so every rank knows
    the expected sum without asking the process group, and a single wrong element in any epoch is caught on the
    device (mismatch counters, read once at the end). Shapes change every epoch (same seeded sequence on all
    ranks: one-shot and two-shot kernels, different grids), a third of the epochs let a GEMM produce the input
    straight into the shared buffer, a third run the fused add+RMSNorm epilogue, the last fifth is ONE captured
    hipGraph replayed with changing inputs — while a side stream streams HBM the whole time."""
    import random
    c.set_handoff(flavour)
    dist.barrier()
    world, rank = c.world, c.rank
    rnd = random.Random(20240923)                               # identical on every rank
    cap = max(1, min(max_rows, c.max_bytes // (hidden * 2)))
    bad = torch.zeros(1, dtype=torch.int64, device=device)
    col = torch.arange(hidden, device=device, dtype=torch.int32)
    eye = torch.eye(hidden, dtype=torch.bfloat16, device=device)
    ones = torch.ones(hidden, dtype=torch.bfloat16, device=device)

    def pattern(r: int, e: int, rows: int) -> torch.Tensor:     # integers in [-8, 8]
        row = torch.arange(rows, device=device, dtype=torch.int32).unsqueeze(1)
        return (((row * 7 + col * 3 + r * 13 + e * 5) % 17) - 8).to(torch.bfloat16)

    def expected(e: int, rows: int) -> torch.Tensor:
        acc = torch.zeros(rows, hidden, dtype=torch.float32, device=device)
        for r in range(world):
            acc += pattern(r, e, rows).float()
        return acc

    # concurrent HBM traffic on another stream for the whole run
    load = torch.cuda.Stream(device=device)
    src = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    main = torch.cuda.current_stream(device)
    eager_epochs = epochs - epochs // 5
    with torch.cuda.stream(load):
        for _ in range(max(4, eager_epochs // 8)):
            dst.copy_(src, non_blocking=True)
    for e in range(eager_epochs):
        rows = rnd.choice((1, 2, 3, cap, max(1, cap // 2), rnd.randint(1, cap)))
        kind = e % 3
        x = pattern(rank, e, rows)
        want = expected(e, rows)
        if kind == 0:
            got = c.all_reduce(x.clone())
            bad += (got.float() != want).sum()
        elif kind == 1 and c.fits(rows, hidden):
            buf = c.input_buffer(rows, hidden, device)
            torch.mm(x, eye, out=buf)                           # a GEMM writes the partial sums into the shared region
            got = c.all_reduce(buf, out=torch.empty_like(x))
            bad += (got.float() != want).sum()
        else:
            res = torch.zeros(rows, hidden, dtype=torch.bfloat16, device=device)
            y = c.all_reduce_add_rmsnorm(x.clone(), res, ones, 1e-6)
            bad += (res.float() != want).sum()                  # residual = bf16(sum + 0): exact
            ref = want * torch.rsqrt(want.pow(2).mean(-1, keepdim=True) + 1e-6)
            bad += ((y.float() - ref).abs() > 0.05 * ref.abs() + 0.02).sum()
    # one captured graph (all-reduce + fused norm), replayed with new inputs
    rows = min(cap, 37)
    xs = torch.zeros(rows, hidden, dtype=torch.bfloat16, device=device)
    res = torch.zeros_like(xs)
    out1 = torch.empty_like(xs)
    c.all_reduce(xs, out=out1)                                  # warm-up outside the capture
    c.all_reduce_add_rmsnorm(xs, res, ones, 1e-6)
    torch.cuda.synchronize(device)
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        c.all_reduce(xs, out=out1)
        y = c.all_reduce_add_rmsnorm(xs, res, ones, 1e-6)
    for e in range(eager_epochs, epochs):
        xs.copy_(pattern(rank, e, rows))
        res.zero_()
        graph.replay()
        want = expected(e, rows)
        bad += (out1.float() != want).sum() + (res.float() != want).sum()
    torch.cuda.synchronize(device)
    main.wait_stream(load)
    ok = int(bad.item()) == 0
    try:
        c.status()
    except Exception:  # noqa: BLE001 - a latched spin timeout
        ok = False
    del graph
    return ok


def shutdown() -> None:
    global _comm, _rank, _size, _backend, _group_of_one, _handoff_report
    if _comm is not None:
        _comm.close()
        _comm = None
    _rank, _size, _backend, _group_of_one, _handoff_report = 0, 1, "", False, ""


# ------------------------------------------------------------------------------------------------
def group_all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM) -> torch.Tensor:
    """In-place all-reduce through the process group (RCCL; gloo stages device tensors through the host
    itself, but has no bf16 device path, so those go through a float copy)."""
    if not _active():
        return t
    if _backend == "gloo" and t.is_cuda:
        h = t.float().cpu() if t.dtype == torch.bfloat16 else t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h.to(t.dtype))
        return t
    if _backend == "nccl" and not t.is_cuda:
        d = t.cuda()
        dist.all_reduce(d, op=op)
        t.copy_(d.cpu())
        return t
    dist.all_reduce(t, op=op)
    return t


def all_reduce(t: torch.Tensor) -> torch.Tensor:
    """Sum over the tensor-parallel ranks, in place."""
    if not _active():
        return t
    if _comm is not None and t.is_cuda and t.dim() == 2 and t.is_contiguous() and t.dtype == torch.bfloat16 \
            and _comm.fits(t.shape[0], t.shape[1]):
        return _comm.all_reduce(t)
    return group_all_reduce(t)


def all_reduce_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """y = RMSNorm.add_rms_forward(all_reduce(x), residual) (residual updated in place): one P2P launch when
    the message fits the comm buffer, all-reduce + nvl_add_rmsnorm otherwise."""
    from . import ops
    if _comm is not None and x.is_contiguous() and _comm.fits(x.shape[0], x.shape[1]):
        return _comm.all_reduce_add_rmsnorm(x, residual, weight, eps)
    return ops.add_rmsnorm(all_reduce(x), residual, weight, eps)


def all_gather_small(t: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[world, ...] <- every rank's t (a few KB: the sampler's per-shard winners)."""
    nbytes = t.numel() * t.element_size()
    if _comm is not None and nbytes <= 4096 and nbytes % 16 == 0:
        return _comm.all_gather(t, out)
    if _backend == "gloo" and t.is_cuda:
        parts = [torch.empty(t.shape, dtype=t.dtype, device="cpu") for _ in range(_size)]
        dist.all_gather(parts, t.cpu())
        out.copy_(torch.stack(parts).view(out.shape))
        return out
    dist.all_gather_into_tensor(out.view(-1), t.reshape(-1))
    return out


def side_stream() -> torch.cuda.Stream:
    global _side_stream
    if _side_stream is None:
        _side_stream = torch.cuda.Stream()
    return _side_stream


def linear_allreduce(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """y = all_reduce(x @ weight.T (+ bias)); chunk-pipelined over tokens for large inputs: GEMM(chunk i+1)
    runs on the compute stream while RCCL reduces chunk i on a side HIP stream."""
    F = torch.nn.functional
    if not _active():
        return F.linear(x, weight, bias)
    n = x.shape[0]
    capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
    if (not x.is_cuda) or capturing or n < _OVERLAP_MIN_TOKENS or _backend != "nccl":
        return all_reduce(F.linear(x, weight, bias))
    y = torch.empty((n, weight.shape[0]), dtype=x.dtype, device=x.device)
    main = torch.cuda.current_stream()
    side = side_stream()
    step = -(-n // _OVERLAP_CHUNKS)
    done = []
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        torch.mm(x[lo:hi], weight.t(), out=y[lo:hi]) if bias is None else y[lo:hi].copy_(F.linear(x[lo:hi], weight, bias))
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            dist.all_reduce(y[lo:hi])            # RCCL on the side stream, overlaps the next GEMM
            ev = torch.cuda.Event()
            ev.record(side)
            done.append(ev)
    for ev in done:
        main.wait_event(ev)
    return y
