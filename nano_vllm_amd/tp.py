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

The P2P path is verified against the process group's own all-reduce when the group is set up
(`init_p2p`); if the check fails or IPC mapping is unavailable the engine falls back to the process group
for everything and says so. `NVL_TP_P2P=0` forces that fallback. With backend "gloo" (the functional test
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


def init(rank: int, size: int) -> None:
    """Declare this process' place in the engine's tensor-parallel group (size 1 = no TP). With
    size > 1 the default torch.distributed group must be that group (engine/runner.py creates it)."""
    global _rank, _size, _backend, _comm
    assert 0 <= rank < size
    assert size == 1 or (dist.is_initialized() and dist.get_world_size() == size), "TP group not initialised"
    if _comm is not None:
        _comm.close()
        _comm = None
    _rank, _size = rank, size
    _backend = dist.get_backend() if size > 1 else ""


def world() -> tuple[int, int]:
    """(rank, world_size) of the engine's tensor-parallel group; (0, 1) without TP."""
    return _rank, _size


def comm():
    return _comm


def capturable() -> bool:
    """Can a decode step's collectives be recorded into a hipGraph?"""
    return _size == 1 or _comm is not None or _backend == "nccl"


# ------------------------------------------------------------------------------------------------
def _exchange_bytes(blob: bytes) -> list[bytes]:
    out = [None] * _size
    dist.all_gather_object(out, blob)
    return out


def init_p2p(max_rows: int, hidden: int, device: torch.device) -> bool:
    """Collective. Set up the xGMI P2P collectives for messages up to [max_rows, hidden] bf16 and verify
    them against the process group. Returns whether they are in use."""
    global _comm
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
    flag = torch.tensor([1 if good else 0], dtype=torch.int32, device="cpu")
    group_all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        c.close()
        warnings.warn(f"nano_vllm_amd: xGMI P2P collectives FAILED their self-check; using {_backend}")
        return False
    _comm = c
    return True


def shutdown() -> None:
    global _comm, _rank, _size, _backend
    if _comm is not None:
        _comm.close()
        _comm = None
    _rank, _size, _backend = 0, 1, ""


# ------------------------------------------------------------------------------------------------
def group_all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM) -> torch.Tensor:
    """In-place all-reduce through the process group (RCCL; gloo stages device tensors through the host
    itself, but has no bf16 device path, so those go through a float copy)."""
    if _size == 1:
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
    if _size == 1:
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
    if _size == 1:
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
