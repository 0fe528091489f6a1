"""Tensor-parallel plumbing: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm).

The reference all-reduces inline on the compute stream after every row-parallel GEMM
(nano-vllm layers/linear.py:153-156, embed_head.py:41). On MI355X the 8 GPUs are fully
connected by point-to-point xGMI links, so a prefill-sized all-reduce (tens of MB) is worth
hiding: `allreduce_overlapped` splits the token dimension into chunks, runs GEMM(chunk i+1) on
the compute stream while RCCL reduces chunk i on a side HIP stream, and joins with events.
Decode-sized messages (a few hundred KB, latency-bound) go straight through RCCL on the
compute stream so that they can be captured in the decode hipGraph.
"""
from __future__ import annotations

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


def init(rank: int, size: int) -> None:
    """Declare this process' place in the engine's tensor-parallel group (size 1 = no TP). With
    size > 1 the default torch.distributed group must be that group (engine/runner.py creates it)."""
    global _rank, _size
    assert 0 <= rank < size
    assert size == 1 or (dist.is_initialized() and dist.get_world_size() == size), "TP group not initialised"
    _rank, _size = rank, size


def world() -> tuple[int, int]:
    """(rank, world_size) of the engine's tensor-parallel group; (0, 1) without TP."""
    return _rank, _size


def all_reduce(t: torch.Tensor) -> torch.Tensor:
    if _size > 1:
        dist.all_reduce(t)
    return t


def side_stream() -> torch.cuda.Stream:
    global _side_stream
    if _side_stream is None:
        _side_stream = torch.cuda.Stream()
    return _side_stream


def linear_allreduce(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """y = all_reduce(x @ weight.T (+ bias)); chunk-pipelined over tokens for large inputs."""
    F = torch.nn.functional
    _, size = world()
    if size == 1:
        return F.linear(x, weight, bias)
    n = x.shape[0]
    capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
    if (not x.is_cuda) or capturing or n < _OVERLAP_MIN_TOKENS:
        y = F.linear(x, weight, bias)
        dist.all_reduce(y)
        return y
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
