"""Replay the decode-attention launches of the bench workload WITHOUT the model.

The host scheduler + block manager (pure Python) are driven through the reference bench.py
workload (seed(0), 256 sequences, in/out U[100,1024]) with a fake token source, which yields the
exact decode batches of the real run (context lengths and block tables depend only on the
schedule, not on token values: ignore_eos). Every `--every`-th decode batch is then replayed
through `nvl_paged_attn_decode` over all 28 layer caches (cold K/V, like in the real step) and
timed with HIP events on the launch stream.

This is the command the per-kernel profiles in profiles/ are taken with:

  rocprofv3 --kernel-trace --stats ... -- python tools/attn_replay.py          (duration)
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex decode_ ... -- python tools/attn_replay.py
                                                                              (HBM traffic)

Prints one JSON line; `algorithmic_bytes_per_launch` is sum_b len_b * 2 * Hkv * 128 * 2 B.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from random import randint, seed
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench_decode_batches(num_seqs: int = 256, num_blocks: int = 9380, every: int = 8, max_num_seqs: int = 512,
                         max_num_batched_tokens: int = 16384, block_size: int = 256, max_model_len: int = 4096,
                         kind: str = "bench"):
    """Host-only pass of the bench workload (`kind` "prefix": BASELINE config 3 — a 512-token system prompt + U[16,256]
    suffix x 256, 128 output tokens, bench.py::workload). Returns (samples, stats): samples = list of
    (n, ctx int32[n], block_table int32[n, max_blocks]) for every `every`-th decode step."""
    from nano_vllm_amd.api import SamplingParams
    from nano_vllm_amd.engine.sched import Scheduler
    from nano_vllm_amd.engine.seq import Sequence
    seed(0)
    if kind == "prefix":
        system = [randint(0, 10000) for _ in range(512)]
        prompts = [system + [randint(0, 10000) for _ in range(randint(16, 256))] for _ in range(num_seqs)]
        outs = [128] * num_seqs
    else:
        prompts = [[randint(0, 10000) for _ in range(randint(100, 1024))] for _ in range(num_seqs)]
        outs = [randint(100, 1024) for _ in range(num_seqs)]
    cfg = SimpleNamespace(max_num_seqs=max_num_seqs, max_num_batched_tokens=max_num_batched_tokens, eos=-1,
                          kvcache_block_size=block_size, num_kvcache_blocks=num_blocks)
    Sequence.block_size = block_size
    sched = Scheduler(cfg)
    for p, m in zip(prompts, outs):
        sched.add(Sequence(p, SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=m)))
    w = max_model_len // block_size
    samples, steps, prefills, ctx_tokens, max_block = [], 0, 0, 0, 0
    while not sched.is_finished():
        batch, is_prefill = sched.schedule()
        if is_prefill:
            prefills += 1
        else:
            n = len(batch)
            lens = np.fromiter((s.num_tokens for s in batch), dtype=np.int32, count=n)
            ctx_tokens += int(lens.sum())
            if steps % every == 0:
                bt = np.full((n, w), -1, dtype=np.int32)
                for i, s in enumerate(batch):
                    bt[i, :len(s.block_table)] = s.block_table
                    max_block = max(max_block, max(s.block_table))
                samples.append((n, lens, bt))
            steps += 1
        sched.postprocess(batch, [1] * len(batch), is_prefill)
    return samples, dict(decode_steps=steps, prefill_steps=prefills, ctx_tokens=ctx_tokens, max_block=max_block)


def replay(torch, kv_cache, samples, hq: int, hkv: int, max_ctx: int, ws, reps: int = 2, fused: bool = False,
           plan: bool = True, graph: bool = True, shared_blocks_of=None):
    """Time nvl_paged_attn_decode (main kernel + split combine) on the recorded batches.
    kv_cache: [2, L, num_blocks, Hkv, block, 128]. Block ids are folded into the cache with a
    modulo when the cache is smaller than the pool the schedule was recorded with.
    plan: as the engine does, one nvl_decode_plan per batch (outside the timed bracket: it is one ~4 us launch per
    STEP, not per layer) shared by the L layer launches.
    shared_blocks_of(bt, lens, n) -> (k, member) (the engine's ModelRunner._prefix_group_worth_a_pass): batches in which
    a group of rows starts with the same KV blocks are replayed WITH the shared-prefix pass, as the engine runs them. The
    bytes such a batch is credited with are the UNIQUE ones — the common blocks once, not once per member — so the achieved
    rate stays a rate of bytes that had to come from HBM; `per_sequence_bytes_per_launch` is the reference's figure
    (flash_attn_with_kvcache reads every sequence's whole table, layers/attention.py:72-74)."""
    from nano_vllm_amd import ops
    L, nblk = kv_cache.shape[1], kv_cache.shape[2]
    dev = kv_cache.device
    scale = 128 ** -0.5
    max_bs = max(n for n, _, _ in samples)
    q_all = torch.randn(max_bs, hq, 128, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q_all)
    if fused:       # the decode step's real entry point: raw qkv rows in, q/k-norm + RoPE + KV store inside
        qkv_all = torch.randn(max_bs, (hq + 2 * hkv) * 128, device=dev, dtype=torch.bfloat16)
        nw = torch.ones(128, device=dev, dtype=torch.bfloat16)
        inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2, device=dev).float() / 128))
        fr = torch.arange(40960, device=dev).float()[:, None] * inv[None]
        table = torch.cat([fr.cos(), fr.sin()], -1).contiguous()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total_ms, total_bytes, launches, per_seq_bytes, px_launches = 0.0, 0, 0, 0, 0
    block_size = kv_cache.shape[4]
    for n, ctx, bt in samples:
        if bt.max() >= nblk:
            bt = np.where(bt >= 0, bt % nblk, bt).astype(np.int32)
        ctx_d = torch.from_numpy(np.ascontiguousarray(ctx)).to(dev)
        bt_d = torch.from_numpy(np.ascontiguousarray(bt)).to(dev)
        q = q_all[:n]
        shared, member = shared_blocks_of(bt, np.asarray(ctx, dtype=np.int64), n) if (plan and shared_blocks_of) else (0, None)
        shp = torch.tensor([shared, *member.astype(np.int32).tolist()], dtype=torch.int32, device=dev) if shared > 0 else None
        groups = 1 if shared == 0 or int(member.max()) <= 1 else 4       # (as the engine: one slot, or MAX_PREFIX_GROUPS)
        step_plan = ops.decode_plan(ctx_d, hq, hkv, max_ctx, shared_prefix=shp, block_size=block_size, prefix_groups=groups) if plan else None

        def layers():
            for layer in range(L):
                if fused:
                    ops.paged_attn_decode_fused(qkv_all[:n], nw, nw, 1e-6, table, kv_cache[0, layer], kv_cache[1, layer],
                                                bt_d, ctx_d, hq, scale, max_ctx, ws, out=out[:n], plan=step_plan)
                else:
                    ops.paged_attn_decode(q, kv_cache[0, layer], kv_cache[1, layer], bt_d, ctx_d, scale, max_ctx, ws,
                                          out=out[:n], plan=step_plan)

        # The L launches of a step are captured into ONE hipGraph and the replay is timed (as the engine runs them:
        # inside its captured decode step) — issued eagerly from Python, two short launches per layer are host-bound
        # (3-4 us of launch cost each: the one-kv-head shape measured 33 us per call eagerly against 19 + 4.5 us of
        # kernel time under rocprofv3) and the bracket would time the host, not the kernel.
        runner = layers
        if graph:
            layers()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # (inference mode: inside bench.py the engine has captured its own graphs under it, and torch's capture
            #  bookkeeping then touches generator state that only inference mode may update)
            with torch.inference_mode(), torch.cuda.graph(g):
                layers()
            runner = g.replay
        for _ in range(reps):                       # keep the last rep (caches are 100s of MB: nothing stays warm)
            start.record()
            runner()
            stop.record()
            torch.cuda.synchronize()
        total_ms += start.elapsed_time(stop)
        tok_bytes = 2 * hkv * 128 * kv_cache.element_size() * L
        per_seq_bytes += int(ctx.sum()) * tok_bytes
        total_bytes += (int(ctx.sum()) - (shared * block_size * (int(member.sum()) - 1) if shared > 0 else 0)) * tok_bytes
        launches += L
        px_launches += L if shared > 0 else 0
    return dict(achieved_GBps=total_bytes / (total_ms * 1e-3) / 1e9, algorithmic_bytes_per_launch=total_bytes / launches,
                per_sequence_bytes_per_launch=per_seq_bytes / launches, launches_with_shared_prefix_pass=px_launches,
                avg_launch_us=total_ms * 1e3 / launches, launches_timed=launches, reps=reps,
                launched="one captured hipGraph per step (L layer launches), replayed" if graph else "eagerly from Python")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--every", type=int, default=8)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--hq", type=int, default=16)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--pool-blocks", type=int, default=9380, help="KV pool the schedule is recorded with")
    ap.add_argument("--fused", action="store_true", help="time nvl_paged_attn_decode_fused (norm+rope+store inside)")
    ap.add_argument("--no-plan", action="store_true", help="every launch derives its own schedule (no nvl_decode_plan)")
    ap.add_argument("--eager", action="store_true", help="issue the launches eagerly from Python instead of replaying one "
                                                         "captured graph per step (host-bound for short launches)")
    ap.add_argument("--layer-major", action="store_true", help="cache laid out [L, 2, blocks, ...] instead of [2, L, blocks, ...]")
    ap.add_argument("--fp8", action="store_true", help="OCP fp8 e4m3 KV cache (opt-in extension; 128-byte rows)")
    ap.add_argument("--workload", default="bench", choices=["bench", "prefix"],
                    help="prefix: BASELINE config 3's schedule (shared 512-token system prompt; use with --hq 32 --hkv 8 --layers 36)")
    ap.add_argument("--no-shared-prefix", action="store_true",
                    help="replay batches that share leading KV blocks WITHOUT the shared-prefix pass (the engine's NVL_SHARED_PREFIX=0)")
    ap.add_argument("--cache-blocks", type=int, default=0,
                    help="allocate the cache with this many blocks per layer (like the engine's pool) instead of only the used ones")
    args = ap.parse_args()
    import torch
    from nano_vllm_amd import ops
    ops.load_library()
    samples, stats = bench_decode_batches(num_blocks=args.pool_blocks, every=args.every, kind=args.workload)
    used = stats["max_block"] + 1
    nblk = max(used, args.cache_blocks)
    dev = torch.device("cuda", 0)
    if args.layer_major:
        kv = torch.empty(args.layers, 2, nblk, args.hkv, 256, 128, dtype=torch.bfloat16, device=dev).transpose(0, 1)
    else:
        kv = torch.empty(2, args.layers, nblk, args.hkv, 256, 128, dtype=torch.bfloat16, device=dev)
    for layer in range(args.layers):               # random (not zero) data: zero-filled inputs clock higher
        kv[:, layer, :used].normal_()
    if args.fp8:
        kv8 = torch.empty(kv.shape, dtype=torch.float8_e4m3fn, device=dev)
        for layer in range(args.layers):
            kv8[:, layer, :used] = kv[:, layer, :used].to(torch.float8_e4m3fn)
        kv = kv8
    ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(512, args.hq, 4096), dtype=torch.uint8, device=dev)
    def group_worth_a_pass(bt, lens, n, min_bytes=160e6):
        # the engine's decision (ModelRunner._prefix_group_worth_a_pass) for this geometry
        from nano_vllm_amd.engine.runner import shared_prefix_group
        k, member = shared_prefix_group(bt[:n], lens, 256)
        if k == 0:
            return 0, None
        m, pack = int(member.sum()), 16 // (args.hq // args.hkv)
        saved = k * 256 * (m - -(-m // pack)) * args.hkv * 2 * 128 * (1 if args.fp8 else 2)
        return (k, member) if saved >= min_bytes else (0, None)

    shares = ops.decode_attention_shares_prefixes(args.hq, args.hkv, 256) and not args.no_shared_prefix and not args.no_plan
    r = replay(torch, kv, samples, args.hq, args.hkv, 4096, ws, reps=args.reps, fused=args.fused, plan=not args.no_plan,
               graph=not args.eager, shared_blocks_of=group_worth_a_pass if shares else None)
    r.update(stats, kernel=f"decode<G={args.hq // args.hkv}, fused={str(args.fused).lower()}, kv={'fp8' if args.fp8 else 'bf16'}>", kv_blocks_used=nblk, samples=len(samples),
             frac_of_8TBps=r["achieved_GBps"] / 8000.0)
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
