"""Headline benchmark: output tokens/s on the reference's bench.py workload
(GeeeekExplorer/nano-vllm bench.py:9-28 — seed(0), 256 sequences, prompt and output lengths
U[100,1024], temperature 0.6, ignore_eos, Qwen3-0.6B, max_model_len 4096), run through the
drop-in `nanovllm.LLM.generate()` on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full pass of the 256-sequence workload (142,827 prompt tokens in, 133,966
tokens out). Weights are synthetic (seeded random, Qwen3-0.6B shapes, generated on device):
there are no checkpoints offline. N>1 runs N independent engine replicas, one per GPU (the
sequences are independent: data-parallel, weak scaling, no data-path collective); the reference's
tensor-parallel mode is `tensor_parallel_size` of the engine itself.

Rank 0 prints ONE JSON line. Besides the driver's fields it carries
  roofline     : the dominant kernel (paged decode attention, HBM-bound). achieved = algorithmic
                 bytes per launch (sum_b len_b * 2 * Hkv * 128 * 2 B) / mean launch time, measured
                 with HIP events on the launch stream by replaying decode batches recorded during
                 the timed pass (every 8th step: real context lengths and block tables, all 28
                 layer caches) — launches inside the captured hipGraph cannot be bracketed
                 individually. The rocprofv3 kernel-trace summary of this command is in profiles/.
  cpu_baseline : the CPU oracle (oracle/engine.py, a port of the reference's path) timed on this
                 box's host cores on a bounded sample of the same seeded workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from random import randint, seed

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REF_4070_LAPTOP_TOKS = 1434.13     # the reference's own number for this workload (other hardware)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--num-seqs", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="enforce_eager=True (no hipGraph)")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.9)
    return ap.parse_args()


def workload(num_seqs: int):
    """bench.py:9-18 of the reference, verbatim semantics."""
    seed(0)
    prompts = [[randint(0, 10000) for _ in range(randint(100, 1024))] for _ in range(num_seqs)]
    outs = [randint(100, 1024) for _ in range(num_seqs)]
    return prompts, outs


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Functional check of the N>1 path on a 1-GPU box (not a measurement): NVL_BENCH_SHARE_GPU=1 puts every
    # rank on GPU 0 and NVL_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    share_gpu = os.environ.get("NVL_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("NVL_BENCH_BACKEND", "nccl")
    if world > 1 and not share_gpu:
        # one replica per GPU: each process sees only its own device
        os.environ["HIP_VISIBLE_DEVICES"] = str(local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    from nano_vllm_amd import build as nvl_build
    if rank == 0 or world == 1:
        nvl_build.build()
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank,
                                    device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
        dist.barrier()

    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from nanovllm import LLM, SamplingParams

    path = os.path.join(tempfile.gettempdir(), f"nvl_{args.model}_r{rank}")
    write_synthetic_checkpoint(path, args.model, with_weights=False)
    llm = LLM(path, enforce_eager=args.eager, max_model_len=4096, dummy_weights=True,
              gpu_memory_utilization=args.gpu_memory_utilization)

    prompts, out_lens = workload(args.num_seqs)
    sps = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=m) for m in out_lens]
    total_out = sum(out_lens)

    # ---- record decode batches of a pass (for the roofline replay) --------------------------
    runner = llm.model_runner
    rec = {"on": False, "steps": 0, "ctx_tokens": 0, "samples": []}
    orig_prepare = runner.prepare_decode

    def prepare_spy(seqs, *a):
        n = orig_prepare(seqs, *a)
        if rec["on"]:
            st = runner.dstage.np
            rec["ctx_tokens"] += int(st["ctx"][:n].sum())
            if rec["steps"] % 8 == 0:
                rec["samples"].append((n, st["ctx"][:n].copy(), st["bt"][:n].copy()))
            rec["steps"] += 1
        return n

    runner.prepare_decode = prepare_spy
    orig_prepare_prefill = runner.prepare_prefill

    def prepare_prefill_spy(seqs):
        info = orig_prepare_prefill(seqs)
        if rec["on"] and not info["paged"]:
            st = runner.pstage.np
            rec.setdefault("prefill", []).append(st["cu_q"][:info["ns"] + 1].copy())
        return info

    runner.prepare_prefill = prepare_prefill_spy

    # ---- host-side time of the timed pass. With the decode lookahead (engine/core.py) schedule, postprocess
    #      and prepare_decode of step N+1 run while the GPU executes step N; only fill_tokens is serial. -----
    host = {"schedule_s": 0.0, "postprocess_s": 0.0, "prepare_decode_s": 0.0}

    def timed(fn, key):
        def wrapper(*a, **kw):
            t = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                if rec["on"]:
                    host[key] += time.perf_counter() - t
        return wrapper

    llm.scheduler.schedule = timed(llm.scheduler.schedule, "schedule_s")
    llm.scheduler.postprocess = timed(llm.scheduler.postprocess, "postprocess_s")     # (postprocess_early calls it)
    llm.scheduler.fill_tokens = timed(llm.scheduler.fill_tokens, "postprocess_s")
    runner.prepare_decode = timed(runner.prepare_decode, "prepare_decode_s")

    llm.generate(["Benchmark: "], SamplingParams(), use_tqdm=False)          # reference bench.py:22
    # Every pass starts COLD, like the reference's single timed generate(): without the reset, pass k+1 would
    # serve the full 256-token blocks of the same prompts from the prefix cache pass k left behind.
    for _ in range(args.warmup):
        llm.reset_prefix_cache()
        llm.generate(prompts, sps, use_tqdm=False)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        rec["on"] = (k == args.steps - 1) and not args.no_roofline
        llm.reset_prefix_cache()
        llm.generate(prompts, sps, use_tqdm=False)
    sync_all()
    elapsed = time.perf_counter() - t0
    rec["on"] = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = {
        "metric": "output tokens/s (bench.py, 256 seqs) Qwen3-0.6B TP=1" if args.model == "qwen3-0.6b" else
                  f"output tokens/s (bench.py, {args.num_seqs} seqs) {args.model} TP=1",
        "value": total_out * args.steps * world / elapsed,
        "unit": "tok/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": f"synthetic (seeded random weights, {args.model} shapes; token ids randint(0,10000) as reference bench.py)",
        "config": {"workload": f"nano-vllm bench.py: {args.num_seqs} seqs, in/out U[100,1024], T=0.6, ignore_eos, "
                               f"max_model_len 4096, {args.model}", "parallelism": f"dp{world} (1 engine replica per GPU, TP=1)",
                   "hipgraph": not args.eager, "kv_blocks": llm.config.num_kvcache_blocks,
                   "output_tokens_per_step": total_out},
    }
    if args.num_seqs == 256 and args.model == "qwen3-0.6b":
        result["vs_baseline"] = result["value"] / REF_4070_LAPTOP_TOKS
        result["config"]["baseline_note"] = "vs_baseline = value / 1434.13 tok/s (reference README, RTX 4070 Laptop: other hardware)"

    if rank == 0 and not args.no_roofline and rec["samples"]:
        result["config"]["host_seconds_in_last_step"] = {k: round(v, 4) for k, v in host.items()}
        result["roofline"] = roofline_replay(torch, runner, rec)
        if rec.get("prefill"):
            result["roofline_prefill"] = prefill_replay(torch, runner, rec["prefill"])
    if rank == 0 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(torch, llm, args.model, prompts, out_lens)
        except Exception as ex:  # noqa: BLE001 — a reported baseline must never sink the bench line
            result["cpu_baseline"] = {"error": repr(ex)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    llm.exit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roofline_replay(torch, runner, rec) -> dict:
    """Time the decode-attention kernel alone on the recorded batches (HIP events on the launch
    stream, every layer's cache => cold K/V like in the real step). Same code path as
    tools/attn_replay.py, which is the command the rocprofv3 duration/PMC profiles are taken with."""
    from tools.attn_replay import replay
    geo = runner.geo
    hq, hkv, L = geo["heads"], geo["kv_heads"], geo["layers"]
    r = replay(torch, runner.kv_cache, rec["samples"], hq, hkv, runner.config.max_model_len, runner.decode_ws, fused=True)
    achieved = r["achieved_GBps"]
    step_bytes = rec["ctx_tokens"] * 2 * hkv * 128 * 2 * L
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": pmc_traffic(r["algorithmic_bytes_per_launch"]),
            "kernel": (f"decode_stream_kernel<{hq // hkv}, fused>" if hq // hkv != 8 else "decode_mfma8_kernel<fused>")
                      + " + decode_stream_combine_kernel (nvl_paged_attn_decode_fused: the launch the decode step makes)",
            "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "avg_launch_us": r["avg_launch_us"],
            "launches_timed": r["launches_timed"], "decode_steps_in_pass": rec["steps"],
            "kv_bytes_read_in_pass": step_bytes, "frac_of_measured_achievable_6.29TBps": achieved / 6290.0}


def prefill_replay(torch, runner, batches) -> dict:
    """MFMA utilisation of the prefill-attention kernel on the prefill batches of the timed pass (their real
    cu_seqlens; random q/k/v of the model's head geometry), HIP events on the launch stream.
    FLOPs = 4 * Hq * 128 * causal (query, key) pairs — the algorithmic count, masked work not credited."""
    from nano_vllm_amd import ops
    geo = runner.geo
    hq, hkv = geo["heads"], geo["kv_heads"]
    dev = runner.device
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    flops, ms, launches = 0.0, 0.0, 0
    for cu in batches:
        n = int(cu[-1])
        lens = (cu[1:] - cu[:-1]).astype("int64")
        q = torch.randn(n, hq, 128, device=dev, dtype=torch.bfloat16)
        k = torch.randn(n, hkv, 128, device=dev, dtype=torch.bfloat16)
        v = torch.randn(n, hkv, 128, device=dev, dtype=torch.bfloat16)
        cu_d = torch.from_numpy(cu.copy()).to(dev)
        out = torch.empty_like(q)
        for _ in range(2):
            start.record()
            for _ in range(4):
                ops.attn_prefill_varlen(q, k, v, cu_d, cu_d, int(lens.max()), 128 ** -0.5, out=out)
            stop.record()
            torch.cuda.synchronize()
        ms += start.elapsed_time(stop)
        launches += 4
        flops += 4 * 4.0 * hq * 128 * float((lens * (lens + 1) // 2).sum())
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0,
            "traffic": None, "kernel": "prefill_attn_kernel<false> (nvl_attn_prefill_varlen)",
            "avg_launch_us": ms * 1e3 / launches, "launches_timed": launches,
            "note": "prefill batches of the timed pass (up to 16,384 tokens of 100-1024-token prompts per launch)"}


def pmc_traffic(alg_bytes_per_launch: float):
    """HBM bytes per launch from the committed rocprofv3 PMC pass (profiles/pmc_traffic.json:
    FETCH_SIZE summed over the decode_stream_kernel launches of tools/attn_replay.py, doubled as
    MI355X_MICROARCH.md §HBM prescribes for 16 B/lane streaming reads on gfx950, divided by the
    algorithmic bytes of the same launches). PMC counters cannot be read from inside this process,
    so the ratio measured by that separate pass is applied to this run's bytes per launch; null
    when no PMC pass has been recorded."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            ratio = float(json.load(fh)["hbm_read_bytes_over_algorithmic"])
    except (OSError, KeyError, ValueError):
        return None
    return ratio * alg_bytes_per_launch


def cpu_baseline(torch, llm, model_name, prompts, out_lens) -> dict:
    """The CPU oracle (a port of the reference's path: oracle/engine.py + oracle/model.py) on a
    bounded sample: the first 8 sequences of the same seeded stream, outputs capped at 16 tokens."""
    from nano_vllm_amd.weights import parameter_shapes, qwen3_config_dict, synth_tensor
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    cfg = qwen3_config_dict(model_name)
    dev = llm.model_runner.device
    weights = {n: synth_tensor(n, s, llm.config.seed, device=dev).cpu() for n, s in parameter_shapes(cfg).items()}
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    n_seq, cap = 8, 16
    sample_p = prompts[:n_seq]
    sample_o = [min(m, cap) for m in out_lens[:n_seq]]
    eng = OracleEngine(OracleQwen3(cfg, weights, compiled=True), num_blocks=64, block_size=256)
    t0 = time.perf_counter()
    eng.generate(sample_p, temperature=0.6, max_tokens=sample_o, ignore_eos=True)
    dt = time.perf_counter() - t0
    return {"value": sum(sample_o) / dt, "unit": "tok/s", "cores": threads, "kind": "port",
            "sample": f"first {n_seq} sequences of the seeded bench stream ({sum(len(p) for p in sample_p)} prompt "
                      f"tokens), outputs capped at {cap} tokens each ({sum(sample_o)} tokens), {dt:.1f} s; "
                      f"host has {cores} logical CPUs, torch threads={threads}"}


if __name__ == "__main__":
    main()
