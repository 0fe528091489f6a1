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

Rank 0 prints ONE JSON line (kept under 10 KB: the prose that explains its fields is `NOTES` below, also written to
gpurun_out/bench_notes.json; the extras' full child lines go to gpurun_out/bench_extras_full.json). Besides the
driver's fields it carries
  roofline     : the dominant kernel (paged decode attention, HBM-bound). achieved = algorithmic
                 bytes per launch (sum_b len_b * 2 * Hkv * 128 * 2 B) / mean launch time, measured
                 with HIP events on the launch stream by replaying decode batches recorded during
                 the timed pass (every 8th step: real context lengths and block tables, all 28
                 layer caches): the engine's own step graph cannot be bracketed per kernel, so each
                 recorded step's 28 attention launches (main kernel + split merge) are captured into
                 a hipGraph of their own and the events bracket its replay. The rocprofv3
                 kernel-trace summary of this command is in profiles/.
  cpu_baseline : the CPU oracle (oracle/engine.py, a port of the reference's path) timed on this
                 box's host cores on a bounded sample of the same seeded workload (first 8 sequences x 9 tokens).
  parity       : the engine's own T = 0.6 tokens on that sample, judged exactly against the oracle's
                 race keys with the draws replayed (a by-product of the cpu_baseline leg: the timed
                 oracle pass is teacher-forced with the engine's tokens).
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
# a small OpenMP pool for the engine's host loop (before torch is imported): with the default of one thread per logical
# CPU (256 on the GPU boxes) the pool's spinning workers compete with the HIP runtime's own threads; the extras have
# always run like this (extra_configs), the headline measures the same either way (gpurun_out r05q: 35.29 / 35.19 k
# unset vs 35.30 / 35.19 k with 8). The CPU baseline leg sets its own thread count (cpu_baseline_prepare).
os.environ.setdefault("OMP_NUM_THREADS", "8")

BENCH_T0 = time.perf_counter()      # process start: the extras are budgeted against the whole run's wall time
REF_4070_LAPTOP_TOKS = 1434.13     # the reference's own number for this workload (other hardware)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


TP_MODELS = ("qwen3-32b", "qwen3-14b")     # models BASELINE.json quotes with tensor parallelism
# a TP = 8 / TP = 4 rank's vocabulary shard (embed_head.py:14-20)
MODEL_VOCAB = {"qwen3-32b-tp8rank": 151936 // 8, "qwen3-32b-tp4rank": 151936 // 4}
SIDE_DIR = os.path.join(ROOT, "gpurun_out")       # full child lines and the prose notes of a run (bench_*.json)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--tp", type=int, default=0,
                    help="tensor_parallel_size of the engine; 0 = auto: --gpus for Qwen3-32B (BASELINE's second metric), "
                         "1 otherwise (then --gpus N runs N data-parallel replicas)")
    ap.add_argument("--workload", default="bench", choices=["bench", "prefix", "long"],
                    help="bench = BASELINE config 2/4 (reference bench.py); prefix = config 3 (512-token shared system "
                         "prompt x 256 seqs); long = config 5 (16 x 16,000-token prompts, 64 output tokens)")
    ap.add_argument("--num-seqs", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-tp-extra", action="store_true",
                    help="with --gpus N > 1 on the default model: skip the additional Qwen3-32B TP=N measurement")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="with --gpus 1 on the default model / workload: skip the BASELINE config 3 (Qwen3-8B, shared "
                         "prefix) and config 5 (Qwen3-32B, 16 x 16,000-token prompts) passes attached as `extra_configs`")
    ap.add_argument("--max-num-seqs", type=int, default=0, help="engine max_num_seqs (0 = the reference default, 512)")
    ap.add_argument("--eager", action="store_true", help="enforce_eager=True (no hipGraph)")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.9)
    ap.add_argument("--num-kvcache-blocks", type=int, default=-1)
    ap.add_argument("--kv-cache-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: opt-in OCP e4m3 KV cache (outside the reference's numerics; never the headline)")
    return ap.parse_args()


def workload(num_seqs: int, kind: str = "bench"):
    """bench: bench.py:9-18 of the reference, verbatim semantics. prefix / long: SURVEY.md §8(d) configs 3 / 5."""
    seed(0)
    if kind == "prefix":
        system = [randint(0, 10000) for _ in range(512)]
        prompts = [system + [randint(0, 10000) for _ in range(randint(16, 256))] for _ in range(num_seqs)]
        return prompts, [128] * num_seqs
    if kind == "long":
        n = min(num_seqs, 16)
        return [[randint(0, 10000) for _ in range(16000)] for _ in range(n)], [64] * n
    prompts = [[randint(0, 10000) for _ in range(randint(100, 1024))] for _ in range(num_seqs)]
    outs = [randint(100, 1024) for _ in range(num_seqs)]
    return prompts, outs


def engine_kwargs(args, tp: int) -> dict:
    kw = dict(enforce_eager=args.eager, max_model_len=4096, dummy_weights=True, tensor_parallel_size=tp,
              gpu_memory_utilization=args.gpu_memory_utilization, num_kvcache_blocks=args.num_kvcache_blocks,
              kv_cache_dtype=args.kv_cache_dtype)
    if args.workload == "long":
        kw.update(max_model_len=32768, max_num_batched_tokens=16384)
    if args.max_num_seqs > 0:
        kw.update(max_num_seqs=args.max_num_seqs)
    return kw


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    tp = args.tp if args.tp > 0 else (max(args.gpus, world) if args.model in TP_MODELS else 1)
    external_tp = world > 1 and tp == world            # torchrun started one process per GPU: they ARE the TP ranks
    assert tp == 1 or world == 1 or external_tp, "--tp must be 1 or equal the number of launched ranks"
    # Functional check of the N>1 paths on a 1-GPU box (not a measurement): NVL_BENCH_SHARE_GPU=1 puts every
    # rank on GPU 0 and NVL_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    share_gpu = os.environ.get("NVL_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("NVL_BENCH_BACKEND", "nccl")
    if share_gpu:
        os.environ.setdefault("NVL_TP_SHARE_GPU", "1")
        if os.environ.get("NVL_BENCH_CU_SPLIT") == "1" and world > 1:
            # every rank gets a DISJOINT set of compute units of the one GPU (ROCr queue CU mask, read when the HSA
            # runtime starts — i.e. before torch touches the device): the ranks' spinning collective kernels can then
            # never occupy each other's wave slots. Used to tell starvation from a protocol bug (profiles/r05_*).
            per = 256 // world
            os.environ["HSA_CU_MASK"] = f"0:{local_rank * per}-{(local_rank + 1) * per - 1}"
    if backend != "nccl":
        os.environ.setdefault("NVL_TP_BACKEND", backend)
    if world > 1 and not share_gpu and not external_tp and args.no_tp_extra:
        # data-parallel replicas only: each process sees only its own device. (With the TP extra measurement, or
        # in TP mode, every GPU stays visible and rank r selects device r, as the reference does.)
        os.environ["HIP_VISIBLE_DEVICES"] = str(local_rank)
        local_rank = 0
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    from nano_vllm_amd import build as nvl_build
    if rank == 0 or world == 1:
        nvl_build.build()
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank,
                                    device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
        dist.barrier()

    pending_cpu = None
    if external_tp:
        result = run_tp_external(args, torch, dist, rank, world, tp)
    else:
        result, pending_cpu = run_replica(args, torch, dist, rank, world, tp, backend)
        if world > 1 and not args.no_tp_extra and args.model == "qwen3-0.6b" and args.workload == "bench":
            extra = tp_extra(args, torch, dist, rank, world, result)
            if rank == 0:
                result["tp_qwen3_32b"] = extra
    with_extras = (world == 1 and tp == 1 and not args.no_extra_configs and args.model == "qwen3-0.6b"
                   and args.workload == "bench" and args.num_seqs == 256 and args.kv_cache_dtype == "bf16")
    if pending_cpu is not None:
        # (sequential on purpose: run on a thread beside the extras' child engines, the oracle's 64 host threads slowed
        #  every one of them — the short-step ones by half, Qwen3-32B by a quarter. Hence the small sample.)
        pending_cpu()
    if with_extras:
        # The headline is SAFE before any extra starts: the complete line (without `extra_configs`) goes to stderr and
        # to gpurun_out/bench_headline.json now; stdout still carries exactly ONE JSON line, printed at the end, and
        # an extra that no longer fits the run's wall budget (NVL_BENCH_WALL_BUDGET, default 262 s) is skipped.
        keep_headline(result)
        result["extra_configs"] = extra_configs(args, torch)
    if rank == 0:
        if "tp_qwen3_32b" in result and isinstance(result["tp_qwen3_32b"], dict):
            attach_tp_scaling(args, torch, result)
        result["config"]["notes"] = write_notes(result)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def keep_headline(result: dict) -> None:
    line = json.dumps(result)
    print("[bench.py] headline line (extras follow; the one stdout line comes at the end): " + line, file=sys.stderr,
          flush=True)
    try:
        os.makedirs(SIDE_DIR, exist_ok=True)
        with open(os.path.join(SIDE_DIR, "bench_headline.json"), "w") as fh:
            fh.write(line + "\n")
    except OSError:
        pass


def attach_tp_scaling(args, torch, result: dict) -> None:
    """`--gpus N` on the default model: the TP = N Qwen3-32B value next to a same-run TP = 1 anchor is not available
    (the anchor needs a GPU to itself for ~50 s, and all N are busy with replicas / TP ranks), so the line points at the
    N = 1 line's `extra_configs.config4_anchor` of the same build; with NVL_BENCH_TP_ANCHOR=<tok/s> (that value) the
    ratio is printed here as well."""
    tpv = result["tp_qwen3_32b"].get("value")
    anchor = os.environ.get("NVL_BENCH_TP_ANCHOR")
    result["tp_qwen3_32b"]["tp1_anchor"] = "extra_configs.config4_anchor.value of the --gpus 1 line of this build"
    if tpv and anchor:
        result["tp_qwen3_32b"]["scaling_vs_tp1_anchor"] = tpv / float(anchor)


NOTES = {
    "roofline": "dominant kernel = paged decode attention (HBM-bound). achieved = algorithmic K/V bytes per launch (sum_b len_b "
                "* 2 * Hkv * 128 * elt) / mean duration of one attention call (planned main kernel + split merge), HIP events "
                "around a captured graph of every 8th recorded decode step's L launches (real lengths, real block tables, all "
                "layer caches). traffic = FETCH_SIZE x 2 ratio of the kernel's own rocprofv3 PMC pass (profiles/pmc_traffic.json, "
                "refused when attn_decode.hip changed since) x this run's bytes per launch. decode_step = every weight once per "
                "step + the K/V of every context token over the pass time minus its host-timed prefill steps.",
    "roofline_prefill": "prefill_attn_kernel on the prefill batches of the timed pass (their cu_seqlens, random q/k/v): 4 * Hq * "
                        "128 * causal pairs FLOP / HIP-event time; peak 2.5 PF dense bf16.",
    "cpu_baseline": "the CPU oracle (oracle/engine.py, a port of the reference's path: /root/reference does not exist on the GPU "
                    "box) on the first 8 sequences of the seeded stream, outputs capped at 9 tokens: one prefill step + 8 "
                    "decode steps at B = 8 (~25-30 s on <= 64 host threads), teacher-forced with the engine's tokens of the "
                    "same sample; nothing runs beside it.",
    "parity": "this build's engine on that sample at T = 0.6 (hipGraph decode, lookahead on): token == argmax(l/T - log E) on "
              "the oracle's logits wherever the key margin > 2 x floor / T, else within 2 x floor / T of the maximum; E = the "
              "engine's counter-based draw (seed, request, position, column) replayed by oracle/philox.py; floor = SURVEY.md's "
              "0.0195 x absmax. Attention boundary: flash-attn's source is absent from the reference tree: kernels are judged at "
              "2e-2 x absmax + |dLSE| <= 2e-3 against a restatement of its documented semantics (oracle/ops.py). T = 0 runs: "
              "tests/test_e2e_gpu.py, smoke().",
    "vs_baseline": "value / 1434.13 tok/s (the reference's README number on an RTX 4070 Laptop: other hardware, not credit).",
    "extra_configs": "one cold pass each (config 3: one warm-up pass) in a child bench.py after the headline line was written to "
                     "stderr and gpurun_out/bench_headline.json. config3 = Qwen3-8B, 512-token shared prefix x 256 seqs; config5 "
                     "= Qwen3-32B, 16 x 16,000-token prompts, 64 out, at TP = 1 (BASELINE quotes TP = 8); config4_anchor = "
                     "Qwen3-32B, bench workload, TP = 1: the denominator of a --gpus N line's tp_qwen3_32b; tp8_rank_shape_* / "
                     "tp4_rank_shape_bench = an engine on what ONE rank of TP = 8 / 4 holds (8/1 or 16/2 heads, intermediate "
                     "3200 / 6400, vocabulary / 8 or / 4; add-RMSNorm where the all-reduce would be): an UPPER BOUND per rank, "
                     "not a TP measurement. Fields: value tok/s, ms = ms per pass, attn = roofline.frac of the decode attention "
                     "call, step = decode step frac of 8 TB/s, pf_TF = prefill attention TFLOP/s, kv = KV blocks, wall = seconds "
                     "incl. engine start. config3 runs the shared-prefix attention pass (prefix-cache blocks shared by a group of "
                     "rows are read once per pack of 16/G rows): its attn / step are fractions in UNIQUE bytes (shared blocks "
                     "counted once — round 4's 0.90 / 0.54 were in per-sequence bytes), attn_ps = the same call priced in the "
                     "reference's per-sequence bytes / 8 TB/s, attn_us = us per attention call. Full child lines: "
                     "gpurun_out/bench_extras_full.json.",
    "tp_fallback": "a tensor-parallel run whose xGMI P2P collectives latch a spin timeout is re-run with NVL_TP_P2P=0 (process "
                   "group = RCCL) and BOTH attempts are reported; value is never null.",
}


def write_notes(result: dict) -> str:
    """The prose that explains the line's fields lives in a side file (the stdout line stays under 10 KB so that the
    driver's record keeps all of it); returns the path relative to the repo."""
    try:
        os.makedirs(SIDE_DIR, exist_ok=True)
        with open(os.path.join(SIDE_DIR, "bench_notes.json"), "w") as fh:
            json.dump({k: v for k, v in NOTES.items()}, fh, indent=1)
    except OSError:
        pass
    return "gpurun_out/bench_notes.json (= bench.py::NOTES)"


def metric_name(args, tp: int) -> str:
    shapes = {"qwen3-0.6b": "Qwen3-0.6B", "qwen3-8b": "Qwen3-8B", "qwen3-32b": "Qwen3-32B",
              "qwen3-32b-tp8rank": "Qwen3-32B per-rank shapes of TP=8 (no collectives), engine"}.get(args.model, args.model)
    if args.workload == "bench":
        return f"output tokens/s (bench.py, {args.num_seqs} seqs) {shapes} TP={tp}"
    return f"output tokens/s ({args.workload} workload) {shapes} TP={tp}"


def workload_note(args) -> str:
    return {"bench": f"nano-vllm bench.py: {args.num_seqs} seqs, in/out U[100,1024], T=0.6, ignore_eos, max_model_len 4096",
            "prefix": f"BASELINE config 3: 512-token shared system prompt + U[16,256] suffix x {args.num_seqs} seqs, 128 out, "
                      "T=0.6, ignore_eos (prefix-cache path)",
            "long": "BASELINE config 5: 16 x 16,000-token prompts, 64 out, max_model_len 32768, one prefill per step"
            }[args.workload] + f", {args.model}"


def timed_passes(args, torch, dist, llm, world, backend, rec=None, sync_group=True):
    """reference bench.py:22-28: warm-up generate, then K timed passes bracketed by barrier + synchronize."""
    from nanovllm import SamplingParams
    prompts, out_lens = workload(args.num_seqs, args.workload)
    sps = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=m) for m in out_lens]
    llm.generate(["Benchmark: "], SamplingParams(), use_tqdm=False)          # reference bench.py:22
    # Every pass starts COLD, like the reference's single timed generate(): without the reset, pass k+1 would
    # serve the full 256-token blocks of the same prompts from the prefix cache pass k left behind. (The prefix
    # workload keeps what it measures: sharing INSIDE a pass; the reset only removes sharing ACROSS passes.)
    for _ in range(args.warmup):
        llm.reset_prefix_cache()
        llm.generate(prompts, sps, use_tqdm=False)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1 and sync_group:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if rec is not None:
            rec["on"] = (k == args.steps - 1) and not args.no_roofline
        llm.reset_prefix_cache()
        llm.generate(prompts, sps, use_tqdm=False)
    sync_all()
    elapsed = time.perf_counter() - t0
    if rec is not None:
        rec["on"] = False
    return elapsed, prompts, out_lens


def base_result(args, tp, world_engines, n_gpus, elapsed, total_out, llm, parallelism):
    return {
        "metric": metric_name(args, tp),
        "value": total_out * args.steps * world_engines / elapsed,
        "unit": "tok/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if tp == 1 else "strong",
        "vs_baseline": None,
        "dtype": "bf16" if args.kv_cache_dtype == "bf16" else "bf16 (KV cache fp8 e4m3: opt-in, outside parity)",
        "data": f"synthetic (seeded random weights, {args.model} shapes; token ids randint(0,10000))",
        "config": {"workload": workload_note(args), "parallelism": parallelism,
                   "hipgraph": not llm.model_runner.enforce_eager, "kv_blocks": llm.config.num_kvcache_blocks,
                   "packed_weight_bytes": getattr(llm.model_runner, "packed_weight_bytes", None),
                   "projections_left_row_major_for_lack_of_budget": getattr(llm.model_runner, "packed_weight_skipped", None),
                   "output_tokens_per_step": total_out},
    }


def with_p2p_fallback(attempt, agree):
    """A tensor-parallel measurement that can never come back as `value: null` (the round-4 review's item 1d).
    `attempt(p2p)` runs one engine + timed pass on THIS rank and returns (result dict on rank 0 | None, latched) —
    `latched`: this rank's xGMI P2P communicator reported a spin timeout (nvl_allreduce_status), i.e. some collective of
    the pass gave up waiting for a peer and its result is invalid. `agree(flag)` = OR over the ranks (collective).
    If any rank latched, EVERY rank runs the pass again with the process group's collectives (NVL_TP_P2P=0: RCCL), and
    the line reports both: the fallback's value as `value`, the first attempt under `tp_p2p_attempt`."""
    first, latched = attempt(True)
    if not agree(bool(latched)):
        return first
    second, _ = attempt(False)
    if second is not None:
        second["tp_p2p_attempt"] = {
            "value_invalid": (first or {}).get("value"), "ms_per_step": (first or {}).get("ms_per_step"),
            "p2p_status": (lambda st: "latched on another rank (rank 0's own communicator was clean)" if st in (None, "ok") else st)(
                ((first or {}).get("config") or {}).get("p2p_status")),
            "p2p_handoff": ((first or {}).get("config") or {}).get("p2p_handoff"),
            "note": "a P2P collective latched a spin timeout in this attempt; `value` is the re-run with NVL_TP_P2P=0"}
        second["config"]["parallelism"] += " [fallback after a latched P2P spin timeout]"
    return second


def _is_latched(ex: BaseException) -> bool:
    """A latched P2P spin timeout: reported by the communicator's status at exit ("spin limit") or — since the latch travels
    to the host with every step's ids — raised from inside generate() ("gave up waiting for a peer", engine/runner.py)."""
    return type(ex).__name__ == "NvlError" and ("spin limit" in str(ex) or "gave up waiting for a peer" in str(ex))


def _exit_after_latch(llm) -> None:
    """Shut an engine down whose pass was aborted by a latched collective: the exit reports the same latch again."""
    try:
        llm.exit()
    except Exception as ex:  # noqa: BLE001
        if not _is_latched(ex):
            raise


def run_tp_external(args, torch, dist, rank, world, tp):
    """One engine, tensor-parallel over the `world` ranks torchrun started (BASELINE: Qwen3-32B TP=N): rank 0
    hosts the scheduler and posts steps, the other ranks execute them (LLMEngine.worker)."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from nanovllm import LLM
    path = os.path.join(tempfile.gettempdir(), f"nvl_{args.model}_r{rank}")
    write_synthetic_checkpoint(path, args.model, with_weights=False, vocab_size=MODEL_VOCAB.get(args.model, 151936))
    kw = engine_kwargs(args, tp)
    user_p2p = os.environ.get("NVL_TP_P2P", "1") != "0"

    def attempt(p2p: bool):
        os.environ["NVL_TP_P2P"] = "1" if (p2p and user_p2p) else "0"
        if rank > 0:
            try:
                LLM.worker(path, **kw)         # returns when rank 0 exits the engine
            except Exception as ex:  # noqa: BLE001 — the worker's exit raises its own communicator's latched status
                if not _is_latched(ex):
                    raise
                return None, True
            return None, False
        llm = LLM(path, **kw)
        try:
            elapsed, prompts, out_lens = timed_passes(args, torch, dist, llm, world, "nccl", sync_group=False)
        except Exception as ex:  # noqa: BLE001 — the latch surfaced inside generate(): no number from this attempt, re-run
            if not _is_latched(ex):
                raise
            _exit_after_latch(llm)
            return {"value": None, "ms_per_step": None, "config": {"p2p_status": repr(ex)}}, True
        result = base_result(args, tp, 1, world, elapsed, sum(out_lens), llm,
                             f"tp{tp} (one engine, tensor-parallel over {tp} GPUs: xGMI P2P all-reduce "
                             f"{'on' if llm.model_runner.p2p else 'OFF (process group: ' + ('RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()) + ')'})")
        from nano_vllm_amd import tp as tp_mod
        result["config"]["p2p_handoff"] = tp_mod.handoff_report()
        latched = False
        try:
            on = llm.model_runner.p2p
            llm.exit()
            result["config"]["p2p_status"] = "ok" if on else "n/a (process group)"
        except Exception as ex:  # noqa: BLE001 — a latched collective timeout invalidates the number: REPORTED + re-run
            if not _is_latched(ex):
                raise
            result["config"]["p2p_status"] = repr(ex)
            latched = True
        return result, latched

    def agree(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    return with_p2p_fallback(attempt, agree)


def tp_extra(args, torch, dist, rank, world, primary) -> dict | None:
    """BASELINE's second metric next to the data-parallel line of a multi-GPU run: Qwen3-32B with
    tensor_parallel_size = world, one cold pass — run as a SEPARATE job: every rank starts a child
    `bench.py --tp world --model qwen3-32b` and the children form their own process group on another port. The
    TP path has only ever run with all ranks on ONE GPU; a fault of its hipIpc P2P kernels on real links must not
    take the data-parallel measurement down with it, and a hang ends at the timeout (the children are killed).
    Whatever happens here, the primary line is printed."""
    import gc
    import subprocess
    gc.collect()
    torch.cuda.empty_cache()                       # the replica's engine has exited: hand the GPU to the child
    env = dict(os.environ)
    env["MASTER_PORT"] = str(1024 + (int(env.get("MASTER_PORT", "29500")) + 101 - 1024) % 60000)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # the children rendezvous among themselves (rank 0 child hosts the store)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--tp", str(world), "--model", "qwen3-32b",
           "--steps", "1", "--warmup", "0", "--num-seqs", str(args.num_seqs), "--no-roofline", "--no-cpu-baseline",
           "--no-tp-extra", "--gpu-memory-utilization", str(args.gpu_memory_utilization),
           "--num-kvcache-blocks", str(args.num_kvcache_blocks), "--kv-cache-dtype", args.kv_cache_dtype]
    if args.eager:
        cmd.append("--eager")
    timeout = float(os.environ.get("NVL_BENCH_TP_EXTRA_TIMEOUT", "900"))
    try:
        cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout:.0f} s"} if rank == 0 else None
    except Exception as ex:  # noqa: BLE001 — a secondary measurement must never sink the bench line
        return {"error": repr(ex)} if rank == 0 else None
    if rank != 0:
        return None
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    if cp.returncode != 0 or not lines:
        return {"error": f"child exit code {cp.returncode}", "stderr_tail": cp.stderr[-600:]}
    try:
        return json.loads(lines[-1])
    except ValueError as ex:
        return {"error": repr(ex)}


def compact_extra(line: dict) -> dict:
    """What the headline line keeps of an extra config's own bench line (the full line goes to the side file)."""
    roof = line.get("roofline") or {}
    out = {"value": line.get("value") and round(line["value"], 1), "ms": line.get("ms_per_step") and round(line["ms_per_step"], 1),
           "attn": roof.get("frac") and round(roof["frac"], 3),
           "step": roof.get("decode_step_frac_of_8TBps") and round(roof["decode_step_frac_of_8TBps"], 3),
           "pf_TF": (line.get("roofline_prefill") or {}).get("achieved") and round(line["roofline_prefill"]["achieved"]),
           "kv": (line.get("config") or {}).get("kv_blocks")}
    if roof.get("launches_with_shared_prefix_pass"):
        # the shared-prefix pass ran: attn / step are fractions in UNIQUE bytes (shared blocks counted once); attn_ps is the
        # same call priced in the reference's per-sequence bytes (can exceed 1: those bytes are not read from HBM any more)
        out["attn_ps"] = round(roof["rate_in_per_sequence_bytes_GBps"] / HBM_PEAK_GBPS, 3)
        out["attn_us"] = round(roof["avg_launch_us"], 1)
    return {k: v for k, v in out.items() if v is not None}


def extra_configs(args, torch) -> dict:
    """The other BASELINE.json configs next to the headline (config 2), each ONE cold pass of its workload in a child
    `bench.py` process (a fresh engine per model; a failure or time-out of an extra never sinks the headline line). The
    stdout line keeps a compact record per config (`compact_extra`; NOTES["extra_configs"] names the fields); the
    children's full lines are written to gpurun_out/bench_extras_full.json."""
    import gc
    import subprocess
    gc.collect()
    torch.cuda.empty_cache()                       # the headline engine has exited: hand the GPU to the children
    out, full = {}, {}
    common = ["--gpus", "1", "--steps", "1", "--no-cpu-baseline", "--no-extra-configs",
              "--gpu-memory-utilization", str(args.gpu_memory_utilization)]
    # (one untimed warm-up pass where a pass is short — a 1.8 s pass measured cold carries the library GEMM's first-call
    #  set-up of every new prefill shape; the 20-35 s passes of the 32B runs stay cold)
    # name -> (arguments, seconds the child took on the round's boxes incl. engine start). BASELINE's own configs first,
    # then the per-rank engines: an extra whose expected time does not fit into what is left of the wall budget is skipped.
    runs = {"config3": (["--model", "qwen3-8b", "--workload", "prefix", "--warmup", "1"], 13),
            # BASELINE config 4's single-GPU anchor: Qwen3-32B on the bench workload at TP = 1 — what a later
            # `--gpus N` line's tp_qwen3_32b divides by
            "config4_anchor": (["--model", "qwen3-32b", "--tp", "1", "--warmup", "0"], 48),
            "config5": (["--model", "qwen3-32b", "--tp", "1", "--workload", "long", "--max-num-seqs", "16", "--warmup", "0"], 30),
            # what ONE rank of the TP = 8 / TP = 4 engine computes, run as a TP = 1 engine: a rank's kernels without any
            # collective => an UPPER BOUND per rank, not a TP measurement (both quoted in BASELINE.md next to the anchor)
            "tp8_rank_shape_bench": (["--model", "qwen3-32b-tp8rank", "--tp", "1", "--warmup", "0"], 20),
            "tp4_rank_shape_bench": (["--model", "qwen3-32b-tp4rank", "--tp", "1", "--warmup", "0"], 23),
            "tp8_rank_shape_16k_prompts": (["--model", "qwen3-32b-tp8rank", "--tp", "1", "--workload", "long",
                                            "--max-num-seqs", "16", "--warmup", "1"], 44)}
    per_child = float(os.environ.get("NVL_BENCH_EXTRA_TIMEOUT", "300"))
    # the WHOLE run (engine start, warm-up and timed passes, roofline replay, CPU baseline, extras) aims at this wall time:
    # with the driver's 20 + 5 passes the headline part takes ~160 s, and the extras that fit are BASELINE's configs 3 / 4-anchor /
    # 5 (+ the TP = 8 rank shape on a fast box); the default 1 + 1 run has room for all six (round 6, 285 s tried: the run
    # took 271 s and the TP = 4 rank shape still did not fit; at 255 s it missed by 3 s on the fast boxes, hence 262 — BASELINE.md quotes it
    # from the default run's record when it is skipped)
    budget = float(os.environ.get("NVL_BENCH_WALL_BUDGET", "262"))
    env = dict(os.environ, OMP_NUM_THREADS="8")    # the children's host loops
    for name, (extra, expected_s) in runs.items():
        t0 = time.perf_counter()
        left = budget - (t0 - BENCH_T0)
        if left < expected_s:
            out[name] = {"error": f"skipped: needs ~{expected_s} s, {max(left, 0):.0f} s left of the {budget:.0f} s wall budget"}
            continue
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), *common, *extra], capture_output=True,
                                text=True, timeout=min(per_child, 2.5 * expected_s + 30), env=env)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode != 0 or not lines:
                out[name] = {"error": f"child exit code {cp.returncode}", "stderr_tail": cp.stderr[-300:]}
            else:
                full[name] = json.loads(lines[-1])
                out[name] = compact_extra(full[name])
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timed out after {min(per_child, 2.5 * expected_s + 30):.0f} s"}
        except Exception as ex:  # noqa: BLE001 — a secondary measurement must never sink the bench line
            out[name] = {"error": repr(ex)}
        out[name]["wall"] = round(time.perf_counter() - t0, 1)
    try:
        os.makedirs(SIDE_DIR, exist_ok=True)
        with open(os.path.join(SIDE_DIR, "bench_extras_full.json"), "w") as fh:
            json.dump(full, fh)
    except OSError:
        pass
    return out


def run_replica(args, torch, dist, rank, world, tp, backend):
    """One engine per launched process (TP = 1: N independent data-parallel replicas; or, launched as a single
    process with --tp T, one engine that spawns its own T - 1 workers)."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from nanovllm import LLM

    path = os.path.join(tempfile.gettempdir(), f"nvl_{args.model}_r{rank}")
    write_synthetic_checkpoint(path, args.model, with_weights=False, vocab_size=MODEL_VOCAB.get(args.model, 151936))
    llm = LLM(path, **engine_kwargs(args, tp))
    if rank == 0:       # which build of the library this run measures (NVL_LIBDIR selects another one for A/B runs)
        from nano_vllm_amd import ops as _ops
        print(f"[bench.py] library: {_ops.LIB_PATH}", file=sys.stderr, flush=True)

    # ---- record decode batches of a pass (for the roofline replay) --------------------------
    runner = llm.model_runner
    rec = {"on": False, "steps": 0, "ctx_tokens": 0, "samples": []}
    orig_prepare = runner.prepare_decode

    def prepare_spy(seqs, *a):
        n = orig_prepare(seqs, *a)
        if rec["on"]:
            st = runner.dstage.np
            rec["ctx_tokens"] += int(st["ctx"][:n].sum())
            if st["shp"][0] > 0:       # blocks a group of rows shares count once (the shared-prefix pass reads them so)
                rec["dedup_tokens"] = rec.get("dedup_tokens", 0) + int(st["shp"][0]) * runner.block_size * (int(st["shp"][1:1 + n].sum()) - 1)
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

    # ---- where the decode time goes by batch size: the GPU is never idle between decode steps (lookahead), so the
    #      interval between two consecutive `decode_end` returns is the later step's GPU time
    by_batch = {}                      # bucket -> [steps, seconds]
    last_end = [None]
    orig_end, orig_run_prefill = runner.decode_end, runner._run_prefill

    def decode_end_spy():
        n = runner._inflight[0][0] if runner._inflight else 0
        out = orig_end()
        now = time.perf_counter()
        if rec["on"] and last_end[0] is not None:
            b = next(c for c in (8, 16, 32, 64, 96, 128, 160, 192, 224, 256, 1 << 30) if n <= c)
            acc = by_batch.setdefault(b, [0, 0.0])
            acc[0] += 1
            acc[1] += now - last_end[0]
        last_end[0] = now
        return out

    def run_prefill_spy(seqs):
        out = orig_run_prefill(seqs)
        last_end[0] = None             # the next decode interval would include this prefill
        return out

    runner.decode_end = decode_end_spy
    runner._run_prefill = run_prefill_spy

    # ---- host-side time of the timed pass. With the decode lookahead (engine/core.py) schedule, postprocess
    #      and prepare_decode of step N+1 run while the GPU executes step N; only fill_tokens is serial. -----
    host = {"schedule_s": 0.0, "postprocess_s": 0.0, "prepare_decode_s": 0.0, "prefill_steps_s": 0.0}

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
    # a prefill step starts on a drained queue and ends with a stream sync: its host wall time is its GPU time
    runner._run_prefill = timed(runner._run_prefill, "prefill_steps_s")

    try:
        elapsed, prompts, out_lens = timed_passes(args, torch, dist, llm, world, backend, rec)
    except Exception as ex:  # noqa: BLE001 — a latched P2P collective surfaced inside generate(): re-run over the process group
        if not (_is_latched(ex) and tp > 1 and os.environ.get("NVL_TP_P2P", "1") != "0"):
            raise
        _exit_after_latch(llm)
        os.environ["NVL_TP_P2P"] = "0"                  # (the engine's workers inherit NVL_* when they are spawned)
        try:
            result, pending_cpu = run_replica(args, torch, dist, rank, world, tp, backend)
        finally:
            os.environ["NVL_TP_P2P"] = "1"
        result["tp_p2p_attempt"] = {"value_invalid": None, "ms_per_step": None, "p2p_status": repr(ex),
                                    "note": "a P2P collective latched a spin timeout inside generate() in this attempt; "
                                            "`value` is the re-run with NVL_TP_P2P=0"}
        result["config"]["parallelism"] += " [fallback after a latched P2P spin timeout]"
        return result, pending_cpu
    total_out = sum(out_lens)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    par = (f"dp{world} (1 engine replica per GPU, TP=1)" if tp == 1 else
           f"tp{tp} (one engine, {tp} ranks spawned by the engine; xGMI P2P all-reduce "
           f"{'on' if runner.p2p else 'OFF (process group: RCCL)'})")
    result = base_result(args, tp, world, world * (tp if world == 1 else 1), elapsed, total_out, llm, par)
    if args.num_seqs == 256 and args.model == "qwen3-0.6b" and args.workload == "bench" and tp == 1:
        result["vs_baseline"] = result["value"] / REF_4070_LAPTOP_TOKS

    result["config"]["decode_step_fusions"] = fusion_state(runner)
    if tp > 1:
        from nano_vllm_amd import tp as tp_mod
        result["config"]["p2p_handoff"] = tp_mod.handoff_report()
    if rank == 0 and not args.no_roofline and rec["samples"] and tp == 1:
        result["config"]["host_seconds_in_last_step"] = {k: round(v, 4) for k, v in host.items()}
        tot = max(sum(x[1] for x in by_batch.values()), 1e-9)
        bb = sorted(by_batch.items())
        # decode time by batch-size bucket (parallel arrays: bucket upper bound, steps, ms per step, share of decode time)
        result["config"]["decode_ms_per_step_by_batch"] = {
            "batch_le": [b if b < (1 << 30) else 1 << 30 for b, _ in bb], "steps": [v[0] for _, v in bb],
            "ms_per_step": [round(v[1] / v[0] * 1e3, 3) for _, v in bb], "share": [round(v[1] / tot, 3) for _, v in bb]}
        result["roofline"] = roofline_replay(torch, runner, rec, args.model if args.kv_cache_dtype == "bf16" else "no-pmc-pass")
        ds = decode_step_roofline(runner, rec, result, host["prefill_steps_s"])
        result["roofline"]["decode_step"] = ds
        # (flat copies: a record that keeps only one level of scalars still carries the whole-step figure)
        result["roofline"]["decode_step_frac_of_8TBps"] = ds["frac_of_8TBps"]
        result["roofline"]["decode_step_achieved_GBps"] = ds["achieved_GBps"]
        if rec.get("prefill"):
            result["roofline_prefill"] = prefill_replay(torch, runner, rec["prefill"])
    pending_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # a reported baseline: N = 1 runs only
        try:
            pending_cpu = cpu_baseline_prepare(torch, llm, args.model, prompts, out_lens, result)
        except Exception as ex:  # noqa: BLE001 — a reported baseline must never sink the bench line
            result["cpu_baseline"] = {"error": repr(ex)}
    try:
        llm.exit()
        if tp > 1:
            result["config"]["p2p_status"] = "ok" if runner.p2p else "n/a (process group)"
    except Exception as ex:  # noqa: BLE001 — a latched P2P spin timeout: the number is invalid, re-run over the process group
        if not (_is_latched(ex) and tp > 1 and os.environ.get("NVL_TP_P2P", "1") != "0"):
            raise
        first = result
        first["config"]["p2p_status"] = repr(ex)
        os.environ["NVL_TP_P2P"] = "0"                  # (the engine's workers inherit NVL_* when they are spawned)
        try:
            result, pending_cpu = run_replica(args, torch, dist, rank, world, tp, backend)
        finally:
            os.environ["NVL_TP_P2P"] = "1"
        result["tp_p2p_attempt"] = {"value_invalid": first["value"], "ms_per_step": first["ms_per_step"],
                                    "p2p_status": first["config"]["p2p_status"],
                                    "p2p_handoff": first["config"].get("p2p_handoff"),
                                    "note": "a P2P collective latched a spin timeout in this attempt; `value` is the "
                                            "re-run with NVL_TP_P2P=0"}
        result["config"]["parallelism"] += " [fallback after a latched P2P spin timeout]"
    return result, pending_cpu


def fusion_state(runner) -> dict:
    """Which seams of the reference's decode step are fused in THIS run, and the resulting kernel launches per step
    (the reference makes 13 launches per layer + lm_head + sampler). The count is for shapes the
    skinny decode GEMM covers (Qwen3-0.6B); deep-K models add a slab-reduce launch where a bf16 / SiLU GEMM splits K."""
    from nano_vllm_amd import layers
    geo = runner.geo
    fused_attn = layers._FUSED_DECODE
    per_layer = 8 if fused_attn else 9
    fixed = 2 + (1 if runner.use_plan else 0) + 1 + 3      # feed_tokens, embedding, [plan], final norm, head + 2 sampler
    return {"qknorm_rope_kvstore_in_attention": fused_attn, "silu_mul_in_gate_up_epilogue": True,
            "splitk_sum_in_add_rmsnorm_prologue": True, "per_step_attention_plan": bool(runner.use_plan),
            "decode_steps_with_shared_prefix_pass": int(runner.prefix_steps),
            "kernel_launches_per_decode_step": geo["layers"] * per_layer + fixed}


def decode_step_roofline(runner, rec, result, prefill_s: float) -> dict:
    """The whole decode STEP against the HBM roofline (not just its dominant kernel): algorithmic bytes of the
    pass's decode steps (every weight once per step + K/V of every context token) over the time of
    the last pass minus its prefill steps (host-timed: a prefill starts on a drained queue and ends in a sync)."""
    geo = runner.geo
    L, hkv = geo["layers"], geo["kv_heads"]
    # (K/V blocks that every row of a step shares — prefix-cache hits taken by the shared-prefix pass — count once)
    kv_bytes = (rec["ctx_tokens"] - rec.get("dedup_tokens", 0)) * 2 * hkv * 128 * runner.kv_cache.element_size() * L
    # streamed once per step: every layer + the lm_head matrix (the tied embedding table when tie_word_embeddings);
    # the embedding GATHER touches only one row per sequence
    params = sum(p.numel() for n, p in runner.model.named_parameters()
                 if not (n.startswith("lm_head") and geo["tie"]) and not ("embed_tokens" in n and not geo["tie"]))
    w_bytes = params * 2 * rec["steps"]
    sec = result["ms_per_step"] * 1e-3 - prefill_s
    gbps = (kv_bytes + w_bytes) / sec / 1e9
    return {"algorithmic_bytes": kv_bytes + w_bytes, "decode_seconds": sec, "prefill_seconds": prefill_s,
            "decode_steps": rec["steps"], "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS}


def roofline_replay(torch, runner, rec, model: str = "qwen3-0.6b") -> dict:
    """Time the decode-attention kernel alone on the recorded batches (HIP events on the launch
    stream, every layer's cache => cold K/V like in the real step). Same code path as
    tools/attn_replay.py, which is the command the rocprofv3 duration/PMC profiles are taken with."""
    from tools.attn_replay import replay
    geo = runner.geo
    hq, hkv, L = geo["heads"], geo["kv_heads"], geo["layers"]
    r = replay(torch, runner.kv_cache, rec["samples"], hq, hkv, runner.config.max_model_len, runner.decode_ws, fused=True,
               plan=runner.use_plan, shared_blocks_of=runner._prefix_group_worth_a_pass if runner.share_prefix else None)
    achieved = r["achieved_GBps"]
    # the kernel nvl_paged_attn_decode_fused dispatches to for this geometry (attn_decode.hip: decode_common)
    G, fp8 = hq // hkv, runner.kv_cache.element_size() == 1
    if G == 8 or (G in (2, 4) and os.environ.get("NVL_DECODE_MFMA", "1") != "0") or (G > 1 and G not in (2, 4)):
        kernel = f"decode_mfma8_kernel<fused, {'fp8' if fp8 else 'bf16'} KV, G={G}>"
    else:
        kernel = f"decode_stream_fp8_kernel<{G}, fused>" if fp8 else f"decode_stream_kernel<{G}, fused>"
    step_bytes = (rec["ctx_tokens"] - rec.get("dedup_tokens", 0)) * 2 * hkv * 128 * runner.kv_cache.element_size() * L
    if r["launches_with_shared_prefix_pass"]:
        # (one launch: the stream-K grid + the workgroups that serve the shared-prefix packs, attn_decode.hip)
        kernel = kernel.replace("decode_mfma8_kernel", "decode_mfma8_shared_kernel")
    extra = {}
    if r["launches_with_shared_prefix_pass"]:
        # the reference's attention reads the shared blocks once per sequence; the bytes credited here are the unique ones
        extra = {"per_sequence_bytes_per_launch": r["per_sequence_bytes_per_launch"],
                 "rate_in_per_sequence_bytes_GBps": r["per_sequence_bytes_per_launch"] / r["avg_launch_us"] / 1e3,
                 "launches_with_shared_prefix_pass": r["launches_with_shared_prefix_pass"]}
    return {**extra, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": pmc_traffic(r["algorithmic_bytes_per_launch"], model, kernel),
            "kernel": kernel + " + decode_stream_combine_kernel",
            "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "avg_launch_us": r["avg_launch_us"],
            "launches_timed": r["launches_timed"], "launched": r["launched"], "decode_steps_in_pass": rec["steps"],
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
    w64 = 0                 # batches the launcher hands to the generated one-wave-per-SIMD loop (attn_prefill.hip's rule)
    for cu in batches:
        n = int(cu[-1])
        lens = (cu[1:] - cu[:-1]).astype("int64")
        w64 += int(os.environ.get("NVL_PREFILL_W64", "1") != "0" and len(lens) <= 64 and lens.max() >= 2048 and n >= 1024 * len(lens))
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
            "traffic": None,
            "kernel": ("prefill_w64_kernel<false>, the generated asm loop" if w64 == len(batches) and w64 else
                       "prefill_attn_kernel<false>" if not w64 else "prefill_attn_kernel<false> / prefill_w64_kernel<false>") +
                      " (nvl_attn_prefill_varlen)",
            "avg_launch_us": ms * 1e3 / launches, "launches_timed": launches}


def pmc_traffic(alg_bytes_per_launch: float, model: str = "qwen3-0.6b", kernel: str = ""):
    """HBM bytes per launch from the committed rocprofv3 PMC pass of THIS kernel instantiation's launches
    (profiles/pmc_traffic.json, one entry per decode-attention kernel: FETCH_SIZE summed over the launches of
    tools/attn_replay.py, doubled as MI355X_MICROARCH.md §HBM prescribes for 16 B/lane streaming reads on
    gfx950, divided by the algorithmic bytes of the same launches; tools/pmc_traffic_update.py). PMC counters cannot
    be read from inside this process, so the ratio measured by that separate pass is applied to this run's bytes
    per launch — it is a property of the kernel's access pattern, not of the run. null when no PMC pass exists
    for the kernel, or when the pass was taken on another version of attn_decode.hip (the entry records the source's
    hash)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            db = json.load(fh)
        rec = next((v for k, v in db.get("kernels", {}).items() if kernel and kernel.startswith(k)), None)
        if rec is None or model == "no-pmc-pass":
            return None
        # the ratio belongs to the kernel SOURCE the PMC pass ran: a later edit of attn_decode.hip invalidates it (null
        # until `tools/gpu_round.sh <tag> pmcg` has been re-run on the new kernel)
        import hashlib
        with open(os.path.join(ROOT, "nano_vllm_amd", "csrc", "attn_decode.hip"), "rb") as fh:
            if hashlib.sha256(fh.read()).hexdigest()[:16] != rec.get("attn_decode_hip_sha16"):
                return None
        ratio = float(rec["hbm_read_bytes_over_algorithmic"])
    except (OSError, KeyError, ValueError):
        return None
    return ratio * alg_bytes_per_launch


def cpu_baseline_prepare(torch, llm, model_name, prompts, out_lens, result: dict):
    """The CPU oracle (a port of the reference's path: oracle/engine.py + oracle/model.py; /root/reference does not
    exist on the GPU box, so the imported reference itself cannot run here) on a bounded sample of the same seeded
    stream: the FIRST 8 sequences, outputs capped at 9 tokens each => one prefill step + 8 decode steps at B = 8 (~25-30 s
    of CPU work on 64 threads; SURVEY.md §8d's 16 sequences take 40 s for their prefill alone), timed separately (a decode-heavy figure like the workload's, not a prefill timing).

    The same leg yields the line's `parity`: the ENGINE first generates exactly that sample at the bench's own
    temperature 0.6 (a fraction of a second), and the oracle pass that is being timed is TEACHER-FORCED with those
    tokens — same forward passes, same cost — so every one of the 8 x 9 sampled tokens is judged against the oracle's
    race keys `l/0.6 - log E` with the draws replayed (oracle/judge.py; floor = the SURVEY constant 0.0195 x absmax
    instead of a second, eager-rounding oracle pass). The oracle is the checker here, never the thing measured or
    shipped; the judging arithmetic itself (Philox replay, top-2 of the keys) is outside the timed steps.

    Two phases: THIS function does what needs the engine (the sample's generation, the weights' host copies) and
    returns a closure with the CPU work; main() runs the closure after the engine has exited (the GPU is idle meanwhile;
    nothing else runs beside it) and it fills result["cpu_baseline"] / result["parity"]."""
    from nano_vllm_amd.weights import parameter_shapes, qwen3_config_dict, synth_tensor
    from nanovllm import SamplingParams
    from oracle.judge import SURVEY_FLOOR_REL, judge_run
    cfg = qwen3_config_dict(model_name, vocab_size=MODEL_VOCAB.get(model_name, 151936))
    dev = llm.model_runner.device
    weights = {n: synth_tensor(n, s, llm.config.seed, device=dev).cpu() for n, s in parameter_shapes(cfg).items()}
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    n_seq, cap, temp = 8, 9, 0.6
    sample_p = prompts[:n_seq]
    sample_o = [min(m, cap) for m in out_lens[:n_seq]]
    seed_ = llm.config.seed

    # ---- the product's run of the sample (recorded step by step like tests/test_e2e_gpu.py does)
    runner = llm.model_runner
    call, rec, open_steps = runner.call, [], []

    def snap(seqs, is_prefill):
        return dict(prefill=is_prefill, seq_ids=[q.seq_id for q in seqs], tables=[list(q.block_table) for q in seqs])

    def spy(method, *a):
        if method == "decode_begin":
            open_steps.append(snap(a[0], False))
        elif method == "run":
            open_steps.append(snap(*a))
        out = call(method, *a)
        if method in ("run", "decode_end"):
            step = open_steps.pop(0)
            step["tokens"] = list(out)
            rec.append(step)
        return out

    llm.reset_prefix_cache()                   # block ids restart at 0, as in the oracle's fresh pool
    base = llm._requests
    runner.call = spy
    try:
        llm.generate(sample_p, [SamplingParams(temperature=temp, ignore_eos=True, max_tokens=m) for m in sample_o],
                     use_tqdm=False)
    finally:
        runner.call = call

    def oracle_leg():
        try:
            torch.set_num_threads(threads)
            t = {True: 0.0, False: 0.0}
            steps = {True: 0, False: 0}

            def on_step(i, sec):               # the oracle, teacher-forced, timed per step
                t[rec[i]["prefill"]] += sec
                steps[rec[i]["prefill"]] += 1

            blocks = sum((len(p) + m + 255) // 256 for p, m in zip(sample_p, sample_o)) + 8
            v = judge_run(cfg, weights, sample_p, sample_o, rec, blocks, temperatures=[temp] * n_seq, seed=seed_,
                          floor_rel=SURVEY_FLOOR_REL, on_step=on_step, ordinal_base=base)
            dt = t[True] + t[False]
            n_prompt = sum(len(p) for p in sample_p)
            n_decode_tok = sum(sample_o) - n_seq
            result["cpu_baseline"] = {
                "value": sum(sample_o) / dt, "unit": "tok/s", "cores": threads, "kind": "port",
                "prefill": {"steps": steps[True], "prompt_tokens": n_prompt, "seconds": round(t[True], 2),
                            "tok_per_s": round(n_prompt / t[True], 1)},
                "decode": {"steps": steps[False], "tokens": n_decode_tok, "seconds": round(t[False], 2),
                           "tok_per_s": round(n_decode_tok / max(t[False], 1e-9), 2)},
                "sample": f"first {n_seq} seqs of the seeded bench stream ({n_prompt} prompt tokens), outputs capped at {cap} "
                          f"({sum(sample_o)} tokens: {steps[True]} prefill + {steps[False]} decode steps at B = {n_seq}), "
                          f"{dt:.1f} s; {cores} logical CPUs, torch threads = {threads}"}
            result["parity"] = {
                "rows": v.rows, "sampled_rows": v.sampled_rows, "exact": v.exact, "decisive": v.decisive,
                "decisive_exact": v.decisive_exact, "violations": len(v.violations),
                "worst_gap_in_floors": round(v.worst_gap_in_floors, 3), "floor_rel": v.floor_rel, "ok": v.ok(),
                "judged": "engine at T = 0.6 on the cpu_baseline sample vs the CPU oracle's race keys (NOTES.parity)"}
        except Exception as ex:  # noqa: BLE001 — a reported baseline must never sink the bench line
            result["cpu_baseline"] = {"error": repr(ex)}

    return oracle_leg


if __name__ == "__main__":
    main()
