"""Tensor parallelism on ONE MI355X (gpurun exposes a single GPU): every rank is a separate process on cuda:0
(NVL_TP_SHARE_GPU=1), the process group is gloo (RCCL refuses two ranks on one device), and the hand-written
xGMI collectives run over hipIpc mappings between the processes — same code as on an 8-GPU node, minus the
links. What this executes that nothing else does: the engine's worker spawn + control channel + staging-image
protocol (engine/core.py, engine/runner.py), vocab-parallel sampling, the P2P all-reduce kernels of
csrc/comm.hip (incl. inside captured hipGraphs), and the process-group fallback.
Not measured here: anything about link bandwidth or cross-device cache behaviour.
"""
import os
import socket
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ---------------------------------------------------------------------------------------------------------
# kernels: nvl_allreduce_run / _add_rmsnorm / _gather between W processes
def _comm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    # every rank on its own slice of the ONE GPU's compute units (read when the HSA runtime starts, i.e. before the first
    # device call of this fresh process): ranks whose spinning collective kernels share CUs starve each other
    # (profiles/r05_tp2_cu_mask_experiment.json) — with disjoint slices the sweep runs at the speed of its kernels
    per = 256 // world
    os.environ["HSA_CU_MASK"] = f"0:{rank * per}-{(rank + 1) * per - 1}"
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    try:
        from nano_vllm_amd import ops
        ops.load_library()

        def exchange(blob):
            out = [None] * world
            dist.all_gather_object(out, blob)
            return out

        hidden = 5120
        comm = ops.P2PComm(rank, world, 256 * hidden * 2, exchange, dist.barrier)
        dev = torch.device("cuda", 0)

        def part(r, rows, hid, salt):
            g = torch.Generator().manual_seed(1000 * salt + r)
            return torch.randn(rows, hid, generator=g).to(torch.bfloat16)

        errs = {}
        calls = 0
        shapes = [(1, 5120), (3, 1024), (37, 5120), (131, 5120), (256, 5120), (64, 4096), (1, 256), (131, 5120), (2, 5120)]
        if world > 2:                      # four processes time-slice ONE GPU here: keep the spin-heavy part short
            shapes = [(1, 5120), (2, 5120), (131, 5120), (64, 4096)]
        big = (131, 5120)
        if world > 4:                      # eight: a peer's kernel may wait behind other processes' queues for seconds
            shapes = [(1, 5120), (2, 5120), (16, 5120), (8, 4096)]     # (one-shot and two-shot; few workgroups each)
            big = (16, 5120)
        for it, (rows, hid) in enumerate(shapes):
            if hid % (8 * world):
                continue
            parts = [part(r, rows, hid, it) for r in range(world)]
            ref32 = sum(p.float() for p in parts)
            ref = ref32.to(torch.bfloat16)
            x = parts[rank].to(dev)
            got = comm.all_reduce(x.clone())
            # summation order is rank order in fp32, one rounding: reproduce exactly
            acc = torch.zeros(rows, hid)
            for p in parts:
                acc += p.float()
            exact = acc.to(torch.bfloat16)
            errs[f"ar_{rows}x{hid}"] = float((got.cpu().float() - exact.float()).abs().max())
            g = torch.Generator().manual_seed(77 + it)
            res = torch.randn(rows, hid, generator=g).to(torch.bfloat16)
            w = (1 + 0.1 * torch.randn(hid, generator=g)).to(torch.bfloat16)
            res_d = res.clone().to(dev)
            y = comm.all_reduce_add_rmsnorm(x.clone(), res_d, w.to(dev), 1e-6)
            s = exact.float() + res.float()
            yref = (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()).to(torch.bfloat16)
            sys.path.insert(0, ROOT)
            from oracle.ops import bf16_ulp_diff
            errs[f"res_{rows}x{hid}"] = float((res_d.cpu().float() - s.to(torch.bfloat16).float()).abs().max())
            errs[f"norm_ulp_{rows}x{hid}"] = int(bf16_ulp_diff(y.cpu(), yref).max())
            calls += 2
        # zero-copy input (the GEMM writes its partial sums straight into the shared region) and the lean hand-off
        # (store drains instead of system-scope fences: the region is uncached), all four combinations
        assert comm.handoff == "fenced"                # a fresh communicator is fenced until somebody validates lean
        for lean in (0, 1):
            comm.set_handoff("lean" if lean else "fenced")
            dist.barrier()
            for it, (rows, hid) in enumerate([big, (2, 5120)] + ([(64, 4096)] if world <= 4 else [])):
                if hid % (8 * world):
                    continue
                parts = [part(r, rows, hid, 50 + it) for r in range(world)]
                acc = torch.zeros(rows, hid)
                for p in parts:
                    acc += p.float()
                exact = acc.to(torch.bfloat16)
                buf = comm.input_buffer(rows, hid, dev)
                buf.copy_(parts[rank])
                got = comm.all_reduce(buf, out=torch.empty(rows, hid, dtype=torch.bfloat16, device=dev))
                errs[f"zero_copy_lean{lean}_{rows}x{hid}"] = float((got.cpu().float() - exact.float()).abs().max())
                x2 = torch.mm(parts[rank].to(dev), torch.eye(hid, dtype=torch.bfloat16, device=dev), out=buf)   # a GEMM as producer
                res = torch.zeros(rows, hid, dtype=torch.bfloat16, device=dev)
                w1 = torch.ones(hid, dtype=torch.bfloat16, device=dev)
                comm.all_reduce_add_rmsnorm(x2, res, w1, 1e-6)
                errs[f"zero_copy_norm_lean{lean}_{rows}x{hid}"] = float((res.cpu().float() - exact.float()).abs().max())
        comm.set_handoff("fenced")
        dist.barrier()
        # small all-gather (the sampler's winners): 512 rows x 8 bytes
        mine = torch.full((512, 2), rank + 1, dtype=torch.int32, device=dev)
        mine[:, 1] = torch.arange(512, dtype=torch.int32, device=dev) * (rank + 1)
        out = torch.zeros((world, 512, 2), dtype=torch.int32, device=dev)
        for _ in range(3):
            comm.all_gather(mine, out)
        torch.cuda.synchronize()
        ok = all(bool((out[r, :, 0] == r + 1).all()) and bool((out[r, :, 1].cpu() == torch.arange(512) * (r + 1)).all())
                 for r in range(world))
        errs["gather_ok"] = 0.0 if ok else 1.0
        # inside a captured hipGraph, replayed: epochs advance on the device
        x = part(rank, big[0], 5120, 999).to(dev)
        stat = x.clone()
        outb = torch.empty_like(x)
        comm.all_reduce(stat, out=outb)                       # warm-up
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            comm.all_reduce(stat, out=outb)
        acc = torch.zeros(big[0], 5120)
        for r in range(world):
            acc += part(r, big[0], 5120, 999).float()
        worst = 0.0
        for _ in range(5):
            outb.zero_()
            graph.replay()
            torch.cuda.synchronize()
            worst = max(worst, float((outb.cpu().float() - acc.to(torch.bfloat16).float()).abs().max()))
        errs["graph_replay"] = worst
        comm.status()
        dist.barrier()
        comm.close()
        q.put((rank, errs))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, pytest.param(4, marks=pytest.mark.slow)])     # (4 processes time-slice ONE GPU: ~60 s; the
def test_p2p_collectives_between_processes(world):                                   #  TP = 4 engine test drives the same kernels)
    """(W = 8 — kMaxWorld ranks, 64-element column slices — runs inside the TP = 8 engine test below: its
    tp.init_p2p stress self-check drives the one-shot, two-shot, fused-norm, gather and captured-graph forms of these
    kernels between eight processes. This standalone sweep with eight processes time-slicing the ONE GPU does not
    finish within minutes: a rank's kernel spins until the other seven get their turn.)"""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, errs in results.items():
        for name, err in errs.items():
            if name.startswith("norm_ulp"):
                assert err <= 1, f"rank {rank} {name}: {err} ulp"
            else:
                assert err == 0.0, f"rank {rank} {name}: {err}"


# ---------------------------------------------------------------------------------------------------------
# engine: tensor_parallel_size=2, judged against the CPU oracle exactly like the TP=1 engine
@pytest.fixture(scope="module")
def tiny_ckpt():
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    path = tempfile.mkdtemp(prefix="qwen3tiny_tp_")
    write_synthetic_checkpoint(path, "qwen3-tiny", seed=0, vocab_size=512, max_position_embeddings=2048)
    return path


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["p2p-graph", pytest.param("p2p-eager", marks=pytest.mark.slow), "group-eager"])
def test_tp2_engine_greedy_parity_on_one_gpu(tiny_ckpt, mode, monkeypatch):
    """(p2p modes: tp.init_p2p runs its full 1,000-epoch stress self-check of both hand-off flavours here.)"""
    from test_e2e_gpu import _check, _judge, _prompts, _run_ours
    monkeypatch.setenv("NVL_TP_SHARE_GPU", "1")
    monkeypatch.setenv("NVL_TP_BACKEND", "gloo")
    monkeypatch.setenv("NVL_TP_PORT", str(_free_port()))
    monkeypatch.setenv("NVL_TP_P2P", "0" if mode == "group-eager" else "1")
    prompts = _prompts(6, 5, 600, 512, seed=3)
    max_tokens = [24, 40, 8, 33, 1, 17]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, enforce_eager=mode != "p2p-graph", max_model_len=2048,
                                num_kvcache_blocks=32, max_num_seqs=16, tensor_parallel_size=2)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    _check(f"tiny TP=2 {mode}", _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, max_num_seqs=16), sum(max_tokens))


@pytest.mark.timeout(900)
def test_tp2_engine_shared_system_prompt_takes_the_shared_prefix_pass_on_every_rank(tiny_ckpt, monkeypatch):
    """The shared-prefix attention pass under tensor parallelism: rank 0 finds the group of rows that share their leading
    KV blocks and ships it inside the staging image; EVERY rank decides from that image, captures the bucket's graph with
    the pass at the same step (P2P collectives inside) and replays it — per rank 2 query heads on 1 kv head (packs of 8
    rows). Tokens, batches and block tables judged against the oracle engine; greedy and sampled rows in one batch."""
    from test_e2e_gpu import _check, _judge, _run_ours
    monkeypatch.setenv("NVL_TP_SHARE_GPU", "1")
    monkeypatch.setenv("NVL_TP_BACKEND", "gloo")
    monkeypatch.setenv("NVL_TP_PORT", str(_free_port()))
    monkeypatch.setenv("NVL_TP_P2P", "1")
    monkeypatch.setenv("NVL_TP_P2P_STRESS_EPOCHS", "100")
    monkeypatch.setenv("NVL_SHARED_PREFIX_MIN_MB", "0")
    g = torch.Generator().manual_seed(131)
    shared = torch.randint(0, 512, (520,), generator=g).tolist()
    prompts = [shared + torch.randint(0, 512, (int(n),), generator=g).tolist() for n in (3, 40, 150, 9, 220, 61)]
    prompts.insert(2, torch.randint(0, 512, (300,), generator=g).tolist())
    max_tokens = [30, 14, 5, 30, 21, 30, 9]
    temps = [0.0, 0.6, 0.0, 0.0, 0.0, 0.9, 0.0]
    info = {}
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, temperatures=temps, info=info, enforce_eager=False,
                                max_model_len=2048, num_kvcache_blocks=32, max_num_seqs=16, max_num_batched_tokens=1536,
                                tensor_parallel_size=2, seed=9)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    assert info["prefix_steps"] > 0 and len(info["prefix_graphs"]) >= 1, info
    _check("tiny TP=2 p2p-graph, shared system prompt",
           _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, temperatures=temps, seed=9, max_num_seqs=16,
                  max_num_batched_tokens=1536), sum(max_tokens))


@pytest.mark.timeout(900)
def test_tp2_engine_sampled_T06_parity_draws_replayed_on_one_gpu(tiny_ckpt, monkeypatch):
    """T > 0 at TP = 2 inside the captured decode graph: each rank races ITS vocabulary shard (Philox keyed by the GLOBAL
    column), 8 bytes per row are exchanged, every rank merges — the merged token must be the argmax of `l/T - log E`
    over the FULL row of the oracle's logits, draws replayed by oracle/philox.py (the reference gathers [B, V] logits
    to rank 0 and samples there, embed_head.py:62-65 + model_runner.py:212-218)."""
    from test_e2e_gpu import _check, _judge, _prompts, _run_ours
    monkeypatch.setenv("NVL_TP_SHARE_GPU", "1")
    monkeypatch.setenv("NVL_TP_BACKEND", "gloo")
    monkeypatch.setenv("NVL_TP_PORT", str(_free_port()))
    monkeypatch.setenv("NVL_TP_P2P", "1")
    monkeypatch.setenv("NVL_TP_P2P_STRESS_EPOCHS", "100")
    prompts = _prompts(6, 5, 600, 512, seed=3)
    max_tokens = [24, 40, 8, 33, 1, 17]
    temps = [0.6, 0.6, 1.0, 0.0, 0.6, 1.3]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, temperatures=temps, enforce_eager=False,
                                max_model_len=2048, num_kvcache_blocks=32, max_num_seqs=16, tensor_parallel_size=2,
                                seed=77)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    v = _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, temperatures=temps, seed=77, max_num_seqs=16)
    _check("tiny TP=2 p2p-graph T>0", v, sum(max_tokens))
    assert v.sampled_rows == sum(max_tokens) - 33


@pytest.mark.timeout(900)
@pytest.mark.parametrize("tp_size", [4, pytest.param(8, marks=pytest.mark.slow)])   # (eight ranks on ONE GPU: ~20 s; same code as 4)
def test_tp4_tp8_engine_greedy_parity_on_one_gpu(tp_size, monkeypatch):
    """The whole TP engine at degrees 4 and 8 (every rank a process on cuda:0): a 16 / 8-head model shards down to
    4 / 2 query heads and 2 / 1 kv heads per rank (models/qwen3.py:29-38 at TP = 8), the vocabulary into 8 shards whose
    sampling winners are merged on every rank, the W = 8 all-reduce kernels run inside the captured decode graph —
    judged against the oracle under the same margin rule as TP = 1."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from test_e2e_gpu import _check, _judge, _prompts, _run_ours
    path = tempfile.mkdtemp(prefix="qwen3tiny_kv8_")
    write_synthetic_checkpoint(path, "qwen3-tiny-kv8", seed=0, vocab_size=512, max_position_embeddings=2048)
    monkeypatch.setenv("NVL_TP_SHARE_GPU", "1")
    monkeypatch.setenv("NVL_TP_BACKEND", "gloo")
    monkeypatch.setenv("NVL_TP_PORT", str(_free_port()))
    monkeypatch.setenv("NVL_TP_P2P", "1")
    monkeypatch.setenv("NVL_TP_P2P_STRESS_EPOCHS", "60")      # 4 / 8 processes time-slice ONE GPU here
    prompts = _prompts(4, 5, 400, 512, seed=13)
    max_tokens = [10, 6, 12, 3]
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, enforce_eager=False, max_model_len=1024,
                                num_kvcache_blocks=16, max_num_seqs=8, tensor_parallel_size=tp_size)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    _check(f"tiny-kv8 TP={tp_size}", _judge(path, prompts, max_tokens, rec, nblk, max_num_seqs=8), sum(max_tokens))


# ---------------------------------------------------------------------------------------------------------
# RCCL: the "nccl" branches of tp.py, executed with a 1-rank group on the one GPU (RCCL refuses two ranks per device)
def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import torch.nn.functional as F
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    res = {}
    try:
        from nano_vllm_amd import ops, tp
        ops.load_library()
        tp.init(0, 1, issue_collectives=True)
        res["backend"] = tp._backend
        res["capturable"] = tp.capturable()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4096, 256, generator=g).to(torch.bfloat16).to(dev)
        w = torch.randn(512, 256, generator=g).to(torch.bfloat16).to(dev)
        # prefill-sized: GEMM chunks on the compute stream, RCCL all-reduce of chunk i on the side stream
        y = tp.linear_allreduce(x, w)
        torch.cuda.synchronize()
        res["overlap_err"] = float((y.float() - F.linear(x, w).float()).abs().max())
        res["side_stream_used"] = tp._side_stream is not None
        # decode-sized: inline all-reduce; host tensors go through a device copy; small all-gather into a tensor
        small = torch.randn(7, 256, generator=g).to(torch.bfloat16).to(dev)
        res["small_err"] = float((tp.all_reduce(small.clone()).float() - small.float()).abs().max())
        host = torch.tensor([5], dtype=torch.int64)
        tp.group_all_reduce(host, op=dist.ReduceOp.MIN)
        res["host_min"] = int(host.item())
        mine = torch.arange(64, dtype=torch.int32, device=dev).view(32, 2)
        allp = torch.zeros(1, 32, 2, dtype=torch.int32, device=dev)
        tp.all_gather_small(mine, allp)
        res["gather_ok"] = bool((allp[0] == mine).all())
        # an RCCL all-reduce captured into a hipGraph next to one of our kernels, replayed
        buf = torch.zeros(16, 256, dtype=torch.bfloat16, device=dev)
        wn = torch.ones(256, dtype=torch.bfloat16, device=dev)
        out = torch.empty_like(buf)
        tp.all_reduce(buf)                                    # warm-up: communicator + stream set-up outside the capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tp.all_reduce(buf)
            ops.rmsnorm(buf, wn, 1e-6, out=out)
        worst = 0.0
        for i in range(3):
            src = torch.randn(16, 256, generator=g).to(torch.bfloat16)
            buf.copy_(src)
            graph.replay()
            torch.cuda.synchronize()
            ref = src.float() * torch.rsqrt(src.float().pow(2).mean(-1, keepdim=True) + 1e-6)
            worst = max(worst, float((out.cpu().float() - ref).abs().max()))
        res["graph_err"] = worst
        tp.shutdown()
    except Exception as ex:  # noqa: BLE001 - reported to the parent
        res["error"] = repr(ex)
    finally:
        q.put(res)
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_code_paths_with_a_one_rank_group():
    """north star: "RCCL all-reduce over xGMI overlapped on a side HIP stream" (linear.py:153-156, model_runner.py:26).
    One GPU cannot host two RCCL ranks, so the RCCL branches of tp.py run here with a 1-rank "nccl" group in
    issue_collectives mode: RCCL initialisation, `linear_allreduce`'s chunked GEMM / side-stream all-reduce pipeline
    (events both ways), the inline all-reduce, the host-tensor and all-gather branches, and an RCCL all-reduce
    captured into a hipGraph together with one of our kernels (what `tp.capturable()` promises for this backend)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert "error" not in res, res
    assert res["backend"] == "nccl" and res["capturable"] and res["side_stream_used"]
    assert res["overlap_err"] == 0.0 and res["small_err"] == 0.0 and res["host_min"] == 5 and res["gather_ok"]
    assert res["graph_err"] <= 0.05


@pytest.mark.timeout(900)
def test_tp2_sampling_equals_tp1_draws(tiny_ckpt, monkeypatch):
    """T > 0: the vocab-parallel sampler keys Philox by the GLOBAL column, so TP=2 draws the same exponentials as
    TP=1; tokens agree wherever the (bf16-noisy) logits do — compared on the first sampled token of each
    sequence, which depends on one forward pass only."""
    from nano_vllm_amd import LLM, SamplingParams
    from test_e2e_gpu import _prompts
    prompts = _prompts(12, 5, 200, 512, seed=41)
    sp = SamplingParams(temperature=0.8, max_tokens=4, ignore_eos=True)

    def run(tp):
        if tp > 1:
            monkeypatch.setenv("NVL_TP_SHARE_GPU", "1")
            monkeypatch.setenv("NVL_TP_BACKEND", "gloo")
            monkeypatch.setenv("NVL_TP_PORT", str(_free_port()))
        llm = LLM(tiny_ckpt, enforce_eager=True, max_model_len=1024, num_kvcache_blocks=32, max_num_seqs=16, seed=5,
                  tensor_parallel_size=tp)
        outs = [o["token_ids"] for o in llm.generate(prompts, sp, use_tqdm=False)]
        llm.exit()
        return outs

    one, two = run(1), run(2)
    same_first = sum(a[0] == b[0] for a, b in zip(one, two))
    print(f"TP=2 vs TP=1 first sampled tokens equal: {same_first}/{len(prompts)}")
    assert same_first >= len(prompts) - 2


@pytest.mark.timeout(900)
def test_tp2_generate_raises_when_a_p2p_collective_has_latched_a_timeout(tiny_ckpt, monkeypatch):
    """A P2P collective whose peer never arrives gives up after a bounded spin, LATCHES the fact in the rank's flag region
    and carries on with an invalid sum (csrc/comm.hip `await`). The serving path must not hand the tokens sampled from
    such a step to the caller: the last node of every step ORs the ranks' latches into one word that travels to the host
    with the step's ids (nvl_allreduce_status_async), and step() / generate() raise. Injected here by writing the latch
    word of rank 0's flag region (offset of Flags::error in the 64 KiB region ahead of the data buffer)."""
    from nano_vllm_amd import LLM, SamplingParams, ops, tp
    monkeypatch.setenv("NVL_TP_SHARE_GPU", "1")
    monkeypatch.setenv("NVL_TP_BACKEND", "gloo")
    monkeypatch.setenv("NVL_TP_PORT", str(_free_port()))
    monkeypatch.setenv("NVL_TP_P2P", "1")
    monkeypatch.setenv("NVL_TP_P2P_STRESS_EPOCHS", "100")
    llm = LLM(tiny_ckpt, enforce_eager=False, max_model_len=2048, num_kvcache_blocks=32, max_num_seqs=16,
              tensor_parallel_size=2)
    try:
        assert llm.model_runner.p2p, "the P2P collectives are not in use: nothing to latch"
        sp = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)
        prompts = [[1, 2, 3, 4, 5], list(range(40, 90))]
        outs = llm.generate(prompts, sp, use_tqdm=False)
        assert [len(o["token_ids"]) for o in outs] == [6, 6]            # a healthy group: nothing raised
        flag_bytes, error_off = 64 * 1024, (2 * 256 * 8 + 256) * 4      # Flags{flag0, flag1, epoch, error} of comm.hip
        ptr = int(ops.lib().nvl_allreduce_buffer(tp.comm()._h)) - flag_bytes + error_off

        class _Word:
            __cuda_array_interface__ = {"shape": (1,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        word = torch.as_tensor(_Word(), device="cuda")
        assert int(word.item()) == 0
        word.fill_(1)
        torch.cuda.synchronize()
        with pytest.raises(ops.NvlError, match="gave up waiting for a peer"):
            llm.generate(prompts, sp, use_tqdm=False)
        word.fill_(0)                                                   # (so that exit()'s own check finds a clean group)
        torch.cuda.synchronize()
    finally:
        llm.exit()
