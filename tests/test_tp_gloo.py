"""Tensor-parallel semantics on CPU: 2, 4 and 8 processes, gloo backend (the N>1 path; RCCL needs GPUs).

Each rank builds the sharded layers, loads its shard from the SAME full checkpoint tensors via
the layers' weight_loader (shard layout of layers/linear.py:54-156, embed_head.py:27-32), runs
forward, and the result must equal the unsharded computation: column-parallel outputs are the
matching slices, row-parallel outputs all-reduce to the full product, the vocab-parallel
embedding all-reduces to the full lookup and the LM head gathers full logits on rank 0.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from nano_vllm_amd import tp
    tp.init(rank, world)
    try:
        import torch.nn.functional as F
        from nano_vllm_amd import layers as L
        from nano_vllm_amd.attn_meta import reset_context, set_context
        g = torch.Generator().manual_seed(0)
        # 16 query / 8 kv heads (Qwen3's kv-head count): 8 / 4 / 2 query and 4 / 2 / 1 kv heads per rank at TP 2 / 4 / 8
        hidden, heads, kv, d, inter, vocab, n = 64, 16, 8, 16, 96, 56, 7
        x = torch.randn(n, hidden, generator=g)
        wq, wk, wv = (torch.randn(s, hidden, generator=g) for s in (heads * d, kv * d, kv * d))
        wo = torch.randn(hidden, heads * d, generator=g)
        wg, wu = torch.randn(inter, hidden, generator=g), torch.randn(inter, hidden, generator=g)
        wd = torch.randn(hidden, inter, generator=g)
        emb = torch.randn(vocab, hidden, generator=g)
        ids = torch.randint(0, vocab, (n,), generator=g)
        errs = {}

        qkv = L.QKVParallelLinear(hidden, d, heads, kv)
        for w, sid in ((wq, "q"), (wk, "k"), (wv, "v")):
            qkv.weight.weight_loader(qkv.weight, w, sid)
        y = qkv(x)
        hq, hk = heads // world, kv // world
        ref = torch.cat([F.linear(x, wq)[:, rank * hq * d:(rank + 1) * hq * d],
                         F.linear(x, wk)[:, rank * hk * d:(rank + 1) * hk * d],
                         F.linear(x, wv)[:, rank * hk * d:(rank + 1) * hk * d]], -1)
        errs["qkv"] = float((y - ref).abs().max())

        gu = L.MergedColumnParallelLinear(hidden, [inter, inter])
        gu.weight.weight_loader(gu.weight, wg, 0)
        gu.weight.weight_loader(gu.weight, wu, 1)
        i = inter // world
        ref = torch.cat([F.linear(x, wg)[:, rank * i:(rank + 1) * i], F.linear(x, wu)[:, rank * i:(rank + 1) * i]], -1)
        errs["gate_up"] = float((gu(x) - ref).abs().max())

        col = L.ColumnParallelLinear(hidden, inter)
        col.weight.weight_loader(col.weight, wg)
        errs["column"] = float((col(x) - F.linear(x, wg)[:, rank * i:(rank + 1) * i]).abs().max())

        row = L.RowParallelLinear(heads * d, hidden)
        row.weight.weight_loader(row.weight, wo)
        full_in = torch.randn(n, heads * d, generator=g)
        part = full_in[:, rank * hq * d:(rank + 1) * hq * d]
        errs["row_allreduce"] = float((row(part) - F.linear(full_in, wo)).abs().max())

        down = L.RowParallelLinear(inter, hidden)
        down.weight.weight_loader(down.weight, wd)
        act = torch.randn(n, inter, generator=g)
        errs["down_allreduce"] = float((down(act[:, rank * i:(rank + 1) * i]) - F.linear(act, wd)).abs().max())

        e = L.VocabParallelEmbedding(vocab, hidden)
        e.weight.weight_loader(e.weight, emb)
        errs["embedding"] = float((e(ids) - F.embedding(ids, emb)).abs().max())

        head = L.ParallelLMHead(vocab, hidden)
        head.weight.weight_loader(head.weight, emb)
        set_context(True, torch.tensor([0, 3, 7], dtype=torch.int32), torch.tensor([0, 3, 7], dtype=torch.int32), 4, 4)
        logits = head(x)
        reset_context()
        if rank == 0:
            errs["lm_head_gather"] = float((logits - F.linear(x[[2, 6]], emb)).abs().max())
        else:
            assert logits is None
        set_context(False)
        logits = head(x)
        reset_context()
        if rank == 0:
            errs["lm_head_decode"] = float((logits - F.linear(x, emb)).abs().max())
        q.put((rank, errs))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_layers_match_unsharded(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, errs in results.items():
        for name, err in errs.items():
            assert err < 1e-4, f"rank {rank} {name}: {err}"
    assert "lm_head_gather" in results[0]


def test_sequence_state_for_tp_workers_roundtrip():
    """Rank 0 -> worker control messages (pickle over shm): slim sequence state survives."""
    import pickle
    sys.path.insert(0, ROOT)
    from nano_vllm_amd.api import SamplingParams
    from nano_vllm_amd.engine.seq import Sequence
    s = Sequence(list(range(300)), SamplingParams(max_tokens=4))
    s.block_table, s.num_scheduled_tokens = [5, 9], 300
    msg = pickle.dumps(["run", [s], True])
    method, seqs, is_prefill = pickle.loads(msg)
    t = seqs[0]
    assert method == "run" and is_prefill and t.token_ids == list(range(300)) and t.block_table == [5, 9]
    assert t.num_scheduled_tokens == 300 and len(t) == 300
