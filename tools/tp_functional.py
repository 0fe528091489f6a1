"""FUNCTIONAL run of the tensor-parallel engine at real model widths on ONE GPU (not a measurement): every rank is
a process on cuda:0 (NVL_TP_SHARE_GPU=1), process group gloo, xGMI P2P collectives over hipIpc, decode steps in
captured hipGraphs. Prints one JSON line: tokens generated, whether the P2P path was in use, its status word.

    python tools/tp_functional.py [model] [tp] [num_prompts] [max_tokens]
"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NVL_TP_SHARE_GPU", "1")
os.environ.setdefault("NVL_TP_BACKEND", "gloo")
os.environ.setdefault("NVL_TP_PORT", "29517")
from random import randint, seed


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-32b"
    tp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    max_tokens = int(sys.argv[4]) if len(sys.argv) > 4 else 48
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from nanovllm import LLM, SamplingParams
    path = os.path.join(tempfile.gettempdir(), f"nvl_tpfunc_{model}")
    write_synthetic_checkpoint(path, model, with_weights=False)
    t0 = time.perf_counter()
    llm = LLM(path, max_model_len=2048, dummy_weights=True, tensor_parallel_size=tp, num_kvcache_blocks=64, max_num_seqs=16)
    t_init = time.perf_counter() - t0
    seed(0)
    prompts = [[randint(0, 10000) for _ in range(randint(50, 400))] for _ in range(n)]
    sp = SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=max_tokens)
    llm.generate(prompts[:2], SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=4), use_tqdm=False)
    t0 = time.perf_counter()
    outs = llm.generate(prompts, sp, use_tqdm=False)
    dt = time.perf_counter() - t0
    runner = llm.model_runner
    res = {"model": model, "tp": tp, "prompts": n, "tokens_out": sum(len(o["token_ids"]) for o in outs),
           "seconds": round(dt, 2), "init_seconds": round(t_init, 1), "p2p_collectives": bool(runner.p2p),
           "hipgraph": not runner.enforce_eager, "geo_per_rank": {k: runner.geo[k] for k in ("heads", "kv_heads", "inter", "vocab_per_rank")},
           "note": "all ranks share ONE GPU: functional evidence only (no links, no bandwidth)"}
    llm.exit()                # raises if any P2P collective ever timed out waiting for its peers
    res["p2p_status"] = "ok"
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
