"""Per-call time of the xGMI P2P collectives (csrc/comm.hip) between W processes. Default: all of them SHARE ONE GPU — the
protocol cost (flag round trips, local copies, launch) without any link; on a real node the link time of the two-shot
exchange adds to it (DESIGN.md §5 does that arithmetic). `--multi-gpu`: rank r on GPU r (a real node: first contact),
process group nccl (= RCCL), and every shape is also timed through `dist.all_reduce` on the same stream — the number the
P2P kernels have to beat. Each rank captures 64 back-to-back calls in a hipGraph.
Prints one JSON line from rank 0.   python tools/p2p_bench.py [world] [--multi-gpu]"""
import json, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


MULTI = "--multi-gpu" in sys.argv


def worker(rank, world, port, q, multi=False):
    dev_index = rank if multi else 0
    torch.cuda.set_device(dev_index)
    dist.init_process_group("nccl" if multi else "gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from nano_vllm_amd import ops
    ops.load_library()

    def exchange(blob):
        out = [None] * world
        dist.all_gather_object(out, blob)
        return out

    comm = ops.P2PComm(rank, world, 256 * 5120 * 2, exchange, dist.barrier)
    res = {}
    for rows, hid, lean, zero_copy in [(r_, h_, l_, z_) for (r_, h_) in ((1, 5120), (16, 5120), (131, 5120), (256, 5120), (131, 4096))
                                       for l_ in (0, 1) for z_ in (0, 1)]:
        if hid % (8 * world):
            continue
        ops.lib().nvl_allreduce_set_fences(comm._h, 0 if lean else 1)
        dist.barrier()
        x = comm.input_buffer(rows, hid, torch.device("cuda", dev_index)) if zero_copy else torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
        x.copy_(torch.randn(rows, hid, device="cuda").to(torch.bfloat16))
        r = torch.randn(rows, hid, device="cuda").to(torch.bfloat16)
        w = torch.ones(hid, device="cuda", dtype=torch.bfloat16)
        y = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
        tag = ("lean" if lean else "fenced") + ("+zero_copy" if zero_copy else "")
        cases = [(f"allreduce[{tag}]", lambda: comm.all_reduce(x, out=y)),
                 (f"allreduce_add_rmsnorm[{tag}]", lambda: comm.all_reduce_add_rmsnorm(x, r, w, 1e-6, out=y))]
        if multi and not lean and not zero_copy:     # the process group's own all-reduce on the same stream (RCCL), once per shape
            xr = torch.randn(rows, hid, device="cuda").to(torch.bfloat16)
            cases.append(("rccl_all_reduce", lambda: dist.all_reduce(xr)))
        for name, fn in cases:
            fn(); torch.cuda.synchronize(); dist.barrier()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(64):
                    fn()
            dist.barrier()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay(); torch.cuda.synchronize(); dist.barrier()
            s.record()
            for _ in range(5):
                g.replay()
            e.record(); torch.cuda.synchronize()
            res[f"{name}_{rows}x{hid}_us"] = round(s.elapsed_time(e) * 1e3 / (5 * 64), 2)
            dist.barrier()
    comm.status()
    dist.barrier()
    comm.close()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    world = int(args[0]) if args else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q, MULTI)) for r in range(world)]
    for p in ps:
        p.start()
    res = q.get(timeout=500)
    for p in ps:
        p.join(60)
    note = "one rank per GPU: links included; rccl_all_reduce_* = the process group on the same shapes" if MULTI else \
        "all ranks on ONE GPU: protocol cost only, no link time"
    print(json.dumps({"world": world, "note": note, "per_call": res}))
