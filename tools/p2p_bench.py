"""Per-call time of the xGMI P2P collectives (csrc/comm.hip) between W processes that SHARE ONE GPU: the protocol
cost (flag round trips, local copies, launch) without any link — on a real node the link time of the two-shot
exchange adds to it (DESIGN.md §5 does that arithmetic). Each rank captures 64 back-to-back calls in a hipGraph.
Prints one JSON line from rank 0.   python tools/p2p_bench.py [world]"""
import json, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, q):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
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
        x = comm.input_buffer(rows, hid, torch.device("cuda", 0)) if zero_copy else torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
        x.copy_(torch.randn(rows, hid, device="cuda").to(torch.bfloat16))
        r = torch.randn(rows, hid, device="cuda").to(torch.bfloat16)
        w = torch.ones(hid, device="cuda", dtype=torch.bfloat16)
        y = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
        tag = ("lean" if lean else "fenced") + ("+zero_copy" if zero_copy else "")
        for name, fn in ((f"allreduce[{tag}]", lambda: comm.all_reduce(x, out=y)),
                         (f"allreduce_add_rmsnorm[{tag}]", lambda: comm.all_reduce_add_rmsnorm(x, r, w, 1e-6, out=y))):
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
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = q.get(timeout=500)
    for p in ps:
        p.join(60)
    print(json.dumps({"world": world, "note": "all ranks on ONE GPU: protocol cost only, no link time", "per_call": res}))
