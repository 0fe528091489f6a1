"""Do parallel branches of a captured hipGraph overlap on MI355X?  (torch 2.10 / ROCm 7)"""
import os, sys, json, torch
dev = torch.device("cuda")
a = torch.empty(1 << 28, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
a2 = torch.empty(1 << 28, dtype=torch.uint8, device=dev); b2 = torch.empty_like(a2)
small = [torch.randn(64, 1024, device=dev) for _ in range(4)]

def big(x, y, n=8):
    for _ in range(n): y.copy_(x)
def chain(n=200):
    t = small[0]
    for i in range(n): t = t * 1.0001
    return t
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
side = torch.cuda.Stream()
res = {}
def both_serial(): big(a, b); chain()
def both_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side): chain()
    big(a, b)
    main.wait_stream(side)
def two_big_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side): big(a2, b2)
    big(a, b)
    main.wait_stream(side)
res["eager_big_ms"] = timeit(lambda: big(a, b))
res["eager_chain_ms"] = timeit(chain)
res["eager_serial_ms"] = timeit(both_serial)
res["eager_2streams_ms"] = timeit(both_streams)
def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return timeit(g.replay)
res["graph_big_ms"] = graphed(lambda: big(a, b))
res["graph_chain_ms"] = graphed(chain)
res["graph_serial_ms"] = graphed(both_serial)
res["graph_fork_ms"] = graphed(both_streams)
res["graph_two_big_fork_ms"] = graphed(two_big_streams)
res["env"] = {k: v for k, v in os.environ.items() if "GRAPH" in k or "HIP_" in k}
print(json.dumps(res, indent=1))
