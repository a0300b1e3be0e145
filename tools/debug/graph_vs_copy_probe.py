"""Does a HIP graph launched while H2D copies are in flight on another stream start before the copies finish?
   python tools/debug/graph_vs_copy_probe.py"""
import time
import torch

dev = torch.device("cuda", 0)
n, H, W = 20, 3000, 4000
host = [torch.zeros((H, W), dtype=torch.uint16).pin_memory() for _ in range(n)]
stage = torch.empty((n, H, W), dtype=torch.uint16, device=dev)
up, side, side2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
x = torch.zeros(1 << 24, device=dev)


separate = [torch.empty((H, W), dtype=torch.uint16, device=dev) for _ in range(n)]
y = torch.zeros((H, W), dtype=torch.float32, device=dev)
SRC = [stage[0]]


def work():
    for _ in range(10):
        x.add_(1.0)
    y.copy_(SRC[0])  # reads frame 0 of the staging memory while later frames are being written


with torch.cuda.stream(side):
    work()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    work()
# a graph with a fork / join over two streams, like the pipeline's
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=side):
    e = torch.cuda.Event()
    e.record(side)
    side2.wait_event(e)
    with torch.cuda.stream(side2):
        work()
    work()
    side.wait_stream(side2)
torch.cuda.synchronize()
for mode in ("eager kernels", "graph", "fork-join graph", "graph, launched after copy 0 only was enqueued",
             "SEPARATE allocations: eager kernels", "SEPARATE allocations: graph"):
    if mode.startswith("SEPARATE") and SRC[0] is stage[0]:
        SRC[0] = separate[0]
        dst = separate
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            work()
    elif not mode.startswith("SEPARATE"):
        dst = [stage[i] for i in range(n)]
    for rep in range(2):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event() for _ in range(n)]
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        with torch.cuda.stream(up):
            for i in range(n if not mode.endswith("enqueued") else 1):
                dst[i].copy_(host[i], non_blocking=True)
                ev[i].record(up)
        ev[0].synchronize()
        t1 = time.perf_counter() - t0
        with torch.cuda.stream(side):
            if mode == "eager kernels":
                work()
            elif mode == "fork-join graph":
                g2.replay()
            else:
                g.replay()
            done.record(side)
        if mode.endswith("enqueued"):
            with torch.cuda.stream(up):
                for i in range(1, n):
                    dst[i].copy_(host[i], non_blocking=True)
                    ev[i].record(up)
        done.synchronize()
        t2 = time.perf_counter() - t0
        ev[n - 1].synchronize()
        t3 = time.perf_counter() - t0
    print(f"{mode}: copy 0 done {1e3 * t1:.2f} ms, side-stream work done {1e3 * t2:.2f} ms, all copies done {1e3 * t3:.2f} ms")
