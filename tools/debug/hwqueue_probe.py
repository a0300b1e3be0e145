"""HIP streams share a few hardware queues (GPU_MAX_HW_QUEUES, default 4): does work on compute stream k start while
20 H2D copies are queued on the upload stream?  With a normal-priority and with a high-priority upload stream.
   python tools/debug/hwqueue_probe.py"""
import os, time
import torch

dev = torch.device("cuda", 0)
n, H, W = 20, 3000, 4000
host = [torch.zeros((H, W), dtype=torch.uint16).pin_memory() for _ in range(n)]
stage = torch.empty((n, H, W), dtype=torch.uint16, device=dev)
x = torch.zeros(1 << 22, device=dev)
comp = [torch.cuda.Stream(dev) for _ in range(8)]
for s in comp:  # first use: the hardware queue is assigned now
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for label, up in (("normal-priority upload stream", torch.cuda.Stream(dev)), ("HIGH-priority upload stream", torch.cuda.Stream(dev, priority=-1))):
    for rep in range(2):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event() for _ in range(n)]
        done = [torch.cuda.Event() for _ in comp]
        t0 = time.perf_counter()
        with torch.cuda.stream(up):
            for i in range(n):
                stage[i].copy_(host[i], non_blocking=True)
                ev[i].record(up)
        ev[0].synchronize()
        for s, d in zip(comp, done):
            with torch.cuda.stream(s):
                x.add_(1.0)
                d.record(s)
        t = []
        for d in done:
            d.synchronize()
            t.append(time.perf_counter() - t0)
        ev[n - 1].synchronize()
        t3 = time.perf_counter() - t0
    print(f"{label}: kernel on compute stream k done at (ms): " + " ".join(f"{1e3 * v:.2f}" for v in t) + f"; copies done {1e3 * t3:.2f}")
