"""Do H2D copies slow down while a heavy kernel stream is busy?  Normal- vs high-priority upload stream.
   python tools/debug/copy_contention_probe.py"""
import time
import torch

dev = torch.device("cuda", 0)
n, H, W = 20, 3000, 4000
host = [torch.zeros((H, W), dtype=torch.uint16).pin_memory() for _ in range(n)]
stage = torch.empty((n, H, W), dtype=torch.uint16, device=dev)
a = torch.randn(8192, 8192, device=dev)
big = torch.zeros(1 << 28, device=dev)  # 1 GiB: HBM-bound elementwise
comp = torch.cuda.Stream(dev)
with torch.cuda.stream(comp):
    big.add_(1.0)
torch.cuda.synchronize()
for label, up in (("normal", torch.cuda.Stream(dev)), ("HIGH", torch.cuda.Stream(dev, priority=-1))):
    for load in ("idle", "HBM-bound kernels", "GEMMs"):
        for rep in range(2):
            torch.cuda.synchronize()
            ev = [torch.cuda.Event() for _ in range(n)]
            t0 = time.perf_counter()
            with torch.cuda.stream(comp):
                if load == "HBM-bound kernels":
                    for _ in range(40):
                        big.add_(1.0)
                elif load == "GEMMs":
                    for _ in range(12):
                        a @ a
            with torch.cuda.stream(up):
                for i in range(n):
                    stage[i].copy_(host[i], non_blocking=True)
                    ev[i].record(up)
            ev[n - 1].synchronize()
            t1 = time.perf_counter() - t0
            torch.cuda.synchronize()
            t2 = time.perf_counter() - t0
        print(f"{label} upload stream, GPU {load}: copies done {1e3 * t1:.2f} ms ({n * H * W * 2 / t1 / 1e9:.1f} GB/s), kernels done {1e3 * t2:.2f} ms")
