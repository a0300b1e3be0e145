"""Is an H2D copy from page-locked memory asynchronous for the host, and when do events behind it complete?
   python tools/debug/copy_async_probe.py"""
import time
import torch

dev = torch.device("cuda", 0)
n, H, W = 20, 3000, 4000
for dtype in (torch.float32, torch.uint16):
    host = [torch.zeros((H, W), dtype=dtype).pin_memory() for _ in range(n)]
    stage = torch.empty((n, H, W), dtype=dtype, device=dev)
    up = torch.cuda.Stream(dev)
    side = torch.cuda.Stream(dev)
    x = torch.zeros(1 << 20, device=dev)
    for mode in ("copy_", "to", "copy_+kernel"):
        for rep in range(2):
            torch.cuda.synchronize()
            ev = [torch.cuda.Event() for _ in range(n)]
            t0 = time.perf_counter()
            marks = []
            with torch.cuda.stream(up):
                for i in range(n):
                    if mode == "to":
                        d = host[i].to(dev, non_blocking=True)
                    else:
                        stage[i].copy_(host[i], non_blocking=True)
                    ev[i].record(up)
                    marks.append(time.perf_counter() - t0)
            t_enq = time.perf_counter() - t0
            done = []
            for i in range(n):
                ev[i].synchronize()
                done.append(time.perf_counter() - t0)
                if mode == "copy_+kernel":
                    with torch.cuda.stream(side):
                        x.add_(1.0)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
        print(f"{dtype} {mode}: enqueue of {n} copies {1e3 * t_enq:.2f} ms (first {1e3 * marks[0]:.2f}); event i done at (ms): "
              + " ".join(f"{1e3 * d:.1f}" for d in done) + f"; all {1e3 * t_all:.2f} ms")
