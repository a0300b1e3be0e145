import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/handheld-multi-frame-super-resolution_amd")
import numpy as np, torch
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth
dev = torch.device("cuda", 0)
H, W, NF = 3000, 4000, 20
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
cfg = hsr.default_config(); cfg.verbose = 0; cfg.scale = 2
hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100, [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
for _ in range(3): hsr.main(ref, comp, cfg)
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    hsr.main(ref, comp, cfg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
