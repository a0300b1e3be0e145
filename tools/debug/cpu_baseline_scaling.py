"""How does the all-cores CPU baseline (oracle.throughput_all_cores) scale with the number of concurrent crops?
   python tools/debug/cpu_baseline_scaling.py"""
import os, sys, time
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd")]
import oracle
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth

H, W, NF = 2048, 3072, 20
ref, comp, _ = synth.make_burst(H, W, NF, seed=1234)
cfg = hsr.default_config()
cfg.verbose = 0
cfg.scale = 2
hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                   [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
print("cores", os.cpu_count(), flush=True)
for c, ks in ((1024, (1, 2, 3, 6)), (512, (1, 6, 13))):
    origins = [(gy * c, gx * c) for gy in range(H // c) for gx in range(W // c)]
    for k in ks:
        crops = [(ref[y:y + c, x:x + c].copy(), comp[:, y:y + c, x:x + c].copy()) for y, x in origins[:k]]
        t0 = time.perf_counter()
        _, _, tc, used = oracle.throughput_all_cores(crops, cfg, cores=k * (NF - 1))
        print(f"crop {c}, {k} crops, {used} processes: {tc:.1f} s -> {k * (2 * c) ** 2 / tc / 1e6:.3f} Mpix/s", flush=True)
