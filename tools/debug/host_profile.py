"""Host-side profile of the eager step: where the ~35 us per launch go (cProfile over 10 bursts, top functions)."""
import cProfile, pstats, sys, io
sys.path.insert(0, "handheld-multi-frame-super-resolution_amd"); sys.path.insert(0, ".")
import numpy as np
import torch
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth

dev = torch.device("cuda")
H, W, NF, sc = 3000, 4000, 20, 2
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1)
cfg = hsr.default_config(); cfg.verbose = 0; cfg.scale = sc
hsr.prepare_config(cfg, np.full((H, W), 0.5, np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100, [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
for _ in range(3):
    hsr.main(ref, comp, cfg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    hsr.main(ref, comp, cfg)
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue()[:6000])
