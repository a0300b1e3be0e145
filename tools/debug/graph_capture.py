"""Experiment: capture main() in a HIP graph (torch.cuda.graph) and replay it — results and step time vs eager."""
import sys, time
sys.path.insert(0, "handheld-multi-frame-super-resolution_amd"); sys.path.insert(0, ".")
import numpy as np
import torch
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth

dev = torch.device("cuda")
H, W, NF, sc = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (3000, 4000, 20, 2))]
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1)
cfg = hsr.default_config(); cfg.verbose = 0; cfg.scale = sc
hsr.prepare_config(cfg, np.full((H, W), 0.5, np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100, [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        eager = hsr.main(ref, comp, cfg)[0]
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
eager = eager.clone()
t0 = time.perf_counter()
for _ in range(10):
    out = hsr.main(ref, comp, cfg)[0]
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 10 * 1e3)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    gout = hsr.main(ref, comp, cfg)[0]
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
same = torch.equal(torch.nan_to_num(gout), torch.nan_to_num(eager))
print("graph replay == eager:", same, float((torch.nan_to_num(gout) - torch.nan_to_num(eager)).abs().max()))
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3)
# new input content in the same buffers -> replay computes the new burst
ref2, comp2, _ = synth.make_burst_torch(H, W, NF, dev, seed=2)
want2 = hsr.main(ref2, comp2, cfg)[0].clone()
ref.copy_(ref2); comp.copy_(comp2)
g.replay(); torch.cuda.synchronize()
print("replay on new content == eager:", torch.equal(torch.nan_to_num(gout), torch.nan_to_num(want2)))
