"""Two bursts in flight: graph replays of two engines alternate on two streams — does the front end of burst i+1 hide
under the merge of burst i?   python tools/debug/pipelined.py [n_engines] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "handheld-multi-frame-super-resolution_amd"))
import numpy as np
import torch
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth, distributed as hdist

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
H, W, NF = 3000, 4000, 20
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
cfg = hsr.default_config()
cfg.verbose = 0
cfg.scale = 2
cfg.hip = {"graph": True}
hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                   [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
engs = [hdist.HipEngine(cfg) for _ in range(ne)]
ins = [(ref.clone(), comp.clone()) for _ in range(ne)]
streams = [torch.cuda.Stream(dev) for _ in range(ne)]
for _ in range(3):
    for e, (r, c), s in zip(engs, ins, streams):
        with torch.cuda.stream(s):
            e.single(r, c)
torch.cuda.synchronize()
print("graphs:", [len(e._runner.graphs) for e in engs])
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % ne
        with torch.cuda.stream(streams[k]):
            out = engs[k].single(*ins[k])
    torch.cuda.synchronize()
    print(f"{ne} engine(s): {1e3 * (time.perf_counter() - t0) / steps:.3f} ms per burst")
