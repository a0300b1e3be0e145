"""`mode: grey` at 12 MP x 20 x2: time per burst, and the fused merge with the x2 tile kernel vs the generic kernel.
   python tools/debug/mono_timing.py"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "handheld-multi-frame-super-resolution_amd"))
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth
from handheld_super_resolution.merge import merge_burst

dev = torch.device("cuda", 0)
H, W, NF = 3000, 4000, 20
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234, cfa=((1, 1), (1, 1)))


def cfg_fn(kern):
    cfg = hsr.default_config()
    cfg.verbose = 0
    cfg.scale = 2
    cfg.mode = "grey"
    cfg.hip = {"merge_kernel": kern}
    hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                       [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
    return cfg


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for kern in ("generic", "auto"):
    cfg = cfg_fn(kern)
    t = timed(lambda: hsr.main(ref, comp, cfg))
    pipe = hsr.BurstPipeline(cfg).init_ref(ref)
    fr = pipe.process_frames([comp[i] for i in range(NF - 1)], None)
    num = torch.empty((2 * H, 2 * W, 3), dtype=torch.float32, device=dev)
    tm = timed(lambda: merge_burst(fr, pipe.ref, pipe.ref_covs, num, None, pipe.cfa, cfg))
    from handheld_super_resolution import distributed as hdist

    eng = hdist.HipEngine(cfg)
    tg = timed(lambda: eng.single(ref, comp), 8)
    print(f"merge kernel {kern}: main() {t:.2f} ms per burst (eager), {tg:.2f} ms replayed from a HIP graph, fused merge alone {tm:.2f} ms")
