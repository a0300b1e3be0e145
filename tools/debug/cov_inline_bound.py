"""Covariance-on-the-fly (second half): measured bound instead of an instruction-count estimate.
What fusing the kernel estimation into the merge could SAVE is (a) the covariance half of the per-frame raw pass and
(b) the HBM traffic of the 19 covariance planes the merge reads; what it would COST is the structure-tensor +
eigen-decomposition arithmetic inside a kernel that is already VALU-bound.  This script measures (a) and (b):
  (a) hhsr_frame_stats with and without the covariance half (same launch geometry, statistics only);
  (b) the fused merge with every frame reading ITS OWN covariance plane vs all frames reading ONE plane (48 MB: stays in
      the 256 MB Infinity Cache) — the second is the merge with the covariance traffic removed but its arithmetic intact,
      i.e. the best case of an inline computation that costs nothing.
   python tools/debug/cov_inline_bound.py"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "handheld-multi-frame-super-resolution_amd"))
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth, kernels, robustness
from handheld_super_resolution.merge import merge_burst

dev = torch.device("cuda", 0)
H, W, NF = 3000, 4000, 20
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
cfg = hsr.default_config()
cfg.verbose = 0
cfg.scale = 2
hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                   [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
cfa, wb = [[0, 1], [1, 2]], [1.0, 1.0, 1.0]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


frames = [comp[i] for i in range(NF - 1)]
t_both = timed(lambda: kernels.frame_stats_batch(frames[:4], cfa, wb, cfg)) / 4
t_stats = timed(lambda: [robustness.compute_local_stats_from_raw(f, cfa, wb, want_vars=False) for f in frames[:4]]) / 4
t_cov = timed(lambda: [kernels.estimate_kernels(f, cfg) for f in frames[:4]]) / 4
print(f"(a) per 12 MP frame: statistics + covariances {1e3 * t_both:.1f} us, statistics only {1e3 * t_stats:.1f} us, "
      f"covariances only {1e3 * t_cov:.1f} us -> the covariance half costs {1e3 * (t_both - t_stats):.1f} us per frame, "
      f"{19 * (t_both - t_stats):.2f} ms per 20-frame burst")
pipe = hsr.BurstPipeline(cfg).init_ref(ref)
fr = pipe.process_frames(frames, None, fuse_local_min=True)
num = torch.empty((2 * H, 2 * W, 3), dtype=torch.float32, device=dev)
t_own = timed(lambda: merge_burst(fr, pipe.ref, pipe.ref_covs, num, None, pipe.cfa, cfg, local_min=True), 5)
shared = [(f[0], f[1], fr[0][2], f[3]) for f in fr]
t_shared = timed(lambda: merge_burst(shared, pipe.ref, pipe.ref_covs, num, None, pipe.cfa, cfg, local_min=True), 5)
print(f"(b) fused merge, 19 frames + reference: own covariance planes {t_own:.3f} ms, ONE shared plane {t_shared:.3f} ms -> "
      f"removing the covariance traffic (19 x 48 MB) saves {t_own - t_shared:.3f} ms")
print(f"upper bound of the gain of a free inline computation: {19 * (t_both - t_stats) + (t_own - t_shared):.2f} ms per burst")
