import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth, distributed as hdist
from helpers import base_config
H, W = int(sys.argv[1]), int(sys.argv[2]); scale = float(sys.argv[3]); scale = int(scale) if scale.is_integer() else scale
ref, comp, shifts = synth.make_burst_torch(H, W, 3, "cuda", seed=5)
cfg = base_config(ts=16, scale=scale, metrics=("L1", "L2", "L2", "L2"))
out, _ = hsr.main(ref, comp, cfg)
eng = hdist.HipEngine(cfg).init_ref(ref)
flows = eng.align_frames([comp[0], comp[1]])
sH = out.shape[0]
r0 = (sH // 2 // 96) * 96; r1 = min(sH, r0 + 96 * 3)
print("rows", r0, r1, "sub", hdist.sub_image_rows(r0, r1, scale, H, 16, float(flows[..., 1].abs().max())), "maxflow", float(flows[..., 1].abs().max()))
slab, _ = eng.merge_rows([comp[0], comp[1]], flows, r0, r1, float(flows[..., 1].abs().max()))
a, b = slab, out[r0:r1]
d = (a - b).abs()
bad = ~((d <= 1e-5 * b.abs() + 1e-7) | (a.isnan() & b.isnan()))
print("bad", int(bad.sum()), "of", bad.numel(), "max", float(torch.nan_to_num(d).max()))
if bad.any():
    rows = bad.any(-1).any(-1).nonzero().flatten().cpu().numpy()
    cols = bad.any(-1).any(0).nonzero().flatten().cpu().numpy()
    print("rows", rows[:20], rows[-5:], "cols", cols[:20], cols[-5:], len(rows), len(cols))
