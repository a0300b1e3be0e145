"""What the perimeter tiles cost the x3 merge (they run the generic per-pixel path: their reference-frame window leaves
the image): the fused merge of a 48 MP x 20 burst (or: frames H W scale) over ALL output rows, over the interior rows only (no top / bottom
tile row), and over the top / bottom 96-row bands alone.   python tools/debug/border_cost.py [frames]"""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd")]
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth
from handheld_super_resolution.merge import merge_burst

dev = torch.device("cuda", 0)
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 20
H, W, SC = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6000, 8000, 3)  # e.g. 20 3000 4000 2
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
cfg = hsr.default_config()
cfg.verbose = 0
cfg.scale = SC
hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                   [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
pipe = hsr.BurstPipeline(cfg).init_ref(ref)
frames = pipe.process_frames([comp[i] for i in range(NF - 1)], None, fuse_local_min=True)
sH, sW = SC * H, SC * W
out = torch.empty((sH, sW, 3), dtype=torch.float32, device=dev)


def timed(rows, n=3):
    r0, nr = rows
    view = out[r0:r0 + nr]
    fn = lambda: merge_burst(frames, pipe.ref, pipe.ref_covs, view, None, pipe.cfa, cfg, do_ref=True, divide=True,  # noqa: E731
                             rows=(r0, nr), out_height=sH, local_min=True)
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, rows in (("all rows", (0, sH)), ("interior rows (96 .. sH - 96)", (96, sH - 192)), ("top 96 rows", (0, 96)),
                   ("bottom 96 rows", (sH - 96, 96)), ("96 rows in the middle", (sH // 2 // 96 * 96, 96))):
    print(f"{name:34s} {timed(rows):8.3f} ms", flush=True)
