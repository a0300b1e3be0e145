"""A/B of the FFT low-pass kernels on one GPU: time hhsr_grey_lowpass_batch (4 frames per launch, like the frame pipeline's
chunks) per value of an environment switch of the library read at plan creation (default HHSR_FFT_STATIC: bit 0 the row
kernels, bit 1 the column kernel run the compile-time plan's passes; 0 = the run-time passes).  Three alternating rounds.
usage: python tools/fft_ab.py [H W] [values...]      (FFT_AB_ENV=NAME picks another switch)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from handheld_super_resolution import utils_image  # noqa: E402

args = [a for a in sys.argv[1:]]
H, W = (int(args[0]), int(args[1])) if len(args) >= 2 else (3000, 4000)
masks = [int(a) for a in args[2:]] or [0, 1, 2, 3]
ENV = os.environ.get("FFT_AB_ENV", "HHSR_FFT_STATIC")
NF, REP = 4, 30
dev = "cuda"
imgs = [torch.rand((H, W), device=dev) for _ in range(NF)]
res = {}
for rnd in range(3):
    for m in masks:
        os.environ[ENV] = str(m)
        utils_image._grey_plans.clear()
        for _ in range(3):
            utils_image.compute_grey_images_batch(imgs, "FFT")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            utils_image.compute_grey_images_batch(imgs, "FFT")
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(m, []).append(e0.elapsed_time(e1) / REP / NF * 1e3)
for m in masks:
    print(f"{ENV}={m}: us per frame ({H}x{W}, {NF} frames per launch): " + " ".join(f"{v:.1f}" for v in res[m]))
