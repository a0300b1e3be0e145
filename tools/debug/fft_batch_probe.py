"""FFT low-pass: us per frame as a function of the frames per launch (the kept spectra of a launch's frames live between the
three kernels: 24 MB per 12 MP frame, 96 MB per 48 MP frame — against the 256 MB of Infinity Cache).
usage: python tools/debug/fft_batch_probe.py H W [frames-per-launch ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from handheld_super_resolution import utils_image  # noqa: E402

H, W = int(sys.argv[1]), int(sys.argv[2])
nfs = [int(a) for a in sys.argv[3:]] or [1, 2, 4]
TOTAL, REP = 4, 20
imgs = [torch.rand((H, W), device="cuda") for _ in range(TOTAL)]
res = {}
for rnd in range(3):
    for nf in nfs:
        def run():
            for i in range(0, TOTAL, nf):
                utils_image.compute_grey_images_batch(imgs[i:i + nf], "FFT")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(nf, []).append(e0.elapsed_time(e1) / REP / TOTAL * 1e3)
for nf in nfs:
    print(f"{H}x{W}, {nf} frame(s) per launch: us per frame " + " ".join(f"{v:.1f}" for v in res[nf]))
