import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle
from handheld_super_resolution import merge
from helpers import base_config
import test_hip_parity as t
scale = 3
H, W, ts = 48, 64, 16
c64 = base_config(ts=ts, scale=scale); c64.hip = {"weight_fp64": True}
ref, fr = t._frames(H, W, 2, ts, 9, c64)
thr = sorted({(2 * scale - 2 * rem - 1) / (2 * scale) for rem in range(scale)} | {0.0, 0.5})
vals = []
for k in (-3.0, -1.0, 0.0, 2.0):
    for tt in thr:
        f = np.float32(k + tt)
        vals += [f, np.nextafter(f, np.float32(-10)), np.nextafter(f, np.float32(10))]
vals = np.array(vals, np.float32)
ny, nx = fr[0][1].shape[:2]
rng = np.random.default_rng(1)
for k in range(2):
    flow = vals[rng.integers(0, len(vals), (ny, nx, 2))]
    fr[k] = (fr[k][0], flow.astype(np.float32), fr[k][2], fr[k][3])
tf = [tuple(t.T(a) for a in f) for f in fr]
sH, sW = scale * H, scale * W
for k, f in enumerate(tf):
    num, den = torch.zeros(sH, sW, 3, device="cuda"), torch.zeros(sH, sW, 3, device="cuda")
    n64, d64 = torch.zeros_like(num), torch.zeros_like(den)
    merge.merge(*f, num, den, [[0, 1], [1, 2]], base_config(ts=ts, scale=scale))
    merge.merge(*f, n64, d64, [[0, 1], [1, 2]], c64)
    rel = ((num - n64).abs() / (n64.abs() + 1e-12))
    idx = ((rel > 2e-5) & ((num - n64).abs() > 1e-6)).nonzero()
    print("frame", k, "bad", idx.shape[0])
    for i in idx[:5].cpu().numpy():
        hi, hj, c = i
        print(" px", hi, hj, c, float(num[hi, hj, c]), float(n64[hi, hj, c]), "den", float(den[hi,hj,c]), float(d64[hi,hj,c]), "flow tile", fr[k][1][(hi // scale) // ts, (hj // scale) // ts], "r", fr[k][3][hi // scale, hj // scale])
        o = oracle
        onum, oden = np.zeros((sH, sW, 3), np.float32), np.zeros((sH, sW, 3), np.float32)
        oracle.merge(*fr[k], onum, oden, [[0, 1], [1, 2]], base_config(ts=ts, scale=scale))
        print("  oracle", onum[hi, hj, c], oden[hi, hj, c])
