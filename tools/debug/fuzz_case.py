"""Dissect ONE case of tests/test_fuzz_parity.py with the oracle's flows injected: where the values above 1e-4 are, the
flows, the robustness of every frame (oracle / HIP) and the accumulated weights there.
   python tools/debug/fuzz_case.py <generator seed> <case index>"""
import sys
import numpy as np
sys.path.insert(0, "handheld-multi-frame-super-resolution_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_fuzz_parity as fz
import handheld_super_resolution as hsr

gs, k = int(sys.argv[1]), int(sys.argv[2])
c = fz.cases(gs, k + 1)[k]
print("case", c)
import oracle
ref, comp = fz.burst(c)
ref, comp, o_own, gflow, hr_own, acc_own = fz.hip_own(c, ref, comp)
cap, cap_h, cap_m = {}, {}, {}
want, _ = oracle.main(ref, comp, fz.config(c), capture=cap, fast=True)
want_h, _ = oracle.main(ref, comp, fz.config(c), capture=cap_h, fast=True, flows=list(gflow), reuse=cap)
oflow, o_r, den_o = np.stack(cap["flow"]), (np.stack(cap["r"]) if c["rob"] else None), cap["den"]
dh = np.where(np.isnan(want_h), 0.0, np.abs(o_own.astype(np.float64) - want_h))
print(f"side H (HIP's flows): max |o - want_h| = {dh.max():.3e} ({int((dh > 1e-4).sum())} > 1e-4); oracle's own move under HIP's "
      f"flows max |want_h - want| = {np.nanmax(np.abs(want_h.astype(np.float64) - want)):.3e}; max |flow diff| = {np.abs(gflow - oflow).max():.3e} px")
if c["rob"]:
    want_m, _ = oracle.main(ref, comp, fz.config(c), capture=cap_m, fast=True, flows=list(gflow), rob=list(hr_own), reuse=cap)
    dm = np.where(np.isnan(want_m), 0.0, np.abs(o_own.astype(np.float64) - want_m))
    print(f"merge alone (HIP's flows and HIP's robustness): max |o - want_hm| = {dm.max():.3e} ({int((dm > 1e-4).sum())} > 1e-4); "
          f"max |r_hip - r_oracle(HIP's flows)| = {np.abs(hr_own - np.stack(cap_h['r'])).max():.3e}")
    for (y, x, ch) in np.argwhere(dh > 1e-4)[:6]:
        ly, lx = int((y + 0.5) / c["scale"]), int((x + 0.5) / c["scale"])
        print(f"   HR ({y}, {x}) ch {ch}: hip {o_own[y, x, ch]:.6f} oracle(HIP flows) {want_h[y, x, ch]:.6f} oracle(HIP flows + r) "
              f"{want_m[y, x, ch]:.6f}; sum of r at LR ({ly}, {lx}): hip {hr_own[:, ly, lx].astype(np.float64).sum():.9f} oracle "
              f"{np.stack(cap_h['r'])[:, ly, lx].astype(np.float64).sum():.9f}")
cfg = fz.config(c, inject_flows=[f for f in oflow])
cfg.debug = True
out, dbg = hsr.main(ref, comp, cfg)
o = out.cpu().numpy()
d = np.where(np.isnan(want), 0.0, np.abs(o.astype(np.float64) - want))
ts, s = c["ts"], c["scale"]
hr = np.stack(dbg["robustness"])
print("max |r_hip - r_oracle| =", float(np.abs(hr - o_r).max()))
for (y, x, ch) in np.argwhere(d > 1e-4)[:8]:
    ly, lx = int((y + 0.5) / s), int((x + 0.5) / s)
    ty, tx = min(ly // ts, oflow.shape[1] - 1), min(lx // ts, oflow.shape[2] - 1)
    print(f"HR ({y}, {x}) ch {ch}: hip {o[y, x, ch]:.6f} oracle {want[y, x, ch]:.6f} diff {d[y, x, ch]:.2e}; LR ({ly}, {lx}) tile ({ty}, {tx})")
    for n in range(oflow.shape[0]):
        win = (slice(max(ly - 2, 0), ly + 3), slice(max(lx - 2, 0), lx + 3))
        print(f"   frame {n}: flow {oflow[n, ty, tx]}, r oracle 5x5 around: min {o_r[n][win].min():.3e} max {o_r[n][win].max():.3e}; "
              f"hip min {hr[n][win].min():.3e} max {hr[n][win].max():.3e}; max |dr| {np.abs(hr[n][win] - o_r[n][win]).max():.2e}")

# own flows: the tiles whose flow differs from the oracle's by more than the ICA noise
cfg = fz.config(c)
cfg.debug = True
_, dbg2 = hsr.main(ref, comp, cfg)
g = np.stack(dbg2["flow"])
df = np.abs(g - oflow).max(-1)
print("own flows: tiles with |flow - oracle flow| > 1e-4 px:")
for (n, ty, tx) in np.argwhere(df > 1e-4):
    print(f"   frame {n} tile ({ty}, {tx}): hip {g[n, ty, tx]} oracle {oflow[n, ty, tx]} diff {df[n, ty, tx]:.3e}")

# taps where exactly one of the two implementations has r == 0 (R = S e - t within rounding of the clamp)
for n in range(oflow.shape[0]):
    z = (hr[n] == 0) != (o_r[n] == 0)
    print(f"frame {n}: {int(z.sum())} raw pixels where exactly one of HIP / oracle has r == 0; there max r = "
          f"{float(np.maximum(hr[n], o_r[n])[z].max()) if z.any() else 0.0:.2e}")
    for (y, x) in np.argwhere(z)[:12]:
        print(f"      raw ({y}, {x}): hip {hr[n, y, x]:.3e} oracle {o_r[n, y, x]:.3e}")
