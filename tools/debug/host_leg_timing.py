"""Time per burst of the host-resident legs (pinned float32 / pinned uint16 / pageable float32), back to back and with a
device synchronisation between bursts.   python tools/debug/host_leg_timing.py"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "handheld-multi-frame-super-resolution_amd"))
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth, distributed as hdist

dev = torch.device("cuda", 0)
H, W, NF = 3000, 4000, 20
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
black, white = 64.0, 1023.0


def config(u16):
    cfg = hsr.default_config()
    cfg.verbose = 0
    cfg.scale = 2
    cfg.hip = {"raw_norm": {"black_levels": [black] * 3, "white_level": white}} if u16 else {}
    for k in ("HHSR_CHUNK", "HHSR_STREAMS"):
        if os.environ.get(k):
            cfg.hip[k[5:].lower()] = int(os.environ[k])
    if os.environ.get("HHSR_MERGE_CHAIN"):
        cfg.hip["merge_chain"] = os.environ["HHSR_MERGE_CHAIN"] == "1"
    if os.environ.get("HHSR_HOST_CHUNKS"):
        cfg.hip["host_chunk_sizes"] = [int(v) for v in os.environ["HHSR_HOST_CHUNKS"].split(",")]
    if os.environ.get("HHSR_LINK_AFTER"):
        cfg.hip["merge_link_after"] = [int(v) for v in os.environ["HHSR_LINK_AFTER"].split(",")]
    hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                       [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
    return cfg


if os.environ.get("HHSR_PRELUDE"):  # what bench.py does before its host-resident legs: device-resident graph replays + eager steps
    cfg0 = config(False)
    e0 = hdist.HipEngine(cfg0)
    for _ in range(8):
        hdist.main_sharded(ref, comp, cfg0, engine=e0)
    import copy
    cfg1 = copy.deepcopy(cfg0)
    cfg1.hip = dict(cfg1.hip, graph=False)
    e1 = hdist.HipEngine(cfg1)
    for _ in range(4):
        hdist.main_sharded(ref, comp, cfg1, engine=e1)
    torch.cuda.synchronize()
    print("prelude done (resident graph engine kept alive:", os.environ["HHSR_PRELUDE"] == "keep", ")")
    if os.environ["HHSR_PRELUDE"] != "keep":
        del e0, e1

c16 = lambda t: torch.from_numpy(np.clip(np.rint(t.cpu().numpy() * (white - black) + black), 0, white).astype(np.uint16))
legs = {
    "pinned f32": (False, ref.cpu().pin_memory(), [comp[i].cpu().pin_memory() for i in range(NF - 1)]),
    "pinned u16": (True, c16(ref).pin_memory(), [c16(comp[i]).pin_memory() for i in range(NF - 1)]),
    "pageable f32 (numpy)": (False, ref.cpu().numpy(), comp.cpu().numpy()),
    "pageable u16 (numpy)": (True, c16(ref).numpy(), np.stack([c16(comp[i]).numpy() for i in range(NF - 1)])),
}
def arena(r, c):  # the same frames as views of ONE page-locked allocation
    a = torch.empty((len(c) + 1, *r.shape), dtype=r.dtype, pin_memory=True)
    a[0].copy_(r)
    for i, f in enumerate(c):
        a[i + 1].copy_(f)
    return a[0], list(a[1:].unbind(0))


legs["pinned f32, ONE arena"] = (False, *arena(legs["pinned f32"][1], legs["pinned f32"][2]))
legs["pinned u16, ONE arena"] = (True, *arena(legs["pinned u16"][1], legs["pinned u16"][2]))
import threading

_stop = threading.Event()


def _busy():  # host memory traffic: keeps the host's data fabric out of its idle power state
    a, b = np.zeros(8 << 20, np.float32), np.zeros(8 << 20, np.float32)
    while not _stop.is_set():
        np.copyto(a, b)


if os.environ.get("HHSR_HOST_BUSY"):
    for _ in range(int(os.environ["HHSR_HOST_BUSY"])):
        threading.Thread(target=_busy, daemon=True).start()
    print("host busy threads:", os.environ["HHSR_HOST_BUSY"])

only = os.environ.get("HHSR_LEGS")  # substring filter, e.g. "pinned u16"
for name, (u16, r, c) in legs.items():
    if only and only not in name + ":":
        continue
    cfg = config(u16)
    eng = hdist.HipEngine(cfg)
    for _ in range(3):
        hdist.main_sharded(r, c, cfg, engine=eng)
    torch.cuda.synchronize()
    K = 8
    t0 = time.perf_counter()
    for _ in range(K):
        hdist.main_sharded(r, c, cfg, engine=eng)
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / K
    ts, th = [], []
    for _ in range(K):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hdist.main_sharded(r, c, cfg, engine=eng)
        th.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{name}: back to back {1e3 * b2b:.2f} ms / burst; one at a time {1e3 * np.median(ts):.2f} ms (call returns after "
          f"{1e3 * np.median(th):.2f} ms); graphs={not eng._host.disabled}")
    del eng
