"""Timeline of one H2D-inclusive step (host uint16 frames): run under
   rocprofv3 --kernel-trace --memory-copy-trace -d <dir> -o t -- python tools/debug/h2d_trace.py run [on|off]
then   python tools/debug/h2d_trace.py report <results.db>"""
import sys, time


def run(split):
    import copy
    import numpy as np
    import torch
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "handheld-multi-frame-super-resolution_amd"))
    import handheld_super_resolution as hsr
    from handheld_super_resolution import synthetic as synth, distributed as hdist

    dev = torch.device("cuda", 0)
    H, W, NF = 3000, 4000, 20
    ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
    cfg = hsr.default_config()
    cfg.verbose = 0
    cfg.scale = 2
    black, white = 64.0, 1023.0
    cfg.hip = {"raw_norm": {"black_levels": [black] * 3, "white_level": white}} if split != "f32" else {}
    if os.environ.get("HHSR_CHUNK"):
        cfg.hip["chunk"] = int(os.environ["HHSR_CHUNK"])
    if os.environ.get("HHSR_STREAMS"):
        cfg.hip["streams"] = int(os.environ["HHSR_STREAMS"])
    hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                       [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
    to_counts = lambda t: torch.from_numpy(np.clip(np.rint(t.cpu().numpy() * (white - black) + black), 0, white)
                                           .astype(np.uint16)).pin_memory()
    if split == "f32":
        to_counts = lambda t: t.cpu().pin_memory()
    ref16, comp16 = to_counts(ref), [to_counts(comp[i]) for i in range(NF - 1)]
    del ref, comp
    eng = hdist.HipEngine(cfg)
    for i in range(5):
        torch.cuda.synchronize()
        time.sleep(0.05)
        t0 = time.perf_counter()
        hdist.main_sharded(ref16, comp16, cfg, engine=eng)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"step {i}: host enqueue {1e3 * (t1 - t0):.2f} ms, total {1e3 * (time.perf_counter() - t0):.2f} ms")


def report(db):
    import sqlite3
    con = sqlite3.connect(db)
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    ks = con.execute("select name, start, end from kernels order by start").fetchall()
    mc = [n for n in names if "memory_cop" in n and "rocpd" not in n]
    print("copy tables:", mc)
    cols = [r[1] for r in con.execute(f"pragma table_info({mc[0]})")]
    print(cols)
    cs = con.execute(f"select start, end, size from {mc[0]} order by start").fetchall()
    ev = sorted([(k[1], k[2], k[0].split("(")[0][-40:]) for k in ks] + [(c[0], c[1], "COPY %d" % c[2]) for c in cs])
    # last cluster (gap > 20 ms before it)
    start = 0
    for i in range(1, len(ev)):
        if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 20e6:
            start = i
    step = ev[start:]
    t0 = step[0][0]
    tend = max(e[1] for e in step)
    print(f"step span {(tend - t0) / 1e6:.3f} ms, {len(step)} events")
    big = [e for e in step if e[2].startswith("COPY") and int(e[2].split()[1]) > 1e6]
    for e in big:
        print(f"  copy {(e[0] - t0) / 1e6:7.3f} .. {(e[1] - t0) / 1e6:7.3f} ms  {int(e[2].split()[1]) / 1e6:.1f} MB")
    for e in step:
        if "k_merge" in e[2] or "k_ref_planes" in e[2] or "k_rows_fwd" in e[2] or "k_rob_frames" in e[2]:
            print(f"  {e[2]:40s} {(e[0] - t0) / 1e6:7.3f} .. {(e[1] - t0) / 1e6:7.3f} ms")
    kk = sorted((e[0], e[1]) for e in step if not e[2].startswith("COPY"))
    busy, last = 0, t0
    for a, b in kk:
        a = max(a, last)
        if b > a:
            busy += b - a
            last = b
    print(f"kernel busy (union) {busy / 1e6:.3f} ms")
    # per-ms busy histogram
    nb = int((tend - t0) / 1e6) + 1
    hist = [0.0] * nb
    last = t0
    for a, b in kk:
        a = max(a, last)
        while a < b:
            bi = int((a - t0) / 1e6)
            e = min(b, t0 + (bi + 1) * 1e6)
            hist[bi] += e - a
            a = e
        last = max(last, b)
    print("busy fraction per ms:", " ".join(f"{h / 1e6:.2f}" for h in hist))


if __name__ == "__main__":
    run(sys.argv[2] if len(sys.argv) > 2 else "auto") if sys.argv[1] == "run" else report(sys.argv[2])
