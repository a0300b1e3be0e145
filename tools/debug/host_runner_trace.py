"""Host / device timeline of graph.HostBurstRunner._replay: host timestamps at every step + timing events on the streams.
   python tools/debug/host_runner_trace.py [u16|f32]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "handheld-multi-frame-super-resolution_amd"))
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth, distributed as hdist, graph as hgraph

kind = sys.argv[1] if len(sys.argv) > 1 else "u16"
dev = torch.device("cuda", 0)
H, W, NF = 3000, 4000, 20
ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
cfg = hsr.default_config()
cfg.verbose = 0
cfg.scale = 2
black, white = 64.0, 1023.0
cfg.hip = {"raw_norm": {"black_levels": [black] * 3, "white_level": white}} if kind == "u16" else {}
hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                   [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
conv = (lambda t: torch.from_numpy(np.clip(np.rint(t.cpu().numpy() * (white - black) + black), 0, white).astype(np.uint16)).pin_memory()) \
    if kind == "u16" else (lambda t: t.cpu().pin_memory())
ref_h, comp_h = conv(ref), [conv(comp[i]) for i in range(NF - 1)]
del ref, comp
eng = hdist.HipEngine(cfg)
for i in range(3):
    hdist.main_sharded(ref_h, comp_h, cfg, engine=eng)
torch.cuda.synchronize()

# instrument: wrap CUDAGraph.replay and Event.synchronize
log = []
t0 = [0.0]
orig_replay = torch.cuda.CUDAGraph.replay
orig_sync = torch.cuda.Event.synchronize
marks = []


def replay(self):
    s = torch.cuda.current_stream()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record(s)
    th = time.perf_counter() - t0[0]
    orig_replay(self)
    b.record(s)
    marks.append((th, time.perf_counter() - t0[0], a, b))


def esync(self):
    th = time.perf_counter() - t0[0]
    orig_sync(self)
    log.append((th, time.perf_counter() - t0[0]))


torch.cuda.CUDAGraph.replay = replay
torch.cuda.Event.synchronize = esync
for step in range(2):
    log.clear(); marks.clear()
    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    base.record(torch.cuda.current_stream())
    t0[0] = time.perf_counter()
    hdist.main_sharded(ref_h, comp_h, cfg, engine=eng)
    th = time.perf_counter() - t0[0]
    torch.cuda.synchronize()
    print(f"step {step}: call returned after {1e3 * th:.2f} ms, all done after {1e3 * (time.perf_counter() - t0[0]):.2f} ms")
    print("  host event waits (start -> end ms): " + "  ".join(f"{1e3 * a:.2f}->{1e3 * b:.2f}" for a, b in log))
    for th0, th1, a, b in marks:
        print(f"  graph launch: host {1e3 * th0:.2f}->{1e3 * th1:.2f} ms; device start {base.elapsed_time(a):.2f} end {base.elapsed_time(b):.2f} ms")
