"""HIP-graph replay of the burst pipeline.

The pipeline enqueues ~280 kernels per 20-frame burst from Python (ctypes + torch allocations): ~9.7 ms of host time at
12 MP — as long as the GPU needs for the kernels themselves — and 5 ms for a 512 x 512 burst whose kernels take 1 ms.
Nothing in main() depends on device results (no host read, no data-dependent launch), so a whole step is captured once
into a HIP graph (stream capture through torch.cuda.graph: the side streams of the frame pipeline fork from / join the
capturing stream through events, which become graph dependencies) and replayed with one launch per burst:
12 MP x 20 x2: 10.6 -> 9.4 ms, 512 x 512 x 20 x2: 5.0 -> 1.0 ms; results bit-identical to the eager path.

A graph is bound to the ADDRESSES of its inputs and to everything the captured Python code decided from shapes and
configuration: GraphRunner keys its graphs by the input tensors (pointer, shape, dtype) and the caller keeps one runner
per configuration.  The first call with a new key runs eagerly (it also creates the per-stream FFT plans and twiddle
tables, which allocate), the second captures, every later one replays.  Outputs are static tensors of the graph's memory
pool: valid until the next call with the same inputs.
"""
import contextlib
import gc
import threading

import torch


_shared = {}  # device index -> (main stream, high-priority upload stream) shared by every runner of the process


def shared_streams(device):
    """The ONE main stream and the ONE upload stream of a device, plus the frame pipeline's side streams, created together.
    HIP streams share a few hardware queues (GPU_MAX_HW_QUEUES, default 4 per priority level, assigned round-robin as the
    streams come into being): every runner that brought its own stream moved the mapping on, until a main stream shared
    its queue with a side stream or with pending copies (measured: the same host-resident burst at 21.5 ms or 26-28 ms
    depending on how many engines the process had created before).  Four normal-priority streams — main + three side
    streams — fit the four queues; the upload stream lives in the high-priority set."""
    from .super_resolution import DEFAULT_STREAMS, _stream_pool

    idx = device.index if device.index is not None else torch.cuda.current_device()
    hit = _shared.get(idx)
    if hit is None:
        dev = torch.device("cuda", idx)
        main = torch.cuda.Stream(dev)
        pool = _stream_pool.setdefault(idx, [])
        while len(pool) < DEFAULT_STREAMS:
            pool.append(torch.cuda.Stream(dev))
        hit = _shared[idx] = (main, torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=-1))
    return hit


def _key(tensors):
    return tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in tensors)


_gc_lock, _gc_depth, _gc_was = threading.Lock(), 0, False


@contextlib.contextmanager
def capture(graph, stream):
    """torch.cuda.graph(graph, stream=...) with the cyclic garbage collector held off while the stream is capturing.
    torch collects once before the capture begins; a collection that an allocation triggers DURING the capture can
    finalise HIP objects of dropped plans / runners that sat in reference cycles (hipGraphDestroy and friends are not
    permitted while a stream of the thread is capturing: the runtime aborts the process — seen as "Fatal Python error:
    Aborted ... Garbage-collecting" in the middle of a RowsPlan capture).  What became garbage meanwhile is collected
    after the capture."""
    # main() may be called from several threads and capture_error_mode is thread-local, so two captures can overlap: the
    # collector goes off with the FIRST and comes back with the LAST of them (a per-call flag let thread A switch it back
    # on while thread B was still capturing — ADVICE r5)
    global _gc_depth, _gc_was
    with _gc_lock:
        if _gc_depth == 0:
            _gc_was = gc.isenabled()
            gc.disable()
        _gc_depth += 1
    try:
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            yield
    finally:
        with _gc_lock:
            _gc_depth -= 1
            if _gc_depth == 0 and _gc_was:
                gc.enable()


class GraphRunner:
    """fn(*tensors) -> pytree of device tensors, replayed from a HIP graph after one eager call per input set.

    `fn` must only enqueue work on torch's current stream (and streams forked from it) and read nothing back to the host.
    Capture failures (an operation that cannot be captured) switch the runner to eager execution for good."""

    MAX_GRAPHS = 4  # input sets kept (each holds a step's intermediates in its memory pool)

    def __init__(self, fn, device):
        self.fn = fn
        self.device = device
        self.stream = shared_streams(device)[0]  # eager warm-up and capture run here: one set of per-stream plans
        self.seen = {}     # key -> number of eager calls
        self.graphs = {}   # key -> (graph, outputs, inputs kept alive)
        self.disabled = False

    def _eager(self, tensors):
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self.fn(*tensors)
        cur.wait_stream(self.stream)
        for t in _leaves(out):
            t.record_stream(cur)
        return out

    def __call__(self, *tensors):
        if self.disabled or not all(torch.is_tensor(t) and t.is_cuda for t in tensors):
            return self.fn(*tensors)
        key = _key(tensors)
        hit = self.graphs.get(key)
        if hit is None:
            if self.seen.get(key, 0) == 0:
                self.seen[key] = 1
                if len(self.seen) > 64:
                    self.seen.pop(next(iter(self.seen)))
                return self._eager(tensors)
            hit = self._capture(key, tensors)
            if hit is None:
                return self.fn(*tensors)
        hit[0].replay()
        return hit[1]

    def _capture(self, key, tensors):
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        graph = torch.cuda.CUDAGraph()
        try:
            # thread_local: only this thread's calls are checked — the RCCL watchdog thread of a multi-GPU job queries
            # events while we capture
            with capture(graph, self.stream):
                out = self.fn(*tensors)
        except Exception as e:  # not capturable in this configuration: stay eager
            self.disabled = True
            self.error = e
            torch.cuda.synchronize(self.device)
            return None
        cur.wait_stream(self.stream)
        while len(self.graphs) >= self.MAX_GRAPHS:
            self.graphs.pop(next(iter(self.graphs)))
        from .utils_image import _grey_plans

        # (the graph holds the FFT plans' spectrum buffers: keep the plan objects alive past the plan cache's eviction)
        self.graphs[key] = (graph, out, tensors, list(_grey_plans.values()))
        return self.graphs[key]


class HostBurstRunner:
    """main() for bursts that START IN HOST MEMORY (the reference's call signature and timer scope,
    super_resolution.py:133-145, 193-195) without the Python launch path: ~280 ctypes launches per burst took the host
    as long as the GPU needs for the kernels, and a graph of the whole step cannot start before the last frame has
    arrived.  Here the step is cut at its natural seams and each piece is a HIP graph over STATIC device staging buffers:

        uploads (eager hipMemcpyAsync, back to back on one upload stream, one event per frame)
        g_ref            reference-frame state                       after the reference frame's copy
        g_chunk[c]       front end + robustness of a chunk of frames after the copies of ITS frames, on side stream c % S
        g_links[k]       (float frames) links of the chained merge   after every chunk of >= 2 frames
        g_merge          (last link of the) fused merge + normalisation   after all chunks

    so the front end of the first chunks runs while later frames are still crossing PCIe, and the host enqueues
    ~20 copies + ~8 graph launches per burst.  Per key (shapes, dtype, frame count): first call eager (creates the
    per-stream FFT plans), second call captures, later calls replay.  Results are bit-identical to the eager path (same
    kernels, same order of the merge).  Frames in PAGEABLE memory (plain NumPy arrays) are first copied into page-locked
    staging by a small thread pool (memcpy at host-memory speed, frame i + 1 while frame i crosses PCIe) — a pageable
    hipMemcpy stages through one bounce buffer on the calling thread instead.
    The call returns when the host buffers may be refilled (all uploads done); the result tensor is produced
    asynchronously on the current stream as usual and belongs to the runner: valid until its next call."""

    COPY_THREADS = 12  # (the MI355X boxes of this project give a container 16 CPUs' worth of quota: more threads get throttled)

    def __init__(self, config, device):
        self.config, self.device = config, device
        self.states = {}
        self.disabled = False
        # the eager calls run on the stream the graphs are captured on and replayed on later: the per-stream FFT plans
        # (which allocate: not capturable) then exist when the capture needs them
        # The upload stream has HIGH priority: HIP streams share a few hardware queues (GPU_MAX_HW_QUEUES, default 4) per
        # priority level.  A stream with 20 copies queued holds its queue with one barrier packet per copy, and every compute
        # stream mapped to the same queue starts after the LAST copy (tools/debug/hwqueue_probe.py: 7 of 8 compute streams
        # blocked for 8.7 ms; with the upload stream in the high-priority queue set: none).  Both streams are shared by
        # all runners of the process (shared_streams).
        self.main, self.up, self.up2 = shared_streams(device)
        import os
        import threading

        self._lock = threading.Lock()  # one burst at a time per runner (static staging, shared streams)

        self.two_up = os.environ.get("HHSR_TWO_UPLOAD_STREAMS") == "1"  # (experiment: copies alternate between two streams)
        # A chunk's copies are queued when the previous chunk's are done, not all 20 up front: with many copies pending on
        # the stream the copy engine sometimes settles at HALF rate (1.6 instead of 0.85 ms per 48 MB for whole bursts,
        # rocprofv3 --memory-copy-trace; seen after compute-only phases of the process, tools/debug/host_leg_timing.py with
        # HHSR_PRELUDE) — queued a chunk at a time it stays at full rate in every state measured, at the price of a gap
        # per chunk (17.3 -> 17.7 ms for 960 MB; 17.3 again with upload_ahead below).  HHSR_PACED_UPLOADS=0: everything up front.
        self.paced_uploads = os.environ.get("HHSR_PACED_UPLOADS", "1") != "0"
        # chunks whose copies are queued AHEAD of the chunk the host is waiting for (paced uploads): with 0 the copy engine
        # idles from a chunk's last copy until the host has woken up, launched the chunk's graph and queued the next copies
        # (~130 us per chunk, 19.3 -> 19.1 ms per float32 burst); 1 and 2 measure the same, "all" is the half-rate mode above
        self.upload_ahead = int(os.environ.get("HHSR_UPLOAD_AHEAD", "1"))

    def _eager(self, ref_img, comp_imgs):
        from .super_resolution import main

        cur = torch.cuda.current_stream(self.device)
        self.main.wait_stream(cur)
        with torch.cuda.stream(self.main):
            out, dbg = main(ref_img, comp_imgs, self.config, _no_runner=True)
        cur.wait_stream(self.main)
        for t in _leaves((out, dbg)):
            t.record_stream(cur)
        return out, dbg

    def close(self):
        """Drop the captured states in a defined order: device idle first, then graphs, staging, copy threads."""
        try:
            if not torch.cuda.is_current_stream_capturing():  # (a finaliser may run in the middle of someone's capture)
                torch.cuda.synchronize(self.device)
        except Exception:
            pass
        for st in list(self.states.values()):
            pool = getattr(st, "pool", None)
            if pool is not None:
                pool.shutdown(wait=True)
        self.states.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def usable(config, ref_img, comp_imgs):
        """Host-resident burst of equal frames in a capturable configuration (no timers / debug / injected flows /
        denoiser: those keep the eager path)."""
        import numpy as np

        hip = config.get("hip", None) if hasattr(config, "get") else None
        if hip is not None and (hip.get("inject_flows", None) is not None or not hip.get("graph", True)
                                or not hip.get("fused_merge", True)):
            return False
        if config.verbose != 0 or config.debug or config.mode != "bayer":
            return False
        den = config.accumulated_robustness_denoiser
        if bool(den.get("enabled", False)) or bool(den.median.enabled or den.gauss.enabled or den.merge.enabled):
            return False
        frames = [ref_img, *[comp_imgs[i] for i in range(len(comp_imgs))]]
        if len(frames) < 2:
            return False
        for f in frames:
            if torch.is_tensor(f):
                if f.is_cuda:
                    return False
            elif not isinstance(f, np.ndarray):
                return False
        t0 = frames[0]
        return all(tuple(f.shape) == tuple(t0.shape) and f.dtype == t0.dtype for f in frames) and len(t0.shape) == 2

    def __call__(self, ref_img, comp_imgs):
        with self._lock:
            return self._call(ref_img, comp_imgs)

    def call_cloned(self, ref_img, comp_imgs):
        """__call__ for callers that may run concurrently (plain main() from several threads): the result is cloned
        INSIDE the runner's lock — the static result tensor is overwritten by the next replay — and the next replay is
        ordered behind that clone even when it is issued from a thread with another current stream."""
        with self._lock:
            out, dbg = self._call(ref_img, comp_imgs)
            if dbg.get("accumulated robustness", None) is not None:
                dbg = dict(dbg, **{"accumulated robustness": dbg["accumulated robustness"].clone()})
            out = out.clone()
            self._consumed = torch.cuda.Event()
            self._consumed.record(torch.cuda.current_stream(self.device))
            return out, dbg

    def _call(self, ref_img, comp_imgs):
        from .super_resolution import main

        frames = [torch.as_tensor(f) for f in (ref_img, *[comp_imgs[i] for i in range(len(comp_imgs))])]
        frames = [f if f.is_contiguous() else f.contiguous() for f in frames]
        key = (tuple(frames[0].shape), frames[0].dtype, len(frames))
        st = self.states.get(key)
        if self.disabled:
            return main(ref_img, comp_imgs, self.config, _no_runner=True)
        if st is None:
            self.states[key] = "seen"
            while len(self.states) > 2:  # (a captured state holds a burst's staging and intermediates: ~6 GB at 12 MP x 20)
                self.states.pop(next(iter(self.states)))
            return self._eager(ref_img, comp_imgs)
        if st == "seen":
            try:
                st = self.states[key] = self._capture(frames)
            except _ConfigError:    # a tuning knob of config.hip that cannot be honoured: the caller's mistake
                raise
            except Exception as e:  # not capturable after all: stay eager
                self.disabled, self.error = True, e
                torch.cuda.synchronize(self.device)
                return main(ref_img, comp_imgs, self.config, _no_runner=True)
        return self._replay(st, frames)

    # ---- capture --------------------------------------------------------------------------------------------------
    def _capture(self, frames):
        from .super_resolution import BurstPipeline, _Staged, _stream_pool, denoiser_enabled
        from .merge import merge_burst, can_fuse_acc_r

        cfg, dev = self.config, self.device
        n = len(frames) - 1
        H, W = frames[0].shape
        st = type("State", (), {})()
        with torch.cuda.device(dev):
            st.stage = torch.empty((n + 1, H, W), dtype=frames[0].dtype, device=dev)
            st.pin = None
            st.main, st.up, st.up2 = self.main, self.up, self.up2
            for i, f in enumerate(frames):  # valid content for the capture-time launches' validation paths
                st.stage[i].copy_(f)
            torch.cuda.synchronize(dev)
            accumulate_r = bool(cfg.robustness.save_mask)
            pipe = st.pipe = BurstPipeline(cfg, dev)
            staged = [_Staged(st.stage[i], None) for i in range(n + 1)]
            st.g_ref = torch.cuda.CUDAGraph()
            with capture(st.g_ref, st.main):
                pipe.init_ref(staged[0])
            sH, sW = pipe.output_size()
            fuse_acc = accumulate_r and can_fuse_acc_r(cfg)
            fuse_min = pipe.fuses_local_min() and (fuse_acc or not accumulate_r)
            acc_r = torch.zeros((H, W), dtype=torch.float32, device=dev) if accumulate_r else None
            if accumulate_r and not fuse_acc:
                raise RuntimeError("accumulated robustness at a non-integer scale: eager path")
            ns = pipe._n_streams(None)
            pool = _stream_pool.setdefault(dev.index, [])
            if len(pool) < ns:
                pool += [torch.cuda.Stream(dev) for _ in range(ns - len(pool))]
            st.chunks = host_chunks(n, pipe._chunk_size())
            hip_ = cfg.get("hip", None) if hasattr(cfg, "get") else None
            sizes = None if hip_ is None else hip_.get("host_chunk_sizes", None)  # explicit chunk sizes (tuning)
            if sizes is not None:
                from ._lib import MAX_BATCH

                if sum(sizes) != n or not all(1 <= int(k) <= MAX_BATCH for k in sizes):
                    raise _ConfigError(f"config.hip.host_chunk_sizes {list(sizes)}: sizes in 1..{MAX_BATCH} that sum to "
                                       f"the {n} compared frames")
                st.chunks, i0 = [], 0
                for k in sizes:
                    st.chunks.append(list(range(i0, i0 + int(k))))
                    i0 += int(k)
            st.streams = [pool[c % ns] for c in range(len(st.chunks))]  # (FFT plans exist per pool stream: eager call)
            if cfg.grey_method == "FFT":  # plans allocate: they have to exist before their stream is captured
                from . import _lib
                from .utils_image import _grey_plan

                for idx, s in zip(st.chunks, st.streams):
                    with torch.cuda.stream(s):
                        _grey_plan(H, W, dev, _lib.MAX_BATCH if len(idx) > 1 else 1)
            st.g_chunks, results = [], []
            for idx, s in zip(st.chunks, st.streams):
                g = torch.cuda.CUDAGraph()
                with capture(g, s):
                    fronts = pipe._front_chunk([staged[1 + i] for i in idx], None, idx, None)
                    results.append(pipe._robustness(fronts, None, fuse_min))
                st.g_chunks.append(g)
            st.frames = [f for chunk in results for f in chunk]
            # The merge needs the LAST frame, and everything after the last upload is on the critical path.  Bursts whose
            # uploads take longer than their kernels (float frames: 0.87 ms of PCIe per 12 MP frame against 0.46 ms of GPU
            # work; uint16 counts are the other way round) leave the GPU idle between chunks: there the merge is CHAINED
            # (merge.merge_burst_chain, bit-identical to the single launch) — after every chunk of >= 2 frames a link adds
            # that chunk's frames to the parked class accumulators while later frames are crossing PCIe, and the last link
            # only has the final single-frame chunks, the reference frame and the normalisation left (12 MP x 20: 1.4 ms
            # instead of 4.0 after the last frame's front end).  Each link moves the 1.6 GB of accumulators twice (~0.7 ms
            # of otherwise idle GPU time): not for GPU-bound bursts.
            from .merge import can_chain, chain_buffer, merge_burst_chain

            hip = cfg.get("hip", None) if hasattr(cfg, "get") else None
            want_chain = (frames[0].dtype.itemsize >= 4) if hip is None or hip.get("merge_chain", None) is None \
                else bool(hip.get("merge_chain"))
            st.links = []  # (index of the chunk after which the link runs, frames merged once it has run)
            if want_chain and fuse_min and can_chain(cfg, (H, W)) and len(st.chunks) > 2:
                done = 0
                after = None if hip is None else hip.get("merge_link_after", None)  # explicit chunk indices (tuning)
                for c, idx in enumerate(st.chunks[:-1]):
                    done = idx[-1] + 1
                    if (len(idx) >= 2) if after is None else (c in after):
                        st.links.append((c, done))
            st.chain = bool(st.links)
            st.g_links = []
            st.num = torch.empty((sH, sW, 3), dtype=torch.float32, device=dev)
            if st.chain:
                st.cls = chain_buffer((H, W), dev)
                prev = 0
                for c, done in st.links:
                    g = torch.cuda.CUDAGraph()
                    with capture(g, st.main):
                        merge_burst_chain(st.frames[:done], prev, pipe.ref, pipe.ref_covs, st.num, pipe.cfa, cfg, st.cls, False,
                                          local_min=fuse_min)
                    st.g_links.append(g)
                    prev = done
            st.g_merge = torch.cuda.CUDAGraph()
            with capture(st.g_merge, st.main):
                if st.chain:
                    merge_burst_chain(st.frames, st.links[-1][1], pipe.ref, pipe.ref_covs, st.num, pipe.cfa, cfg, st.cls, True,
                                      acc_r=acc_r, local_min=fuse_min)
                else:
                    merge_burst(st.frames, pipe.ref, pipe.ref_covs, st.num, None, pipe.cfa, cfg, do_ref=True, divide=True,
                                acc_r=acc_r, local_min=fuse_min)
            st.e_chunk = [torch.cuda.Event() for _ in st.chunks]
            st.acc_r = acc_r
            st.e_up = [torch.cuda.Event() for _ in range(n + 1)]
            st.e_ref = torch.cuda.Event()
            from .utils_image import _grey_plans

            st.plans = list(_grey_plans.values())  # the graphs hold the plans' spectrum buffers: keep them alive
        return st

    # ---- replay ---------------------------------------------------------------------------------------------------
    def _replay(self, st, frames):
        import numpy as np

        dev = self.device
        cur = torch.cuda.current_stream(dev)
        n = len(frames) - 1
        pinned = all(f.is_pinned() for f in frames)
        futs = None
        if not pinned:
            # pageable frames -> page-locked staging.  Every frame is cut into COPY_THREADS row slabs that the pool copies
            # in frame order (np.copyto releases the GIL): frame i is complete after (i + 1) x 48 MB / (aggregate memcpy
            # bandwidth) and starts crossing PCIe while the threads are on frame i + 1
            from concurrent.futures import ThreadPoolExecutor

            if st.pin is None:
                st.pin = torch.empty(tuple(st.stage.shape), dtype=st.stage.dtype, pin_memory=True)
                st.pin_np = st.pin.numpy()
                st.pool = ThreadPoolExecutor(self.COPY_THREADS)
            # (the previous burst's DMA out of the staging finished before its call returned)
            H = st.stage.shape[1]
            cuts = [H * k // self.COPY_THREADS for k in range(self.COPY_THREADS + 1)]
            futs = []
            for i, f in enumerate(frames):
                if f.is_pinned():
                    futs.append(None)
                    continue
                src = f.numpy()
                futs.append([st.pool.submit(np.copyto, st.pin_np[i, a:b], src[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a])
        with torch.cuda.device(dev):
            st.up.wait_stream(st.main)   # the previous burst's kernels are done with the device staging buffers
            if self.two_up:
                st.up2.wait_stream(st.main)
            st.main.wait_stream(cur)
            consumed = getattr(self, "_consumed", None)
            if consumed is not None:  # the previous call's clone of the static result (call_cloned), on whatever stream
                st.main.wait_event(consumed)
                self._consumed = None

            def upload(i):
                src = frames[i]
                if futs is not None and futs[i] is not None:
                    for f in futs[i]:
                        f.result()
                    src = st.pin[i]
                up = st.up2 if (self.two_up and i % 2) else st.up
                with torch.cuda.stream(up):
                    st.stage[i].copy_(src, non_blocking=True)
                    st.e_up[i].record(up)

            upload(0)
            paced = futs is not None or self.paced_uploads
            if not paced:  # (page-locked frames, HHSR_PACED_UPLOADS=0: all copies queued up front, back to back)
                for i in range(1, n + 1):
                    upload(i)
            # The graphs are launched HOST-PACED: the host waits for a chunk's last copy, then launches its graph.  Queued
            # up front behind event waits they did not start before the LAST copy had finished (measured: first kernel at
            # 9.0 ms of a 9.0 ms upload sequence) — HIP streams share a few hardware queues, and a barrier packet that
            # waits for a late copy blocks everything behind it in its queue, other streams' runnable work included.  The
            # call blocks until the uploads are done anyway (the host buffers are the caller's again when it returns).
            st.e_up[0].synchronize()
            with torch.cuda.stream(st.main):
                st.g_ref.replay()
                st.e_ref.record(st.main)
            queued = 0  # chunks whose copies are queued
            for c, (idx, s, g) in enumerate(zip(st.chunks, st.streams, st.g_chunks)):
                while paced and queued < len(st.chunks) and queued <= c + (self.upload_ahead if futs is None else 0):
                    for i in st.chunks[queued]:
                        upload(1 + i)
                    queued += 1
                for i in (idx if self.two_up else idx[-1:]):
                    st.e_up[1 + i].synchronize()  # (one upload stream: copies complete in order)
                with torch.cuda.stream(s):
                    s.wait_event(st.e_ref)
                    g.replay()
                    st.e_chunk[c].record(s)
                for (lc, _), gl in zip(st.links, st.g_links):
                    if lc == c:  # the link after this chunk: everything it waits for is launched (short waits only)
                        with torch.cuda.stream(st.main):
                            for k in range(c + 1):
                                st.main.wait_event(st.e_chunk[k])
                            gl.replay()
            with torch.cuda.stream(st.main):
                for e in st.e_chunk:
                    st.main.wait_event(e)
                st.g_merge.replay()
            cur.wait_stream(st.main)
        debug = {"robustness": [], "flow": []}
        if st.acc_r is not None:
            debug["accumulated robustness"] = st.acc_r
        return st.num, debug


class _ConfigError(ValueError):
    pass


def host_chunks(n, size):
    """Chunk sizes for frames that arrive one after the other over PCIe: chunks of `size` first (one launch per stage and
    chunk: efficient while later frames are still in flight), then 3, 2, 1, 1 — what runs after the LAST frame has arrived
    is on the critical path of the burst, so the last chunks are short.  Returns lists of frame indices."""
    tail = []
    for t in (1, 1, 2, 3):
        if n - sum(tail) - t >= size:
            tail.append(t)
    body, rem = [], n - sum(tail)
    while rem > 0:
        body.append(min(size, rem))
        rem -= body[-1]
    if len(body) > 1 and body[-1] < body[-2]:  # the remainder chunk goes first of the short ones: sizes never grow
        pass
    sizes = body + tail[::-1]
    out, i = [], 0
    for k in sizes:
        out.append(list(range(i, i + k)))
        i += k
    return out


def _leaves(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _leaves(v)
    elif isinstance(x, (tuple, list)):
        for v in x:
            yield from _leaves(v)


def signature(config):
    """Cheap fingerprint of everything in a configuration that the captured Python code may have decided from: all scalar
    settings; long lists (the two 1001-entry noise curves) by identity, length and ends.  An engine drops its graphs
    when the fingerprint of its configuration changes (the reference mutates configs in place)."""
    def walk(v):
        if isinstance(v, dict):
            return tuple((k, walk(x)) for k, x in v.items())
        if isinstance(v, (list, tuple)):
            if len(v) > 32:  # the 1001-entry noise curves: identity + ends (a full walk would cost 0.2 ms per burst)
                return ("list", id(v), len(v), repr(v[0]), repr(v[-1]))
            return tuple(walk(x) for x in v)
        if isinstance(v, (int, float, str, bool, type(None))):
            return v
        return ("object", id(v))

    return walk(config)


class ConfigWatch:
    """O(#nested mappings) check that a configuration has not been edited in place since the last call.  Config trees:
    the tuple of (identity, edit counter) of every mapping of the tree plus the content of its short lists (in-place
    list edits such as `tile_sizes[0] = 8` bypass the counters; the two 1001-entry noise curves are taken by identity,
    length and ends) — a replaced nested mapping changes the identities, so edits cannot cancel out the way a sum of
    counters could.  Other mapping types (a real OmegaConf DictConfig) are fingerprinted in full with signature()."""

    def __init__(self):
        self.state = None
        self.config = None  # keeps the watched object alive: its id() cannot be reused

    def changed(self, config):
        from .config import Config

        state = self._state(config) if isinstance(config, Config) else signature(config)
        first = self.state is None or self.config is not config
        same = state == self.state and not first
        self.state, self.config = state, config
        return not same and not first

    @staticmethod
    def _state(config):
        from .config import Config

        out, stack = [], [config]
        while stack:
            c = stack.pop()
            out.append((id(c), c.version()))
            for v in c.values():
                if isinstance(v, Config):
                    stack.append(v)
                elif isinstance(v, list):
                    if len(v) > 32:
                        out.append(("list", id(v), len(v), repr(v[0]), repr(v[-1])))
                    else:
                        out.append(tuple(x if isinstance(x, (int, float, str, bool, type(None))) else id(x) for x in v))
                        stack.extend(x for x in v if isinstance(x, Config))
        return tuple(out)


def capturable(config, tensors):
    """main() can be captured: no host-synchronising timers / debug copies / injected host arrays, device inputs."""
    hip = config.get("hip", None) if hasattr(config, "get") else None
    if hip is not None and (hip.get("inject_flows", None) is not None or not hip.get("graph", True)):
        return False
    return config.verbose == 0 and not config.debug and all(torch.is_tensor(t) and t.is_cuda for t in tensors)
