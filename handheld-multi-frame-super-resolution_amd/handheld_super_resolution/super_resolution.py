"""Alg. 1 driver: main() and the process() facade (reference super_resolution.py:41-360).

MI355X design notes (vs the reference's per-frame Python loop):
  * everything is enqueued on ONE HIP stream (torch's current stream): no host synchronisation inside
    the hot path, no torch<->Numba stream fences;
  * each frame is uploaded once (the reference uploads it twice);
  * a frame's (raw, flow, covariances, robustness) stay resident in HBM (144 MB per 12 MP frame; a
    20-frame burst is <3 GB of the 288 GB) and the merge of the whole burst is ONE kernel with the
    accumulators in registers (merge.merge_burst), instead of a read-modify-write of the 2x576 MB
    accumulators per frame;
  * the front end (grey FFT, pyramid, alignment levels, raw pass) runs one launch per STAGE AND CHUNK of frames, chunks
    round-robin on two HIP streams (one for frames of 40 MP and more); bursts that start in host memory run as eager uploads + per-chunk HIP graphs
    (graph.HostBurstRunner), device-resident bursts as one graph per burst (graph.GraphRunner);
  * multi-GPU (distributed.py): alignment frame-parallel, one all-gather of the flow fields, robustness / kernels /
    merge row-parallel (default), or frames one per GPU with one reduce-scatter of the num / den accumulators.
"""
import os
import time

import numpy as np
import torch

from . import _lib
from .utils_image import compute_grey_images, compute_grey_images_batch
from .utils import divide, add, getTime, timer
from .alignment import (align, init_alignment, build_gaussian_pyramid, build_gaussian_pyramids, align_batch,
                        can_align_batch)
from .params import sanitize_config, update_snr_config
from .robustness import (init_robustness, compute_robustness, compute_robustness_group, noise_curves_to_device, RobustnessSum,
                         ref_planes, upscale_warp_stats, mono_sigma_sq)
from .kernels import estimate_kernels, frame_stats, frame_stats_batch
from .merge import merge, merge_ref, merge_burst, can_fuse_acc_r, can_fuse_local_min


def denoiser_enabled(config):
    """config.accumulated_robustness_denoiser.enabled is derived by process() upstream
    (super_resolution.py:291-296); derive it here too when main() is called directly."""
    den = config.accumulated_robustness_denoiser
    if "enabled" in den:
        return bool(den.enabled)
    den.enabled = bool(den.median.enabled or den.gauss.enabled or den.merge.enabled)
    return den.enabled


def n_images_of(comp_imgs):
    return len(comp_imgs)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("handheld_super_resolution (MI355X build): no HIP device is visible; this package has no "
                           "CPU path (the NumPy oracle under oracle/ is test infrastructure only)")
    return torch.device("cuda", torch.cuda.current_device())


ROB_GROUP = 4  # frames per robustness launch (hhsr_rob_frames)


def _tensors(x):
    """All tensors in a nested tuple / list."""
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (tuple, list)):
        for y in x:
            yield from _tensors(y)


DEFAULT_STREAMS = 2  # round 1, 12 MP x 20: 1 stream 15.3 ms, 2: 14.3, 3: 13.9, 4: 14.3 -> 3.  Round 6, with the batched front end and the
#                      graph-replayed step (four alternating bench runs each, one MI355X): 2 streams 7.70 - 7.80 ms, 3: 7.76 - 7.88
#                      (round 5's sweep had said the same: 8.21 against 8.36); eager, host-resident and denoiser legs the same or better
LARGE_FRAME = 40_000_000  # raw pixels from which the frame pipeline runs on ONE stream: every kernel of a 48 MP frame fills the
#                           GPU on its own and side streams only contend for the caches (round 6, one MI355X, 20 frames: 48 MP x3
#                           51.9 ms on one stream / 52.5 on three, 48 MP x2 32.4 / 32.7; 30 MP x2 21.7 / 21.2, 24 MP 17.9 / 17.1,
#                           12 MP 8.33 / 8.24: the smaller frames keep several)
_stream_pool = {}  # device index -> side streams, shared by all pipelines of the process


class _Staged:
    """A host frame whose upload is in flight on the upload stream: device tensor (dtype of the host frame) + event."""
    __slots__ = ("tensor", "event")

    def __init__(self, tensor, event):
        self.tensor, self.event = tensor, event


_upload_streams = {}  # device index -> the stream all prefetched uploads are queued on, in frame order


_LATE_FORK = os.environ.get("HHSR_LATE_FORK") is not None  # A/B switch (read once): side streams fork behind the reference precompute


class BurstPipeline:
    """Device-resident state of one burst: reference-frame precompute + per-frame stage chain."""

    def __init__(self, config, device=None):
        self.config = config
        self.device = device or _device()
        # `mode: grey` (monochrome sensors): the frames are their own grey images (super_resolution.py:106-109, 144-147),
        # one-channel robustness, per-pixel kernels, one-channel merge — the generic kernels of each stage
        self.mono = config.mode != "bayer"
        self.cfa = [[int(v) for v in row] for row in config.exif.cfa_pattern]
        self.wb = [float(v) for v in config.exif.white_balance]
        self.curves = noise_curves_to_device(config.noise_model.std_curve, config.noise_model.diff_curve, self.device)
        self.grey_method = config.grey_method
        self.ref = None
        self._streams = _stream_pool.setdefault(self.device.index, [])
        hip = config.get("hip", None) if hasattr(config, "get") else None
        # validation hook (bench.py's parity attribution, tests): per-frame flow fields that replace align()
        self._inject_flows = hip.get("inject_flows", None) if hip is not None else None
        # frames given as integer sensor counts: {"black_levels": [R, G, B], "white_level": w} (see _ingest)
        self._raw_norm = hip.get("raw_norm", None) if hip is not None else None
        # chunk-batched front end (one launch per stage for a chunk of frames, see _front_chunk); config.hip.batch: false
        # keeps the per-frame launches (A/B, tests)
        # (before, after): the flow fields handed to this pipeline are row slices (views) of larger fields with that many
        # tile rows around them — the sub-image pipelines of distributed.py; see robustness.compute_s
        self.flow_rows = (0, 0)
        # False: the side streams do not wait for the reference-frame events — the caller orders the reference precompute
        # before the frames itself (distributed.RowsPlan captures them as separate graphs: an event recorded in one
        # capture cannot be waited for in another)
        self.ref_wait = True
        self._batch = (True if hip is None else bool(hip.get("batch", True))) \
            and (self.mono or self.grey_method == "FFT") and config.verbose < 2

    def _ingest(self, img):
        """One frame -> float32 device tensor, on the current stream.  Float frames are the reference's call signature
        (normalised, white-balanced RAW).  Integer frames are sensor counts as the DNG holds them: they are uploaded as
        they are — half the PCIe bytes of the float32 frame — and normalised on the device with the reference loader's
        arithmetic (utils_dng.py:149-160, hhsr_normalize_raw_u16) using config.hip.raw_norm and config.exif."""
        staged = isinstance(img, _Staged)
        if staged:  # prefetch(): the copy was queued on the upload stream
            if img.event is not None:  # (None: a static staging buffer of graph.HostBurstRunner, ordered by the runner)
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(img.event)
                img.tensor.record_stream(cur)
            img = img.tensor
        t = img if torch.is_tensor(img) else torch.as_tensor(img)
        if t.dtype.is_floating_point:
            return _lib.f32c(t, self.device)
        if (t.is_cuda and not staged) or self._raw_norm is None:
            raise ValueError("integer frames are sensor counts: give config.hip.raw_norm = {black_levels, white_level} "
                             "(host arrays), or pass normalised float frames")
        from .utils_dng import normalize_burst

        return normalize_burst(t, self._raw_norm["black_levels"], self._raw_norm["white_level"], self.wb, self.cfa,
                               device=self.device)

    def prefetch(self, imgs):
        """Queue the uploads of all page-locked host frames of a list NOW, back to back on one upload stream, instead of
        one by one as the host gets to each frame's kernels: enqueuing a frame's ~13 launches takes the host longer
        (0.5-0.7 ms) than the copy engine needs for the frame (0.43 ms for 24 MB of uint16 counts), so host-paced
        uploads leave the link idle half of the time (measured: last copy done at 13.9 ms instead of 8.5).  Frames in
        pageable memory keep the per-frame order (their copies block the host; the GPU works on frame i meanwhile)."""
        out, stream = [], None
        for img in imgs:
            t = img if torch.is_tensor(img) else (torch.as_tensor(img) if isinstance(img, np.ndarray) else None)
            if t is None or t.is_cuda or not t.is_pinned():
                out.append(img)
                continue
            if stream is None:
                stream = _upload_streams.get(self.device.index)
                if stream is None:
                    # (high priority = its own hardware queue: see graph.HostBurstRunner)
                    from .graph import shared_streams

                    stream = _upload_streams[self.device.index] = shared_streams(self.device)[1]
            if t.dtype == torch.float64:  # (half the PCIe bytes: the kernels work on float32 frames anyway)
                t = t.to(torch.float32).pin_memory()
            with torch.cuda.stream(stream):
                d = t.contiguous().to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
            self._last_upload = ev
            out.append(_Staged(d, ev))
        return out

    def uploads_done(self):
        """Block until the last queued upload has left the host: the caller's page-locked buffers may be refilled."""
        ev = getattr(self, "_last_upload", None)
        if ev is not None:
            ev.synchronize()
            self._last_upload = None

    def _timed(self, func, level, start_s=None, end_s=None):
        """The reference's per-stage timers (super_resolution.py:72-81): synchronising wall-clock wrappers, active from
        config.verbose >= level (2: stages, 3: grey images), plain function otherwise."""
        return timer(func, self.config.verbose >= level, start_s, end_s)

    def init_ref(self, ref_img, alignment=True, robustness=True):
        """Reference-frame precompute.  `alignment=False` / `robustness=False` skip the halves the multi-GPU
        path does not need on a given pipeline (distributed.py: whole-frame alignment state vs the sub-image's
        robustness / kernel state)."""
        with torch.cuda.device(self.device):  # every launch below goes to this device's current stream
            return self._init_ref(ref_img, alignment, robustness)

    def _init_ref(self, ref_img, alignment, robustness):
        cfg = self.config
        main = torch.cuda.current_stream(self.device)
        self._entry = torch.cuda.Event()  # everything the caller enqueued before (e.g. the frames' upload) is done
        self._entry.record(main)
        self._entry_fresh = True  # the next _on_streams() forks its side streams HERE, not behind the reference precompute
        self.ref = self._ingest(ref_img)
        self.align_state = self.grey_ref = None
        if alignment:
            sanitize_config(cfg, tuple(self.ref.shape))
            grey = self.ref if self.mono else self._timed(
                compute_grey_images, 3, end_s="- Ref grey image estimated by {}".format(self.grey_method))(
                self.ref, self.grey_method)
            self.align_state = self._timed(init_alignment, 2, "\nInitializing alignment", "Alignment initialized (Total)")(
                grey, cfg)
            self.grey_ref = grey
        # two milestones of the reference precompute: the frames' alignment only needs the first (grey image, pyramid,
        # gradients, Hessians: ~0.25 ms); the robustness state behind it (raw pass + upsampled planes, another ~0.25 ms)
        # is first needed after a frame's alignment and raw pass — the side streams do not sit out the second half
        # (measured: no change of the 12 MP step, the other frames' FFTs already fill that time; kept for short bursts)
        self._align_ready = torch.cuda.Event()
        self._align_ready.record(main)
        self.ref_means = self.ref_vars = self.ref_covs = self.ref_sigma_sq = None
        if robustness and cfg.robustness.enabled and self.mono:
            m, v, self.ref_covs = self._timed(frame_stats, 2, "\nEstimating ref image local stats + kernels",
                                              "Local stats + kernels estimated (Total)")(
                self.ref, self.cfa, self.wb, cfg, want_vars=True)
            self.ref_means, self.ref_vars = upscale_warp_stats(m), upscale_warp_stats(v)
            self.ref_sigma_sq = mono_sigma_sq(self.ref_means, self.ref_vars, self.curves[0])
        elif robustness and cfg.robustness.enabled:  # init_robustness + the reference frame's kernels from one raw pass,
            # then the upsampled means, sigma^2 and curve indices from one pass over the guide statistics
            m, v, self.ref_covs = self._timed(frame_stats, 2, "\nEstimating ref image local stats + kernels",
                                              "Local stats + kernels estimated (Total)")(
                self.ref, self.cfa, self.wb, cfg, want_vars=True)
            self.ref_means, self.ref_sigma_sq = ref_planes(m, v, self.curves[0])
            self.ref_vars = None  # only needed for sigma^2, which is already there
        elif robustness:
            self.ref_means, self.ref_vars = init_robustness(self.ref, self.cfa, self.wb, cfg)
            self.ref_covs = estimate_kernels(self.ref, cfg)
        self._ref_ready = torch.cuda.Event()
        self._ref_ready.record(main)
        return self

    def flow_grid(self):
        """(ny, nx) of the flow field: tiles of the (padded) finest reference level."""
        lvl0 = self.align_state[0][-1]
        ts = int(self.config.block_matching.tuning.tile_size)
        return lvl0.shape[0] // ts, lvl0.shape[1] // ts

    def align_frame(self, img, wait_ref=None):
        """grey -> pyramid -> coarse-to-fine alignment of one comp frame: flow float32 [ny, nx, 2]."""
        cfg = self.config
        raw = self._ingest(img)
        grey = raw if self.mono else compute_grey_images(raw, self.grey_method)
        pyramid = build_gaussian_pyramid(grey, cfg.block_matching.tuning.factors)
        if wait_ref is not None:
            torch.cuda.current_stream(self.device).wait_event(self._align_ready)
        return align(*self.align_state, grey, cfg, moving_pyramid=pyramid)

    def _align_chunk(self, imgs, wait_ref=None):
        """align_frame() of a chunk of frames with ONE launch per stage (grey transform phases, pyramid levels,
        alignment levels) when the configuration allows it; per frame bit-identical to align_frame()."""
        cfg = self.config
        if len(imgs) < 2 or not self._batch or not can_align_batch(cfg):
            return [self.align_frame(img, wait_ref) for img in imgs]
        raws = [self._ingest(img) for img in imgs]
        greys = raws if self.mono else compute_grey_images_batch(raws, self.grey_method)
        pyramids = build_gaussian_pyramids(greys, cfg.block_matching.tuning.factors)
        if wait_ref is not None:
            torch.cuda.current_stream(self.device).wait_event(self._align_ready)
        return align_batch(self.align_state[0], self.align_state[5], pyramids, cfg)

    def align_frames(self, comp_imgs, n_streams=None):
        """align_frame() over a list of frames: chunks of frames round-robin on the side streams (like process_frames)."""
        with torch.cuda.device(self.device):
            comp_imgs = self.prefetch([comp_imgs[i] for i in range(len(comp_imgs))])
            chunks = self._chunks(len(comp_imgs), n_streams)
            out = self._on_streams(len(chunks), n_streams, False,
                                   lambda ci, wait: self._align_chunk([comp_imgs[i] for i in chunks[ci]], wait_ref=wait))
            return [f for chunk in out for f in chunk]

    def _front_chunk(self, imgs, wait_ref, indices, flows):
        """_front() of a chunk of frames.  With the default configuration (Bayer frames, FFT grey image, fused level
        kernels) every stage is ONE launch for the whole chunk — 3 transform phases, 3 pyramid levels, 4 alignment
        levels, 1 raw pass: 11 launches per chunk instead of per frame — and the latency-bound stages (FFT phases,
        coarse pyramid / alignment levels: a few hundred workgroups per frame) see enough work to fill the GPU.
        Results per frame are bit-identical to _front() (tests: batch == single)."""
        cfg = self.config
        inj = [None if flows is None else flows[i] for i in indices]
        if flows is None and self._inject_flows is not None:
            inj = [self._inject_flows[i] for i in indices]
        if len(imgs) >= 2 and self._batch and all(f is not None for f in inj) and cfg.robustness.enabled and not self.mono:
            # every flow is given (multi-GPU step B on a row slab): no alignment, ONE raw pass for the chunk — on a slab
            # a per-frame launch is a few hundred workgroups and leaves most of the GPU idle
            raws = [self._ingest(img) for img in imgs]
            stats = frame_stats_batch(raws, self.cfa, self.wb, cfg)
            return [(raw, _lib.f32c(f, self.device), st[2], st[0]) for raw, f, st in zip(raws, inj, stats)]
        if len(imgs) < 2 or not self._batch or not can_align_batch(cfg) or any(f is not None for f in inj):
            return [self._front(img, wait_ref, i, f) for img, i, f in zip(imgs, indices, inj)]
        raws = [self._ingest(img) for img in imgs]
        greys = raws if self.mono else compute_grey_images_batch(raws, self.grey_method)  # mono: the frame itself
        pyramids = build_gaussian_pyramids(greys, cfg.block_matching.tuning.factors)
        if wait_ref is not None:
            torch.cuda.current_stream(self.device).wait_event(self._align_ready)
        fl = align_batch(self.align_state[0], self.align_state[5], pyramids, cfg)
        if cfg.robustness.enabled:
            stats = ([frame_stats(raw, self.cfa, self.wb, cfg) for raw in raws] if self.mono  # (per-pixel covariances)
                     else frame_stats_batch(raws, self.cfa, self.wb, cfg))
            return [(raw, f, st[2], st[0]) for raw, f, st in zip(raws, fl, stats)]
        return [(raw, f, estimate_kernels(raw, cfg), None) for raw, f in zip(raws, fl)]

    def _front(self, img, wait_ref=None, index=None, flow=None):
        """grey -> pyramid -> alignment -> guide means + kernel covariances of one comp frame:
        (raw, flow, covs, guide means or None)."""
        cfg = self.config
        raw = self._ingest(img)
        if flow is None and self._inject_flows is not None and index is not None:
            flow = self._inject_flows[index]
        if flow is not None:
            flow = _lib.f32c(flow, self.device)  # (nothing here needs the reference frame: the caller waits before the robustness)
        else:
            grey = raw if self.mono else self._timed(
                compute_grey_images, 3, end_s="- grey images estimated by {}".format(self.grey_method))(raw, self.grey_method)
            pyramid = build_gaussian_pyramid(grey, cfg.block_matching.tuning.factors)
            if wait_ref is not None:
                torch.cuda.current_stream(self.device).wait_event(self._align_ready)
            flow = self._timed(align, 2, "\nBeginning alignment", "Image aligned (Total)")(
                *self.align_state, grey, cfg, moving_pyramid=pyramid)
        if cfg.robustness.enabled:  # guide means + kernel covariances from one pass over the raw frame
            means, _, covs = self._timed(frame_stats, 2, "\nEstimating kernels + guide statistics",
                                         "Kernels + guide statistics estimated (Total)")(raw, self.cfa, self.wb, cfg)
        else:
            means, covs = None, self._timed(estimate_kernels, 2, "\nEstimating kernels", "Kernels estimated (Total)")(raw, cfg)
        return raw, flow, covs, means

    def _robustness(self, fronts, accumulate_r=None, fuse_local_min=False):
        """Robustness of a group of frames of the burst (one launch per 4 frames shares the pass over the
        reference-frame planes): list of (raw, flow, covs, r)."""
        cfg = self.config
        if not cfg.robustness.enabled or len(fronts) == 1 or self.mono:
            rob = self._timed(compute_robustness, 2, "\nEstimating robustness", "Robustness estimated (Total)")
            return [(raw, flow, covs,
                     rob(raw, self.ref_means, self.ref_vars, flow, self.cfa, self.wb, self.curves, cfg,
                                        accumulate_into=accumulate_r, ref_sigma_sq=self.ref_sigma_sq, comp_means=means,
                                        fuse_local_min=fuse_local_min and cfg.robustness.enabled,
                                        flow_rows=self.flow_rows))
                    for raw, flow, covs, means in fronts]
        rs = compute_robustness_group([f[0] for f in fronts], self.ref_means, [f[1] for f in fronts], self.curves, cfg,
                                      self.ref_sigma_sq, [f[3] for f in fronts], accumulate_into=accumulate_r,
                                      fuse_local_min=fuse_local_min, flow_rows=self.flow_rows)
        return [(f[0], f[1], f[2], r) for f, r in zip(fronts, rs)]

    def process_frame(self, img, accumulate_r=None, wait_ref=None, fuse_local_min=False, index=None, flow=None):
        """grey -> kernels -> align -> robustness for one comp frame; returns (raw, flow, covs, r).
        `accumulate_r`: optional float32 [H, W] that receives += r (fused into the local-min pass).
        `wait_ref`: event after which the reference-frame state is complete — the frame's own grey image and
        pyramid do not need it and are enqueued before the wait, so on a side stream they overlap the
        (latency-bound) reference precompute.
        `flow`: a flow field that replaces the alignment (multi-GPU step B; validation hook
        config.hip.inject_flows via `index`)."""
        front = self._front(img, wait_ref, index, flow)
        if wait_ref is not None:
            torch.cuda.current_stream(self.device).wait_event(wait_ref)
        return self._robustness([front], accumulate_r, fuse_local_min)[0]

    def fuses_local_min(self):
        """True when the fused merge can take the un-filtered robustness maps (see merge.can_fuse_local_min)."""
        return bool(self.config.robustness.enabled) and can_fuse_local_min(self.config, tuple(self.ref.shape))

    def process_frames(self, comp_imgs, accumulate_r=None, n_streams=None, fuse_local_min=False, flows=None):
        """process_frame() over a list of frames.  Frames are independent until the merge, so they are
        issued round-robin on `n_streams` HIP streams (config.hip.streams, default 2; 1 for frames >= LARGE_FRAME): the launch-latency-
        bound coarse pyramid levels of one frame overlap the bandwidth-bound kernels of another.  The caller's
        stream waits for all of them before returning.  A per-frame `accumulate_r` (read-modify-write of one
        map) forces a single stream.  `flows`: per-frame flow fields that replace the alignment."""
        with torch.cuda.device(self.device):
            n = len(comp_imgs)
            comp_imgs = self.prefetch(comp_imgs)
            chunks = self._chunks(n, n_streams)

            def work(ci, wait):
                fronts = self._front_chunk([comp_imgs[i] for i in chunks[ci]], wait, chunks[ci], flows)
                if wait is not None:  # the robustness needs the second half of the reference precompute
                    torch.cuda.current_stream(self.device).wait_event(wait)
                return self._robustness(fronts, accumulate_r if wait is None else None, fuse_local_min)

            out = self._on_streams(len(chunks), n_streams, accumulate_r is not None, work)
            return [f for chunk in out for f in chunk]

    def _chunks(self, n, n_streams=None):
        """Frame indices per chunk.  Chunks of up to `config.hip.chunk` (default ROB_GROUP) frames stay together on a
        stream: their robustness is one launch per ROB_GROUP frames that reads the reference-frame planes once
        (hhsr_rob_frames), and with the batched front end every stage is one launch per chunk; chunk sizes are balanced
        over the streams."""
        streams = self._n_streams(n_streams)
        size = self._chunk_size()
        n_chunks = min(n, streams * -(-n // (streams * size))) if n else 0
        chunks, i = [], 0
        for c in range(n_chunks):
            k = n // n_chunks + (1 if c < n % n_chunks else 0)
            chunks.append(list(range(i, i + k)))
            i += k
        return chunks

    def _chunk_size(self):
        hip = self.config.get("hip", None) if hasattr(self.config, "get") else None
        return max(1, min(int(hip.get("chunk", ROB_GROUP)) if hip is not None else ROB_GROUP, _lib.MAX_BATCH))

    def _n_streams(self, n_streams):
        if n_streams is None:
            hip = self.config.get("hip", None) if hasattr(self.config, "get") else None
            default = DEFAULT_STREAMS
            ref = getattr(self, "ref", None)
            if ref is not None and ref.shape[0] * ref.shape[1] >= LARGE_FRAME:
                default = 1
            n_streams = int(hip.get("streams", default)) if hip is not None else default
        return max(1, int(n_streams))

    def _on_streams(self, n, n_streams, serial, work):
        """work(i, wait_event) for i < n; on one stream (wait_event None) or round-robin on the side streams."""
        n_streams = self._n_streams(n_streams)
        # Round 5: the side streams fork at the START of the reference precompute (self._entry), not behind it: a frame's own
        # grey image and pyramid do not need the reference frame, the alignment waits for _align_ready, the robustness
        # for `entry` below (the whole precompute and whatever else the caller's stream holds).  Until now the fork event
        # was recorded here, i.e. behind the ~0.5 ms of single-frame, latency-bound reference kernels — 6 % of the 12 MP
        # step, a sixth of a rank's step on 8 GPUs — and the _align_ready / _ref_ready waits had nothing left to wait for.
        # CONTRACT of the early fork: the side streams are ordered behind the caller's stream as of the START of
        # init_ref only.  Whatever the caller enqueues on its stream between init_ref() and this call — producing or
        # normalising comp frames on the device, a flow field that needs a copy — is NOT ordered before the side streams'
        # first reads; the in-repo callers hand over inputs that predate init_ref (or staged / host frames) and contiguous
        # float32 flow views.  HHSR_LATE_FORK=1 restores the fork at this call for callers that cannot promise that.
        early = bool(self.ref_wait and getattr(self, "_entry_fresh", False) and n_streams > 1 and not serial and n >= 1
                     and not _LATE_FORK)
        self._entry_fresh = False
        if not early and (n_streams <= 1 or serial or n < 2):
            return [work(i, None) for i in range(n)]
        main = torch.cuda.current_stream(self.device)
        if len(self._streams) < n_streams:
            self._streams += [torch.cuda.Stream(self.device) for _ in range(n_streams - len(self._streams))]
        pool = self._streams[:n_streams]
        # everything the caller enqueued so far (frame uploads / normalisation, a previous call's merge that still
        # reads buffers the allocator may hand out again) is ordered before the side streams' work
        entry = torch.cuda.Event()
        entry.record(main)
        for s in pool:
            s.wait_event(self._entry if early else entry)
        results = []
        for i in range(n):
            s = pool[i % n_streams]
            with torch.cuda.stream(s):
                f = work(i, entry if early else (self._ref_ready if self.ref_wait else None))
            for t in _tensors(f):
                t.record_stream(main)  # consumed by the merge on the caller's stream (raw too: it is allocated on
                # the side stream when the frame was uploaded / converted there)
            results.append(f)
        for s in pool:
            main.wait_stream(s)
        return results

    def output_size(self):
        s = self.config.scale
        H, W = self.ref.shape
        return round(s * H), round(s * W)


_main_runners = []  # [(config, ConfigWatch, HostBurstRunner)], most recently used last
_main_runners_lock = __import__("threading").Lock()  # main() may be called from several threads


def _drop_host_runners():
    """Interpreter exit: release the cached runners (HIP graphs, page-locked staging, copy threads) while the HIP runtime
    is still up instead of in whatever order module teardown picks."""
    while _main_runners:
        _, _, runner = _main_runners.pop()
        runner.close()


import atexit as _atexit  # noqa: E402

_atexit.register(_drop_host_runners)


def _host_runner(config, device):
    """The HostBurstRunner of a configuration object that main() is called with again and again (a serving loop); a
    configuration edited in place gets a new one.  At most two are kept (each holds a burst's intermediates)."""
    from .graph import ConfigWatch, HostBurstRunner

    with _main_runners_lock:
        for k, (cfg, watch, runner) in enumerate(_main_runners):
            if cfg is config and runner.device == device:
                _main_runners.append(_main_runners.pop(k))
                if watch.changed(config):
                    _main_runners.pop()
                    break
                return runner
        watch = ConfigWatch()
        watch.changed(config)
        runner = HostBurstRunner(config, device)
        _main_runners.append((config, watch, runner))
        while len(_main_runners) > 2:
            _main_runners.pop(0)[2].close()
        return runner


def main(ref_img, comp_imgs, config, *, _no_runner=False):
    """Alg. 1 (reference super_resolution.py:41-200).

    ref_img [H, W], comp_imgs [N-1, H, W]: float32 NumPy arrays (as in the reference) or torch tensors
    (host or already device-resident).  Returns (num, debug_dict): num is the normalised RGB image,
    a float32 GPU tensor [sH, sW, 3] WITHOUT post-processing; debug_dict has 'flow' / 'robustness' lists
    (NumPy, only if config.debug) and 'accumulated robustness' (GPU tensor) when the mask is requested.

    Host-resident bursts (the reference's signature) that come back with the same configuration object, shapes and
    dtype — a serving loop — run through graph.HostBurstRunner from the third call on: uploads as eager copies, the
    kernels as per-chunk HIP graphs, bit-identical results; main() then returns when the host buffers may be refilled
    (all uploads done) and hands out fresh tensors like the reference (one device copy of the result).
    `config.hip.graph: false` keeps the eager path."""
    if not _no_runner and torch.cuda.is_available():
        from .graph import HostBurstRunner

        denoiser_enabled(config)  # derives den.enabled in place NOW: the runner's ConfigWatch snapshot then already has it
        if HostBurstRunner.usable(config, ref_img, comp_imgs):
            # (the runner keeps a burst's staging and intermediates on the device — ~6 GB at 12 MP x 20, plus 1.6 GB for the
            # chained merge of float bursts — until the configuration object is dropped or edited; config.hip.graph: false
            # opts out.)  The result is cloned inside the runner's lock: concurrent callers never see each other's pixels
            return _host_runner(config, _device()).call_cloned(ref_img, comp_imgs)
    verbose = config.verbose >= 1
    debug_mode = bool(config.debug)
    debug_dict = {"robustness": [], "flow": []}
    denoiser_on = denoiser_enabled(config)
    accumulate_r = denoiser_on or bool(config.robustness.save_mask)
    hip_cfg = config.get("hip", None) if hasattr(config, "get") else None
    fused = True if hip_cfg is None else bool(hip_cfg.get("fused_merge", True))
    # Denoiser on (round 6): the comp frames still go through the batched front end and ONE fused merge launch — without the
    # reference frame and without normalising —, the float64 robustness sum is one pass over the frames' maps
    # (hhsr_rob_sum), then the reference's sequential tail: merge_ref with the decision map (its overwrite / widen rules,
    # merge.py:223-228) and divide.  Until now this configuration ran the per-frame operator path: a read-modify-write of
    # the accumulators and of a float64 torch tensor per frame.
    den_fused = fused and denoiser_on and can_fuse_acc_r(config) and n_images_of(comp_imgs) > 0 \
        and not (config.verbose >= 1 or bool(config.debug))
    fused = fused and not denoiser_on

    if verbose:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("\nProcessing reference image ---------\n")
    pipe = BurstPipeline(config).init_ref(ref_img)
    dev = pipe.device
    H, W = pipe.ref.shape
    sH, sW = pipe.output_size()
    # the denoiser DECIDES on the accumulated robustness: float64 sum like the reference's (robustness.RobustnessSum); a sum
    # that is only reported (save_mask) stays the float32 sum of the kernels
    acc_sum = RobustnessSum((H, W), dev) if denoiser_on else None
    accumulated_r = torch.zeros((H, W), dtype=torch.float32, device=dev) if (accumulate_r and not denoiser_on) else None
    fuse_acc = accumulate_r and fused and can_fuse_acc_r(config) and n_images_of(comp_imgs) > 0
    num = torch.empty((sH, sW, 3), dtype=torch.float32, device=dev)
    den = None
    if den_fused:
        den = torch.empty_like(num)
    elif not fused:
        num.zero_()
        den = torch.zeros_like(num)
    if verbose:
        torch.cuda.synchronize()
        getTime(t1, "\nRef Img processed (Total)")

    frames = []
    n_images = len(comp_imgs)
    fuse_min = False
    if fused and not verbose and not debug_mode:
        # the x2 merge kernel takes the 5x5 local minimum of the robustness itself (one pass and 8 B/pixel less per
        # frame); a separately accumulated robustness map needs the filtered maps
        fuse_min = pipe.fuses_local_min() and (fuse_acc or not accumulate_r)
        # (page-locked host frames: all uploads are queued now, back to back — BurstPipeline.prefetch)
        frames = pipe.process_frames([comp_imgs[i] for i in range(n_images)], None if fuse_acc else accumulated_r,
                                     fuse_local_min=fuse_min)
        n_images = 0  # the per-frame loop below is the verbose / debug / sequential-merge path
    elif den_fused:
        # the merge kernel takes the 5x5 minimum itself where it can (x2 / x3 kernels); the float64 sum then takes it on
        # the way in (HHSR_ROB_SUM_MIN5): the filtered maps are never written
        fuse_min = pipe.fuses_local_min()
        frames = pipe.process_frames([comp_imgs[i] for i in range(n_images)], None, fuse_local_min=fuse_min)
        acc_sum.add_many([f[3] for f in frames], unfiltered=fuse_min)
        n_images = 0
    for im_id in range(n_images):
        if verbose:
            torch.cuda.synchronize()
            print("\nProcessing image {} ---------\n".format(im_id + 1))
            im_time = time.perf_counter()
        raw, flow, covs, r = pipe.process_frame(comp_imgs[im_id], None if fuse_acc else accumulated_r, index=im_id)
        if acc_sum is not None:
            acc_sum.add(r)
        if fused:
            frames.append((raw, flow, covs, r))
        else:
            pipe._timed(merge, 2, "\nAccumulating Image", "Image accumulated (Total)")(raw, flow, covs, r, num, den, pipe.cfa,
                                                                                          config)
        if debug_mode:  # the reference crashes here at HEAD (Tensor.copy_to_host)
            debug_dict["flow"].append(flow.cpu().numpy())
            debug_dict["robustness"].append(r.cpu().numpy())
        if verbose:
            torch.cuda.synchronize()
            getTime(im_time, "\nImage processed (Total)")

    ref_covs = pipe.ref_covs
    if fused:
        pipe._timed(merge_burst, 2, "\nAccumulating all frames + ref Img + normalising", "Burst merged (Total)")(
            frames, pipe.ref, ref_covs, num, None, pipe.cfa, config, do_ref=True, divide=True,
            acc_r=accumulated_r if fuse_acc else None, local_min=fuse_min)
    else:
        if den_fused:
            merge_burst(frames, None, None, num, den, pipe.cfa, config, do_ref=False, divide=False, store_den=True,
                        local_min=fuse_min)
        dec = acc_sum.for_decisions(config.accumulated_robustness_denoiser.merge.max_frame_count) if denoiser_on else None
        if den_fused:  # reference frame + normalisation in ONE pass over the accumulators (HHSR_REF_DIVIDE), float32 weights
            # where the denoiser does not widen (HHSR_REF_FAST: the arithmetic of the fused merge's reference frame)
            merge_ref(pipe.ref, ref_covs, num, den, pipe.cfa, config, dec, divide=True, fast=True)
        else:
            pipe._timed(merge_ref, 2, "\nAccumulating ref Img", "Ref Img accumulated (Total)")(
                pipe.ref, ref_covs, num, den, pipe.cfa, config, dec)
            pipe._timed(divide, 2, end_s="\n------------------------\nImage normalized (Total)")(num, den)
    if verbose:
        torch.cuda.synchronize()
        s = "\nTotal ellapsed time : "
        print(s, " " * (50 - len(s)), ": ", round((time.perf_counter() - t1), 2), "seconds")
    if accumulate_r:
        debug_dict["accumulated robustness"] = acc_sum.mask() if denoiser_on else accumulated_r
    # host-buffer contract: main() returns when the caller's (page-locked) frames have been read — the uploads were
    # queued asynchronously (BurstPipeline.prefetch) and a serving loop refills its staging buffers right after the
    # call; the kernels keep running asynchronously on the current stream as before
    pipe.uploads_done()
    return num, debug_dict


def prepare_config(config, ref_raw, alpha=None, beta=None, cfa_pattern=None, white_balance=None, iso=100,
                   std_curve=None, diff_curve=None, shape=None):
    """The parameter derivation process() performs between loading the burst and calling main()
    (reference super_resolution.py:227-296): noise model, SNR -> tile size / merge tunings, sanity
    checks, exif block, denoiser switch.  Mutates `config` in place like the reference."""
    from .synthetic import noise_curves

    if config.noise_model.get("alpha", None) is not None:
        alpha, beta = config.noise_model.alpha, config.noise_model.beta
    if alpha is None or beta is None:
        raise ValueError("noise model (alpha, beta) missing: give it in config.noise_model or in the burst")
    config.noise_model.update({"alpha": float(alpha), "beta": float(beta)})
    if std_curve is None or diff_curve is None:
        # the reference draws these by an unseeded Monte-Carlo (fast_monte_carlo.py); here: its analytic
        # un-clipped limit (synthetic.noise_curves) unless the caller supplies curves
        std_curve, diff_curve = noise_curves(float(alpha), float(beta))
    brightness = float(ref_raw.mean()) if torch.is_tensor(ref_raw) else float(np.mean(ref_raw))
    id_noise = min(max(round(1000 * brightness), 0), len(std_curve) - 1)
    SNR = brightness / float(std_curve[id_noise])
    if config.verbose >= 1:
        print(" ", 10 * "-")
        print("|ISO : {}".format(iso))
        print("|Image brightness : {:.2f}".format(brightness))
        print("|expected noise std : {:.2e}".format(float(std_curve[id_noise])))
        print("|Estimated SNR : {:.2f}".format(SNR))
    update_snr_config(config, SNR)
    sanitize_config(config, tuple(shape if shape is not None else ref_raw.shape))
    config.exif = {"cfa_pattern": np.asarray(cfa_pattern).tolist(), "iso": iso,
                   "white_balance": [float(v) for v in white_balance]}
    config.noise_model.update({"std_curve": np.asarray(std_curve).tolist(), "diff_curve": np.asarray(diff_curve).tolist()})
    den = config.accumulated_robustness_denoiser
    den.enabled = bool(den.median.enabled or den.gauss.enabled or den.merge.enabled)
    return config


def process(burst_path, config):
    """process(burst_path, config) -> (float32 ndarray [sH, sW, 3], debug_dict)   (reference :203-360).

    `burst_path` is either a burst held in memory / in an .npz file — a mapping with keys ``ref`` [H,W],
    ``comp`` [N-1,H,W], ``cfa_pattern`` [2,2], ``white_balance`` [>=3], ``alpha``, ``beta`` and optionally
    ``iso``, ``std_curve``, ``diff_curve``, ``orientation`` (EXIF 1..8), ``xyz2cam`` (3x3, DNG ColorMatrix1, for
    ``postprocessing.do_color_correction``) — ``cfa_pattern`` / ``white_balance`` may be left out with ``mode: grey``
    (monochrome sensors: channel 0 of the result is the image, channels 1 and 2 are NaN like the reference's); ref / comp are either normalised white-balanced float RAW or integer
    sensor counts with ``black_levels`` and ``white_level`` (normalised on the GPU like utils_dng.py:149-160) —
    or a folder of .dng files, which needs rawpy + exifread like the reference (absent from this image:
    ImportError).  Noise curves: given in the burst, or ``config.noise_model.estimator``: "monte_carlo" (default — the
    reference's estimator run_fast_MC, super_resolution.py:252, whose curves include the clipping of the noisy samples
    to [0, 1]: near black and near saturation sigma_t and d_t are up to ~1.6x smaller than the un-clipped law; seeded by
    ``config.noise_model.seed``, default 0, so process() is reproducible where the reference is not, D18) or
    "analytic" (the un-clipped limit, synthetic.noise_curves — what prepare_config() uses when called directly).
    After main(): the frame-count denoisers (``accumulated_robustness_denoiser.median / .gauss``), then
    ``postprocessing`` (colour matrix, unsharp mask, devignetting, gamma — ON by default like the reference's YAML; tone
    mapping raises NotImplementedError) and the EXIF orientation, all on the GPU (utils_image.py, raw2rgb.py); only the
    finished image is copied to the host."""
    import os

    from .utils_dng import load_dng_burst, normalize_burst

    if isinstance(burst_path, (str, os.PathLike)) and str(burst_path).endswith(".npz"):
        burst = dict(np.load(burst_path))
    elif isinstance(burst_path, (str, os.PathLike)):
        ref_raw, raw_comp, iso, tags, cfa, _, white_balance, _ = load_dng_burst(burst_path)  # needs rawpy + exifread
        nm = tags["Image Tag 0xC761"].values  # DNG NoiseProfile, already scaled for the ISO (reference :232-238)
        if config.mode == "grey":  # one (alpha, beta) pair (reference :233-235)
            alpha_beta = {"alpha": nm[0][0], "beta": nm[1][0]}
        else:                      # mean over the three colour planes (reference :236-238)
            alpha_beta = {"alpha": sum(x[0] for x in nm[::2]) / 3, "beta": sum(x[0] for x in nm[1::2]) / 3}
        burst = {"ref": ref_raw, "comp": raw_comp, "iso": iso, "cfa_pattern": cfa, "white_balance": white_balance,
                 **alpha_beta}
        if "Image Orientation" in tags:
            burst["orientation"] = tags["Image Orientation"].values[0]
        if "Image Tag 0xC621" in tags:  # DNG ColorMatrix1 (raw2rgb.py:12-27)
            burst["xyz2cam"] = np.array([x.decimal() for x in tags["Image Tag 0xC621"].values], np.float32).reshape(3, 3)
    else:
        burst = burst_path
    if config.mode == "grey":  # monochrome sensor: no colour filter array, no white balance (both unused by main())
        burst = dict(burst)
        burst.setdefault("cfa_pattern", [[0, 1], [1, 2]])
        burst.setdefault("white_balance", [1.0, 1.0, 1.0])
    ref_raw, raw_comp = burst["ref"], burst["comp"]
    if not torch.is_tensor(ref_raw) and np.issubdtype(np.asarray(ref_raw).dtype, np.integer):
        # sensor counts + metadata: the normalisation / white balance of utils_dng.py:149-160, on the GPU.  The
        # reference frame now (its brightness picks the SNR-based parameters); the other frames stay integer host arrays
        # and are uploaded as counts — half the bytes of float32 frames — and normalised frame by frame on the pipeline
        # streams (BurstPipeline._ingest), overlapping the other frames' kernels
        ref_raw = normalize_burst(np.asarray(ref_raw), burst["black_levels"], burst["white_level"],
                                  burst["white_balance"], burst["cfa_pattern"])
        raw_comp = [np.asarray(f) for f in raw_comp]
        hip = dict(config.get("hip", None) or {})
        hip["raw_norm"] = {"black_levels": [float(v) for v in list(burst["black_levels"])[:3]],
                           "white_level": float(burst["white_level"])}
        config.hip = hip
        brightness_src = ref_raw.mean().item()
    else:
        ref_raw = np.asarray(ref_raw, dtype=np.float32) if not torch.is_tensor(ref_raw) else ref_raw
        raw_comp = np.asarray(raw_comp, dtype=np.float32) if not torch.is_tensor(raw_comp) else raw_comp
        brightness_src = None
    std_curve, diff_curve = burst.get("std_curve"), burst.get("diff_curve")
    alpha, beta = burst.get("alpha"), burst.get("beta")
    if (std_curve is None or diff_curve is None) and config.noise_model.get("estimator", "monte_carlo") == "monte_carlo":
        from .fast_monte_carlo import run_fast_MC  # the reference's estimator (super_resolution.py:252), seeded

        if config.noise_model.get("alpha", None) is not None:
            alpha, beta = config.noise_model.alpha, config.noise_model.beta
        std_curve, diff_curve = run_fast_MC(float(alpha), float(beta), seed=int(config.noise_model.get("seed", 0)))
    ref_for_stats = ref_raw if brightness_src is None else np.full((1, 1), brightness_src, np.float32)
    prepare_config(config, ref_for_stats, alpha, beta, burst["cfa_pattern"], burst["white_balance"],
                   int(burst.get("iso", 100)), std_curve, diff_curve,
                   shape=tuple(ref_raw.shape))
    out, debug_dict = main(ref_raw, raw_comp, config)

    # ---- after the path (reference super_resolution.py:303-356), on the device --------------------------------------
    from . import raw2rgb
    from .utils_image import apply_orientation, frame_count_denoising_gauss, frame_count_denoising_median

    den = config.accumulated_robustness_denoiser
    compat = config.get("compat", None) or {}
    half_index = bool(compat.get("post_denoiser_half_index", True))
    if den.median.enabled:
        out = frame_count_denoising_median(out, debug_dict["accumulated robustness"], den.median, scale=config.scale,
                                           half_index=half_index, mode=config.mode)
    if den.gauss.enabled:
        out = frame_count_denoising_gauss(out, debug_dict["accumulated robustness"], den.gauss, scale=config.scale,
                                          half_index=half_index, mode=config.mode)
    ori = int(burst.get("orientation", 1))  # EXIF 'Image Orientation' (reference :346-352); 1 when the burst has none
    pp = config.postprocessing
    if pp.enabled:
        out = raw2rgb.postprocess(burst.get("rawpy_image", None), out, pp.do_color_correction, pp.do_tonemapping,
                                  pp.do_gamma_correction, pp.sharpening, pp.do_devignetting, burst.get("xyz2cam", None),
                                  orientation=ori)
    else:
        out = apply_orientation(out, ori)
    output_image = out.cpu().numpy()
    if "accumulated robustness" in debug_dict:
        debug_dict["accumulated robustness"] = apply_orientation(debug_dict["accumulated robustness"], ori).cpu().numpy()
    return output_image, debug_dict
