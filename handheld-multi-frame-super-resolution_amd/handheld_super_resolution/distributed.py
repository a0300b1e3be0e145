"""Multi-GPU burst merge.  The reference is single-GPU; this is the one parallelism the build adds.

One process per GPU, two selectable strategies (`main_sharded(..., strategy=)`, `config.hip.strategy`, `bench.py --strategy`):

"rows" (default).  The path has two kinds of work and they shard differently:
  A. alignment (FFT grey image, pyramid, block matching, ICA) needs a WHOLE frame -> frame-parallel:
     rank k aligns comp frames k, k+G, k+2G, ... (reference super_resolution.py:133-151 per frame);
  B. kernel estimation, robustness and the merge are local in the image plane (a few pixels of halo plus the flow)
     -> row-parallel: the output is cut into G row slabs and rank j runs steps B for ALL frames on the raw rows its
     slab needs (slab + |flow| + HALO rows), finishing its slab completely: reference frame, normalisation.
  The only data-path exchange is ONE all-gather of the flow fields (ny x nx x 2 floats per frame: 376 kB at 12 MP, 7 MB
  per 20-frame burst).  Host-resident frames: a rank uploads its own N/G frames whole (step A) and rows [S0, S1) of every
  frame (step B): (N/G + N (1/G + 2 halo/H)) frames' worth of bytes instead of N.

"reduce" (the north star's formulation).  Frames shard one per GPU round-robin; every rank runs the WHOLE single-GPU
  chain on its frames and merges them into float32 accumulators num / den [sH, sW, 3] (reference merge.py:432-434);
  ONE reduce-scatter (sum) over row slabs — issued as two RCCL calls, num and den — leaves rank j with the summed
  accumulators of slab j, to which it adds the reference frame and normalises.  1.15 GB of accumulators per rank at
  12 MP x2 cross xGMI (per-link bound for a ring: modelled 8-10 ms against 1.1 ms of compute per rank at G = 8), so
  this strategy only pays for very long bursts; it exists so that both can be measured.  Results differ from the
  single-GPU run by float32 summation order (partial sums are added), ~1e-7 relative.

The output stays sharded (`gather=False`: rank j owns rows slab_bounds[j] .. slab_bounds[j+1]) or is gathered to rank 0
(`gather=True`, the reference's single-image result; 72 MB per rank at 12 MP x2 over 7 parallel links).

"rows", step B works on SUB-IMAGES: row ranges [S0, S1) of every raw frame, S0 a multiple of the flow tile size (so the
tile grid, the Bayer phase and — for integer S0 * scale — the output grid of the sub-image coincide with the full
image's).  The kernels are the single-GPU ones; per-pixel results inside the slab are bit-identical to the single-GPU
run (every scale: the merge evaluates positions in full-frame coordinates, `lr_row_offset`) because every pixel the
slab's outputs depend on lies at least 8 rows inside the sub-image (the first rows of a sub-image see an artificial
image border: D6's r = 0 rows, clamped neighbourhoods — they are halo) — and because the one quantity that is NOT
row-local, the flow-irregularity weight S of a tile (spread of the flow over its 3 x 3 TILE neighbourhood,
robustness.py:570-612), is evaluated on the full gathered flow field: the sub-image pipelines get row slices (views) of
the full fields together with the number of tile rows around them (`flow_rows`, hhsr_rob_frames / hhsr_rob_s), so the
first and last tile rows of a sub-image see their real neighbours (round 2 recomputed S from the slice: ghost-rejection
strength changed at slab seams exactly where objects move — tests/test_distributed_gloo.py::test_moving_object_at_seam).

The engine that does the per-rank compute is injected so that the sharding / exchange logic can be exercised without
a GPU (tests run it on gloo with the NumPy oracle as the engine, world_size 2 and 3).
"""
import math

import torch
import torch.distributed as dist

SLAB_ALIGN = 96  # slabs start on the workgroup grids of all merge kernels: 32 output rows (x2), 48 (x3), 16 (tile kernel)
HALO = 12        # rows of context beyond slab + |flow|.  r[y] = min of R over y +- 2 (robustness.py:641-686); R[y'] reads
                 # the guide statistics at (y' + flow) / 2 with Dodgson taps +- 1.5 guide pixels on 3 x 3 local means
                 # (robustness.py:207-294, 359-421): raw rows y' +- 7 (+ flow) -> r depends on rows y +- 9; kernel
                 # estimation and the merge window need less.  + 2 rows for the clamped neighbourhoods at a sub-image's
                 # artificial border, + 1 spare.


def shard_indices(n_frames, rank, world):
    """Indices of the comp frames rank `rank` of `world` processes aligns (round-robin: balanced to +-1)."""
    return list(range(rank, n_frames, world))


def slab_rows(sH, world):
    """Rows per slab: ceil(sH / world) rounded up to SLAB_ALIGN.  All slabs have this (padded) size so that
    every collective moves equal chunks; slab j covers output rows [j*rows, min((j+1)*rows, sH))."""
    rows = -(-sH // world)
    return -(-rows // SLAB_ALIGN) * SLAB_ALIGN


def slab_bounds(sH, world):
    """Valid (un-padded) row range [b[j], b[j+1]) of every slab."""
    rows = slab_rows(sH, world)
    return [min(j * rows, sH) for j in range(world + 1)]


def sub_image_rows(r0, r1, scale, H, ts, max_flow_y, from_top=False):
    """Raw rows [S0, S1) that output rows [r0, r1) depend on, for flows of at most `max_flow_y` pixels.
    S0 is a multiple of the tile size with S0 * scale an integer (else 0: the sub-image starts at the top);
    S1 is even (Bayer quads).  Returns (S0, S1, row0_sub) with row0_sub = r0 - S0 * scale, the slab's first row
    inside the sub-image's output."""
    halo = int(math.ceil(max_flow_y)) + HALO
    lo = int(math.floor(r0 / scale)) - halo
    hi = int(math.ceil(r1 / scale)) + halo
    S0 = 0 if from_top else max(0, (lo // ts) * ts)
    while S0 > 0 and abs(S0 * scale - round(S0 * scale)) > 1e-9:
        S0 -= ts
    S1 = min(H, hi + (hi & 1))
    return S0, S1, r0 - int(round(S0 * scale))


class HipEngine:
    """Per-rank compute on the local MI355X."""

    def __init__(self, config):
        from .super_resolution import denoiser_enabled

        self.config = config
        self.denoiser_on = denoiser_enabled(config)
        self.accumulate_r = self.denoiser_on or bool(config.robustness.save_mask)
        self.pipe = None
        self._runner = None  # HIP-graph replay of main() for device-resident bursts (graph.py)
        self._runner_a, self._runners_b, self._flows_static = None, {}, None  # ... and of the two multi-GPU steps
        self._runner_p, self._runners_f = None, {}  # strategy "reduce": partial merge of the rank's frames / slab finish
        self._host = None  # graph.HostBurstRunner: host-resident bursts on one GPU
        self._buffers = {}  # static exchange buffers (a graph is bound to the addresses of its inputs)

    def single(self, ref_img, comp_imgs):
        """world = 1: the single-GPU path itself.  An engine that is kept across bursts replays main() from a HIP graph
        when the same device tensors come back (config.hip.graph, default on; the first call runs eagerly, the second
        captures): the returned tensors then belong to the graph and are overwritten by the next call."""
        from .super_resolution import main
        from .graph import GraphRunner, capturable

        from .graph import HostBurstRunner

        if HostBurstRunner.usable(self.config, ref_img, comp_imgs):
            # host-resident burst: eager uploads + per-chunk HIP graphs over static staging buffers (graph.py); the
            # returned tensors belong to the runner like the graph path's below
            self._check_config()
            if self._host is None:
                from .super_resolution import _device

                self._host = HostBurstRunner(self.config, _device())
            return self._host(ref_img, comp_imgs)
        packed = torch.is_tensor(comp_imgs)
        tensors = (ref_img, comp_imgs) if packed else (ref_img, *comp_imgs)
        if not capturable(self.config, tensors):
            return main(ref_img, comp_imgs, self.config, _no_runner=True)
        self._check_config()
        if self._runner is None:
            cfg = self.config
            self._runner = GraphRunner(lambda ref, *comp: main(ref, comp[0] if packed else list(comp), cfg), ref_img.device)
            self._packed = packed
        if self._packed != packed:
            return main(ref_img, comp_imgs, self.config, _no_runner=True)
        return self._runner(*tensors)

    def _check_config(self):
        """A graph holds what the captured Python code decided from the configuration: when the configuration was edited
        in place since the graphs were captured, drop them."""
        from .graph import ConfigWatch

        if getattr(self, "_watch", None) is None:
            self._watch = ConfigWatch()
        if self._watch.changed(self.config):
            self._runner, self._runner_a, self._runners_b, self._flows_static = None, None, {}, None
            self._runner_p, self._runners_f = None, {}
            self._host = None

    def init_ref(self, ref_img):
        """Replicated on every rank: the reference frame's alignment state (step A needs the whole frame)."""
        from .super_resolution import BurstPipeline

        self.pipe = BurstPipeline(self.config)
        self.device = self.pipe.device
        self.pipe.init_ref(ref_img, robustness=False)
        return self

    def step_a(self, ref_img, my_frames):
        """init_ref + align_frames; replayed from a HIP graph when an engine sees the same device tensors again.
        Returns (flows [n, ny, nx, 2], the device-resident reference frame)."""
        from .graph import GraphRunner, capturable

        def fn(ref, *frames):
            self.init_ref(ref)
            return self.align_frames(list(frames)), self.pipe.ref

        tensors = (ref_img, *my_frames)
        if not capturable(self.config, tensors):
            return fn(*tensors)
        self._check_config()
        if self._runner_a is None:
            self._runner_a = GraphRunner(fn, ref_img.device)
        out = self._runner_a(*tensors)
        self.device = ref_img.device
        return out

    def shape(self):
        return tuple(self.pipe.ref.shape)

    def output_shape(self):
        return (*self.pipe.output_size(), 3)

    def tile_size(self):
        return int(self.config.block_matching.tuning.tile_size)

    def align_frames(self, comp_imgs):
        """Step A for this rank's frames: float32 [n, ny, nx, 2] on the device (n may be 0)."""
        flows = self.pipe.align_frames(list(comp_imgs))
        ny, nx = self.pipe.flow_grid()
        if not flows:
            return torch.empty((0, ny, nx, 2), dtype=torch.float32, device=self.device)
        return torch.stack(flows)

    def merge_rows(self, comp_imgs, flows, r0, r1, max_flow_y, ref_dev=None):
        """Step B: output rows [r0, r1) from ALL frames (flows: [N-1, ny, nx, 2]).  Returns (slab float32
        [r1 - r0, sW, 3], accumulated robustness of the raw rows [ceil(r0 / scale), ceil(r1 / scale)) — the rows whose
        first output row lies in the slab: a disjoint cover over the slabs — or None).  With the device-resident
        reference frame of step_a (`ref_dev`) and device-resident frames the step is replayed from a HIP graph: the
        gathered flows are copied into a static buffer, the flow bound is rounded up to a multiple of 8 rows (the result
        does not depend on the halo, only the sub-image's extent does) and one graph is kept per (r0, r1, bound)."""
        from .graph import GraphRunner, capturable

        packed = torch.is_tensor(comp_imgs)
        tensors = (comp_imgs,) if packed else tuple(comp_imgs)
        if ref_dev is None or flows is None or not capturable(self.config, (ref_dev, flows, *tensors)):
            return self._merge_rows(comp_imgs, flows, r0, r1, max_flow_y, self.pipe.ref if ref_dev is None else ref_dev)
        bound = float(math.ceil(max_flow_y / 8.0) * 8)
        if self._flows_static is None or self._flows_static.shape != flows.shape:
            self._flows_static = torch.empty_like(flows)
            self._runners_b = {}
        self._flows_static.copy_(flows)
        key = (int(r0), int(r1), bound, packed)
        runner = self._runners_b.get(key)
        if runner is None:
            if len(self._runners_b) >= 4:
                self._runners_b.pop(next(iter(self._runners_b)))
            runner = self._runners_b[key] = GraphRunner(
                lambda ref, fl, *comp: self._merge_rows(comp[0] if packed else list(comp), fl, r0, r1, bound, ref),
                ref_dev.device)
        return runner(ref_dev, self._flows_static, *tensors)

    def _merge_rows(self, comp_imgs, flows, r0, r1, max_flow_y, ref_dev):
        from .super_resolution import BurstPipeline
        from .merge import merge, merge_ref, merge_burst, can_fuse_acc_r
        from .utils import divide

        cfg = self.config
        H, W = self.shape()
        sH, sW, _ = self.output_shape()
        scale, ts = cfg.scale, self.tile_size()
        pow2 = float(scale) in (1.0, 2.0, 4.0, 8.0)
        # (the per-frame operator path of the denoiser has no row offset: for scales whose positions idx / scale are
        # not exact in float32 its sub-image starts at the top of the frame, where sub-image = full-frame coordinates)
        S0, S1, row0 = sub_image_rows(r0, r1, scale, H, ts, max_flow_y, from_top=self.denoiser_on and not pow2)
        Hs = S1 - S0
        sHs = int(round(scale * Hs))
        nrows = r1 - r0
        t0, t1 = S0 // ts, -(-S1 // ts)
        sub = BurstPipeline(cfg, self.device)
        sub.init_ref(ref_dev[S0:S1], alignment=False)  # device-resident rows of the replicated reference frame
        n = len(comp_imgs)
        # row slices = VIEWS of the full fields + the tile rows around them: the flow-irregularity weight S of the
        # sub-image's first / last tile row is evaluated on the full field (module docstring)
        flows = flows.contiguous()
        sub_flows = [flows[i, t0:t1] for i in range(n)]
        sub.flow_rows = (t0, int(flows.shape[1]) - t1)
        out = torch.empty((nrows, sW, 3), dtype=torch.float32, device=self.device)
        acc_r = torch.zeros((Hs, W), dtype=torch.float32, device=self.device) if self.accumulate_r else None
        L0 = int(math.ceil(r0 / scale)) - S0
        L1 = min(Hs, int(math.ceil(r1 / scale)) - S0)
        if self.denoiser_on:
            # the accumulated-robustness denoiser (merge.py:223-228) needs sum_n r_n before the reference frame is
            # merged: sequential operator path on the sub-image
            num = torch.zeros((sHs, sW, 3), dtype=torch.float32, device=self.device)
            den = torch.zeros_like(num)
            for i in range(n):
                raw, flow, covs, r = sub.process_frame(comp_imgs[i][S0:S1], acc_r, flow=sub_flows[i])
                merge(raw, flow, covs, r, num, den, sub.cfa, cfg)
            merge_ref(sub.ref, sub.ref_covs, num, den, sub.cfa, cfg, acc_r)
            divide(num, den)
            out.copy_(num[row0:row0 + nrows])
        else:
            fuse_acc = acc_r is not None and can_fuse_acc_r(cfg) and n > 0
            fuse_min = sub.fuses_local_min() and (fuse_acc or acc_r is None) and row0 % SLAB_ALIGN == 0
            frames = sub.process_frames([img[S0:S1] for img in comp_imgs], None if fuse_acc else acc_r,
                                        fuse_local_min=fuse_min, flows=sub_flows)
            merge_burst(frames, sub.ref, sub.ref_covs, out, None, sub.cfa, cfg, do_ref=True, divide=True,
                        acc_r=acc_r if fuse_acc else None, rows=(row0, nrows), out_height=sHs, local_min=fuse_min,
                        lr_row_offset=S0)
        return out, (acc_r[L0:L1] if acc_r is not None else None)

    # ---- strategy "reduce": frame-sharded merge + one reduce-scatter of the accumulators ---------------------------------
    def partial(self, ref_img, my_frames, padded_rows):
        """This rank's frames through the whole single-GPU chain and into raw accumulators (no reference frame, no
        normalisation): (num, den) float32 [padded_rows, sW, 3] — the first sH rows are the image, the padding rows
        (equal slabs for the reduce-scatter) are zero — and the rank's accumulated robustness [H, W] or None.
        Replayed from a HIP graph for device-resident inputs like step_a."""
        from .graph import GraphRunner, capturable

        if self.denoiser_on:
            raise NotImplementedError("strategy 'reduce' does not apply the accumulated-robustness denoiser "
                                      "(merge.py:223-228 needs the reduced robustness before the reference frame): use 'rows'")

        def fn(ref, *frames):
            from .super_resolution import BurstPipeline
            from .merge import merge_burst, can_fuse_acc_r

            cfg = self.config
            self.pipe = pipe = BurstPipeline(cfg).init_ref(ref)
            self.device = pipe.device
            H, W = pipe.ref.shape
            sH, sW = pipe.output_size()
            acc = torch.zeros((2, padded_rows, sW, 3), dtype=torch.float32, device=self.device)
            acc_r = torch.zeros((H, W), dtype=torch.float32, device=self.device) if self.accumulate_r else None
            if frames:
                fuse_acc = acc_r is not None and can_fuse_acc_r(cfg)
                fuse_min = pipe.fuses_local_min() and (fuse_acc or acc_r is None)
                fr = pipe.process_frames(list(frames), None if fuse_acc else acc_r, fuse_local_min=fuse_min)
                merge_burst(fr, None, None, acc[0, :sH], acc[1, :sH], pipe.cfa, cfg, do_ref=False, divide=False,
                            store_den=True, acc_r=acc_r if fuse_acc else None, local_min=fuse_min)
            return acc, acc_r, pipe.ref, pipe.ref_covs

        tensors = (ref_img, *my_frames)
        if not capturable(self.config, tensors):
            return fn(*tensors)
        self._check_config()
        if self._runner_p is None:
            self._runner_p = GraphRunner(fn, ref_img.device)
        out = self._runner_p(*tensors)
        self.device = ref_img.device
        return out

    def finish_rows(self, acc_slab, r0, r1, ref_dev, ref_covs):
        """Reference frame + normalisation of output rows [r0, r1) on top of the reduced accumulators
        acc_slab [2, rows, sW, 3] (merge.py:83-233, utils.py:85): slab float32 [r1 - r0, sW, 3]."""
        from .graph import GraphRunner, capturable
        from .merge import merge_burst

        cfg = self.config
        H, W = ref_dev.shape
        sH = round(cfg.scale * H)

        def fn(acc, ref, covs):
            num, den = acc[0, : r1 - r0], acc[1, : r1 - r0]
            merge_burst([], ref, covs, num, den, self.pipe.cfa, cfg, load_acc=True, do_ref=True, divide=True,
                        rows=(r0, r1 - r0), out_height=sH)
            return num

        if not capturable(cfg, (acc_slab, ref_dev, ref_covs)):
            return fn(acc_slab, ref_dev, ref_covs)
        key = (int(r0), int(r1))
        runner = self._runners_f.get(key)
        if runner is None:
            runner = self._runners_f[key] = GraphRunner(fn, ref_dev.device)
        return runner(acc_slab, ref_dev, ref_covs)


def _staged(t, group):
    """Host-only backends (gloo, used by the CPU tests) get CPU copies of device tensors."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _all_gather(t, world, group):
    """Equal-size all-gather: [world, *t.shape].  RCCL ("nccl") gathers device tensors in place over xGMI."""
    src = t.cpu() if _staged(t, group) else t.contiguous()
    out = torch.empty((world, *src.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group) if src.is_cuda else dist.all_gather(list(out.unbind(0)), src, group=group)
    return out.to(t.device)


def _gather(t, dst, world, group):
    """Equal-size gather of `t` to global rank `dst`; returns the stacked tensor there, None elsewhere."""
    me = dist.get_rank()
    staged = _staged(t, group)
    src = t.cpu() if staged else t.contiguous()
    out = torch.empty((world, *src.shape), dtype=src.dtype, device=src.device) if me == dst else None
    dist.gather(src, list(out.unbind(0)) if me == dst else None, dst=dst, group=group)  # received in place
    if me != dst:
        return None
    return out.to(t.device) if staged else out


def _reduce_scatter_rows(acc, world, group, buffers=None):
    """acc [2, world * rows, sW, 3] -> this rank's summed slab [2, rows, sW, 3].  RCCL: reduce-scatter over xGMI (two
    calls: num, den) into a buffer that is kept across bursts (`buffers`: the engine's dict — the finishing step is a HIP
    graph bound to its address); host backends (gloo, CPU tests) have no reduce-scatter: all-reduce + slice."""
    rows = acc.shape[1] // world
    rank = dist.get_rank(group)
    if acc.is_cuda and dist.get_backend(group) == "nccl":
        shape = (2, rows, *acc.shape[2:])
        out = buffers.get("rs_out") if buffers is not None else None
        if out is None or tuple(out.shape) != shape or out.device != acc.device:
            out = torch.empty(shape, dtype=acc.dtype, device=acc.device)
            if buffers is not None:
                buffers["rs_out"] = out
        for k in range(2):
            dist.reduce_scatter_tensor(out[k], acc[k], op=dist.ReduceOp.SUM, group=group)
        return out
    host = acc.cpu() if acc.is_cuda else acc.clone()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    return host[:, rank * rows:(rank + 1) * rows].contiguous().to(acc.device)


def _all_reduce(t, group):
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _strategy(config, strategy):
    if strategy is None:
        hip = config.get("hip", None) if hasattr(config, "get") else None
        strategy = hip.get("strategy", "rows") if hip is not None else "rows"
    if strategy not in ("rows", "reduce"):
        raise ValueError(f"unknown multi-GPU strategy {strategy!r} (rows | reduce)")
    return strategy


def main_sharded(ref_img, comp_imgs, config, group=None, engine=None, gather=True, max_flow=None, strategy=None):
    """Multi-GPU equivalent of main() (see the module docstring).

    gather=True : returns (output [sH, sW, 3], debug_dict) on rank 0 and (None, {}) on the other ranks;
    gather=False: every rank returns (its slab [rows_j, sW, 3], {"rows": (r0, r1), ...}) — the output stays sharded.
    `strategy`: "rows" (default; config.hip.strategy) or "reduce".
    `max_flow` (strategy "rows"; also config.hip.max_flow): bound on |flow_y| in pixels that sizes the sub-image halo.
    Given, the step runs WITHOUT any host read (debug["flow_bound_exceeded"] is a 0-dim device tensor the caller may
    inspect later: True = some flow left the halo, rows near slab seams are then not the single-GPU result); default:
    measured from the gathered flow fields — the one device-to-host read of a scalar of the step, after which the
    sub-image extent is rounded up to multiples of 8 rows so that the captured step-B graph is reused from burst to
    burst.  Works un-initialised / with world_size 1 (then it IS main())."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    eng = engine if engine is not None else HipEngine(config)
    n = len(comp_imgs)
    if world == 1:
        out, debug = eng.single(ref_img, comp_imgs)
        if not gather:
            debug = dict(debug, rows=(0, int(out.shape[0])))
        return out, debug
    if config.mode != "bayer":
        # the reference's one-channel robustness reads the statistics of row y / 2 for row y (robustness.py:358 with the
        # same-size map of :337-343): a row slab is not self-contained there, so monochrome bursts run on one GPU
        raise NotImplementedError("mode 'grey' is not sharded over GPUs (its robustness is not row-local); use main()")
    root = dist.get_global_rank(group, 0) if group is not None else 0
    strategy = _strategy(config, strategy)
    mine = shard_indices(n, rank, world)
    want_acc = bool(getattr(eng, "accumulate_r", False))
    debug = {"robustness": [], "flow": []}
    slab, acc_r, acc_full = None, None, None

    if strategy == "reduce":
        # ---- every rank: its frames through the whole chain into accumulators; ONE reduce-scatter over row slabs ------
        H, W = tuple(ref_img.shape)
        sH, sW = round(config.scale * H), round(config.scale * W)
        rows = slab_rows(sH, world)
        bounds = slab_bounds(sH, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        acc, acc_r_part, ref_dev, ref_covs = eng.partial(ref_img, [comp_imgs[i] for i in mine], world * rows)
        red = _reduce_scatter_rows(acc, world, group, getattr(eng, "_buffers", None))                   # [2, rows, sW, 3]: this rank's slab, summed
        if r1 > r0:
            slab = eng.finish_rows(red, r0, r1, ref_dev, ref_covs)
        if want_acc:
            acc_full = _all_reduce(acc_r_part, group)                       # [H, W], 48 MB at 12 MP
            a0, a1 = int(math.ceil(r0 / config.scale)), min(H, int(math.ceil(r1 / config.scale)))
            acc_r = acc_full[a0:a1] if r1 > r0 else None
        dev = acc.device
    else:
        # ---- A: frame-parallel alignment, then ONE all-gather of the flow fields -----------------------------------
        per_rank = -(-n // world) if n else 0
        flows, ref_dev = None, None
        if hasattr(eng, "step_a"):  # reference-frame state + alignment of this rank's frames (a HIP graph on replay)
            local, ref_dev = eng.step_a(ref_img, [comp_imgs[i] for i in mine])
        else:
            eng.init_ref(ref_img)
            local = eng.align_frames([comp_imgs[i] for i in mine]) if n else None  # [len(mine), ny, nx, 2]
        sH, sW, _ = eng.output_shape()
        H, W = eng.shape()
        if max_flow is None:
            hip = config.get("hip", None) if hasattr(config, "get") else None
            max_flow = hip.get("max_flow", None) if hip is not None else None
        if n:
            padded = torch.zeros((per_rank, *local.shape[1:]), dtype=local.dtype, device=local.device)
            padded[: local.shape[0]] = local
            allf = _all_gather(padded, world, group)                            # [world, per_rank, ny, nx, 2]
            # frame i was aligned by rank i % world as its (i // world)-th frame
            flows = allf.transpose(0, 1).reshape(per_rank * world, *local.shape[1:])[:n]
            if max_flow is None:
                max_flow = float(flows[..., 1].abs().max())                     # the one host read of the step
                if not math.isfinite(max_flow):
                    max_flow = float(H)
            else:  # caller's bound: no host read; the check stays on the device
                debug["flow_bound_exceeded"] = ~(flows[..., 1].abs().amax() <= float(max_flow))
        max_flow = 0.0 if max_flow is None else float(max_flow)

        # ---- B: row-parallel kernels + robustness + merge + reference frame + normalisation ------------------------
        rows = slab_rows(sH, world)
        bounds = slab_bounds(sH, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        if r1 > r0:
            slab, acc_r = (eng.merge_rows(comp_imgs, flows, r0, r1, max_flow, ref_dev=ref_dev) if ref_dev is not None
                           else eng.merge_rows(comp_imgs, flows, r0, r1, max_flow))
        dev = flows.device if flows is not None else (slab.device if slab is not None else torch.device("cpu"))
    debug["rows"] = (r0, r1)
    if not gather:
        if acc_r is not None:
            debug["accumulated robustness"] = acc_r
        return slab, debug

    # ---- optional: the finished slabs to rank 0 (equal padded chunks) -----------------------------------------------
    send = torch.zeros((rows, sW, 3), dtype=torch.float32, device=dev)
    if slab is not None:
        send[: r1 - r0] = slab
    gathered = _gather(send, root, world, group)                            # [world, rows, sW, 3] on rank 0
    acc_all = None
    if want_acc and acc_full is None:
        lrows = int(math.ceil(rows / config.scale)) + 2  # raw rows [floor(r0 / s), ceil(r1 / s)) of any slab
        a_send = torch.zeros((lrows, W), dtype=torch.float32, device=dev)
        if acc_r is not None:
            a_send[: acc_r.shape[0]] = acc_r
        acc_all = _gather(a_send, root, world, group)
    if rank != 0:
        return None, {}
    out = gathered.view(world * rows, sW, 3)[:sH]
    debug = {"robustness": [], "flow": []}
    if want_acc and acc_full is not None:
        debug["accumulated robustness"] = acc_full
    elif want_acc:
        parts = []
        for j in range(world):
            a0 = int(math.ceil(bounds[j] / config.scale))
            a1 = min(H, int(math.ceil(bounds[j + 1] / config.scale)))
            if bounds[j + 1] > bounds[j]:
                parts.append((a0, acc_all[j, : a1 - a0]))
        acc_f = torch.zeros((H, W), dtype=torch.float32, device=dev)
        for a0, p in parts:
            acc_f[a0: a0 + p.shape[0]] = p
        debug["accumulated robustness"] = acc_f
    return out, debug
