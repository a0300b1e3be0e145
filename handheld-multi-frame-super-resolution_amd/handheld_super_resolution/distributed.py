"""Multi-GPU burst merge.  The reference is single-GPU; this is the one parallelism the build adds.

One process per GPU, two selectable strategies (`main_sharded(..., strategy=)`, `config.hip.strategy`, `bench.py --strategy`):

"rows" (default).  The path has two kinds of work and they shard differently:
  A. alignment (FFT grey image, pyramid, block matching, ICA) needs a WHOLE frame -> frame-parallel:
     rank k aligns comp frames k, k+G, k+2G, ... (reference super_resolution.py:133-151 per frame);
  B. kernel estimation, robustness and the merge are local in the image plane (a few pixels of halo plus the flow)
     -> row-parallel: the output is cut into G row slabs and rank j runs steps B for ALL frames on the raw rows its
     slab needs (slab + |flow| + HALO rows), finishing its slab completely: reference frame, normalisation.
  The only data-path exchange is the all-gather of the flow fields (ny x nx x 2 floats per frame: 376 kB at 12 MP, 7 MB
  per 20-frame burst).  It can be PIPELINED (round 4, config.hip.stage_frames): the burst is cut into STAGES of whole
  rounds (round r = frames r G .. r G + G - 1, one per rank: stage_plan) and stage s's flows are gathered and its frames'
  step B (raw pass, robustness on the slab) runs while stage s + 1 is being aligned — step A on one HIP stream, step B
  on another, the RCCL all-gather between them, every piece a HIP graph on replay (RowsPlan).  Measured per-rank
  compute says ONE stage is faster on this hardware (STAGE_FRAMES below), so that is the default: the same plan with a
  single all-gather, and still no host read between step A and step B.  The sub-image extent
  needs a bound on |flow_y| BEFORE the flows exist: the plan is captured with the bound its first (eager) burst measured
  plus a margin, every burst checks its gathered flows against it on the device, the host reads that flag when the last
  stage's step B has been queued (the merge is still running: nothing waits for the read) and a burst that exceeds the
  bound is recomputed eagerly with its own measured bound, which the next capture takes over.  Slabs are UNEVEN: ranks
  that align one frame fewer take more rows (slab_bounds with `align_cost`).
  Host-resident frames: a rank uploads its own N/G frames whole (step A) and rows [S0, S1) of every frame (step B):
  (N/G + N (1/G + 2 halo/H)) frames' worth of bytes instead of N.

"reduce" (the north star's formulation).  Frames shard one per GPU round-robin; every rank runs the WHOLE single-GPU
  chain on its frames and merges them into float32 accumulators num / den [sH, sW, 3] (reference merge.py:432-434);
  ONE reduce-scatter (sum) over row slabs — one RCCL call on the packed [world, 2, rows, sW, 3] buffer — leaves rank j with the summed
  accumulators of slab j, to which it adds the reference frame and normalises.  1.15 GB of accumulators per rank at
  12 MP x2 cross xGMI (per-link bound for a ring: modelled 8-10 ms against 2.7 ms of measured compute per rank at G = 8), so
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
import os

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


def slab_align(scale):
    """Row granularity of the UNEVEN slabs of strategy "rows": the workgroup grid of the merge kernel that scale runs —
    16 LR rows = 16 * scale output rows at integer scales (32 at x2, 48 at x3), SLAB_ALIGN otherwise.  (At x2 of 12 MP on 8
    ranks the model wants 643 / 814 rows; in 96-row units that became 576 / 864 and the ranks differed by 8 %.)"""
    s = float(scale)
    return 16 * int(s) if s.is_integer() and s >= 1.0 else SLAB_ALIGN


def slab_rows(sH, world):
    """Rows per EQUAL slab: ceil(sH / world) rounded up to SLAB_ALIGN.  All slabs have this (padded) size so that
    every collective moves equal chunks; slab j covers output rows [j*rows, min((j+1)*rows, sH))."""
    rows = -(-sH // world)
    return -(-rows // SLAB_ALIGN) * SLAB_ALIGN


STAGE_FRAMES = 0  # frames per stage of strategy "rows" (config.hip.stage_frames); 0 = ONE stage: step A of all the rank's
                  # frames in one batched launch per kernel, one all-gather, step B.  Measured (tools/debug/emulate_ranks.py, one
                  # MI355X running each rank's graphs, 12 MP x 20 x2, per-rank compute at G = 8): one stage 1.85 ms, stages of
                  # >= 4 frames (3 stages) 2.22 ms — the step-A kernels of ONE frame per launch are latency-bound (they were
                  # sized for chunks of 4), and step A and step B on two streams do not overlap: either fills the GPU
                  # (G = 1: 9.81 ms pipelined, 3.81 + 5.84 alone).  Staging pays only where the all-gather's latency does.
def align_cost(scale):
    """rho: step A of one frame costs about as much as step B of one frame over this fraction of the image.  Step A does
    not depend on the scale, step B grows with the output: B / A = 0.20 + 0.32 scale^2 fits the per-rank measurements of
    tools/debug/emulate_ranks.py at x2 (12 MP: A 0.18 ms per frame, B 0.27 ms per frame and image: rho 0.67) and x3
    (48 MP: 0.70 / 2.2 ms: rho 0.32).  Re-fitted late in round 4 (the FFT got faster: A 0.15 ms per frame at 12 MP,
    0.68 ms at 48 MP; B 0.276 / 2.2 ms per frame and image): B / A = 0.74 + 0.276 scale^2, rho 0.54 at x2 and 0.31 at x3.
    config.hip.align_cost overrides it."""
    return 1.0 / (0.74 + 0.276 * float(scale) ** 2)


ALIGN_COST = align_cost(2)  # (the x2 value; main_sharded uses align_cost(config.scale))


def slab_bounds(sH, world, n_frames=0, align_cost=0.0, align=SLAB_ALIGN):
    """Valid (un-padded) row range [b[j], b[j+1]) of every slab, boundaries on multiples of `align` (uneven slabs; equal
    slabs: SLAB_ALIGN).
    align_cost = 0: equal slabs (strategy "reduce": the reduce-scatter moves equal chunks).  Strategy "rows" passes the
    frame count and align_cost = rho: rank j aligns a_j = |{j, j+G, ...}| frames, and its slab is sized so that
    a_j rho sH + rows_j n is the same for every rank — rows_j = sH ((1 + rho) / G - a_j rho / n): with 19 frames on 8 ranks
    and rho = 0.54, align = 32 the three ranks that align 3 frames get 640 rows, the five that align 2 get 784 - 832 (x2 of 12 MP)."""
    if not n_frames or align_cost <= 0.0 or world == 1:
        rows = slab_rows(sH, world)
        return [min(j * rows, sH) for j in range(world + 1)]
    want = [max(0.0, sH * ((1.0 + align_cost) / world - len(range(j, n_frames, world)) * align_cost / n_frames))
            for j in range(world)]
    # apportion the SLAB_ALIGN-row units: every rank at least one (while there are enough — a rank without rows would
    # make the whole job fall back to the eager path), the rest by the model, largest remainders first
    units = -(-sH // align)
    if units < world:
        rows = slab_rows(sH, world)
        return [min(j * rows, sH) for j in range(world + 1)]
    total = sum(want) or 1.0
    share = [max(0.0, w / total * units - 1.0) for w in want]      # beyond the guaranteed unit
    scale_ = (units - world) / (sum(share) or 1.0)
    exact = [1.0 + sh * scale_ for sh in share]
    got = [int(e) for e in exact]
    for j in sorted(range(world), key=lambda j: exact[j] - got[j], reverse=True)[: units - sum(got)]:
        got[j] += 1
    b, acc = [0], 0
    for j in range(world):
        acc += got[j]
        b.append(min(sH, acc * align))
    b[-1] = sH
    return b


def stage_plan(n_frames, world, min_frames=4):
    """Stages of the pipelined "rows" strategy: [(first round, rounds)], whole rounds each (round r = frames r G ..
    r G + G - 1, frame i aligned by rank i % G), at least `min_frames` frames per full stage (the robustness kernel
    shares its pass over the reference planes among 4 frames; the batched front end wants several frames per launch).
    min_frames >= n_frames: ONE stage = step A of every frame, one all-gather, step B (the un-pipelined form)."""
    rounds = -(-n_frames // world) if n_frames else 0
    per = max(1, -(-int(min_frames) // world))
    return [(r, min(per, rounds - r)) for r in range(0, rounds, per)]


def stage_frames(stage, n_frames, world, rank=None):
    """Frame indices of a stage (all of them, or the ones `rank` aligns), in frame order."""
    r0, k = stage
    idx = range(r0 * world, min(n_frames, (r0 + k) * world))
    return [i for i in idx if rank is None or i % world == rank]


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
        self._plans, self._plan_seen, self._bounds = {}, {}, {}  # strategy "rows": RowsPlan per input set, measured flow bounds
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
            self._runner, self._plans, self._plan_seen = None, {}, {}
            self._runner_p, self._runners_f = None, {}
            self._host = None

    def init_ref(self, ref_img):
        """Replicated on every rank: the reference frame's alignment state (step A needs the whole frame)."""
        from .super_resolution import BurstPipeline

        self.pipe = BurstPipeline(self.config)
        self.device = self.pipe.device
        self.pipe.init_ref(ref_img, robustness=False)
        return self

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
        """Step B in one piece (eager): output rows [r0, r1) from ALL frames (flows: [N-1, ny, nx, 2]).  Returns (slab
        float32 [r1 - r0, sW, 3], accumulated robustness of the raw rows [ceil(r0 / scale), ceil(r1 / scale)) — the rows
        whose first output row lies in the slab: a disjoint cover over the slabs — or None)."""
        return self._merge_rows(comp_imgs, flows, r0, r1, max_flow_y, self.pipe.ref if ref_dev is None else ref_dev)

    def rows_open(self, ref_img, comp_imgs, stages, rank, world, rows, max_flow, group=None, every_rank_has_rows=True):
        """Strategy "rows": the per-rank context main_sharded drives stage by stage (align / gather buffer / front /
        finish).  Device-resident float32 bursts that an engine sees again are replayed from HIP graphs, pipelined over
        two streams (RowsPlan: first call eager — it measures the flow bound —, second call captures); everything else
        runs eagerly (EagerRows: step A per stage, the bound measured from all flows, then step B in one piece)."""
        from .graph import capturable, _key

        packed = torch.is_tensor(comp_imgs)
        tensors = (ref_img, comp_imgs) if packed else (ref_img, *comp_imgs)
        # The ranks must agree on eager / capture / replay — they run different collectives (an eager rank all-gathers, a
        # capturing rank first all-reduces the consensus flag, a plan whose bound is exceeded re-runs eagerly).  Every
        # condition below is therefore one that an SPMD caller produces identically on every rank: properties of the
        # tensors, and whether the caller handed over THE SAME TENSORS (storages) as on an earlier call — NOT whether two
        # calls' tensors merely share addresses: the caching allocator may give a fresh burst an old address on one rank
        # and not on another (slab sizes differ per rank).  The address key is only trusted while the tensors it was
        # made from are alive: a plan keeps them alive itself (RowsPlan.keep); a first sighting keeps weak references.
        ok = capturable(self.config, tensors) and all(t.dtype == torch.float32 and t.is_contiguous() for t in tensors) \
            and len(comp_imgs) > 0 and every_rank_has_rows
        if not ok:
            return EagerRows(self, ref_img, comp_imgs, stages, rank, world, rows, max_flow)
        self._check_config()
        key = (_key(tensors), tuple(stages), rank, world, tuple(rows), None if max_flow is None else float(max_flow))
        plan = self._plans.get(key)
        if plan is not None:
            return plan.open()
        seen = self._plan_seen.get(key)
        if seen is not None and not all(w() is not None for w in seen["refs"]):
            seen = None  # the burst this entry was made for is gone; another one got its addresses: a first sighting
            self._plan_seen.pop(key)
            self._bounds.pop(key, None)
        if seen is None:  # first burst of this input set: eager; remembers the bound it measured
            while len(self._plan_seen) >= 64:  # (a caller that hands over fresh tensors every burst never replays: bounded)
                old = next(iter(self._plan_seen))
                self._plan_seen.pop(old)
                self._bounds.pop(old, None)
            self._mark_seen(key, tensors)
            return EagerRows(self, ref_img, comp_imgs, stages, rank, world, rows, max_flow, remember=key)
        bound = float(max_flow) if max_flow is not None else self._bounds.get(key)
        if bound is None:  # (cannot happen: the eager call stored it)
            return EagerRows(self, ref_img, comp_imgs, stages, rank, world, rows, max_flow, remember=key)
        if seen["state"] == "failed":
            return EagerRows(self, ref_img, comp_imgs, stages, rank, world, rows, max_flow)
        plan = None
        try:
            plan = RowsPlan(self, ref_img, comp_imgs, stages, rank, world, rows, bound, check=max_flow is None, key=key)
        except Exception as e:  # not capturable after all: stay eager for this input set
            self._plan_error = e
            torch.cuda.synchronize(ref_img.device)
        # every rank or none: a plan that finds its flow bound exceeded re-runs the burst eagerly (more all-gathers), an
        # eager rank never does — ranks in different modes would stop matching each other's collectives.  One all-reduce,
        # once per input set (the capture call), settles it.
        if world > 1 and dist.is_available() and dist.is_initialized():
            ok_t = torch.tensor([1 if plan is not None else 0], dtype=torch.int32,
                                device=ref_img.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN, group=group)
            if int(ok_t.item()) == 0:
                plan = None
        if plan is None:
            seen["state"] = "failed"
            return EagerRows(self, ref_img, comp_imgs, stages, rank, world, rows, max_flow)
        while len(self._plans) >= 2:  # (a plan holds a burst's intermediates of this rank)
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = plan
        return plan.open()

    def _mark_seen(self, key, tensors):
        """First sighting of an input set (or a fresh start after its plan was dropped): weak references to the tensors'
        storages-owning objects — a later call with the same address key counts as "the same inputs again" only while
        they are alive (see rows_open).  The liveness is tied to the Python tensor OBJECTS (views of one base count as the
        base): a caller that re-wraps the same storages in fresh tensor objects for every burst (.detach(), DLPack,
        as_tensor of a pointer) is a first sighting every time — correct, ranks agree, but the rows plan is never captured
        and every burst runs eagerly.  Pass the same tensors (or views of one base); `never_replayed` counts the sightings
        that found their earlier wrappers dead, for whoever wonders why nothing is replayed."""
        import weakref

        old = self._plan_seen.get(key)
        if old is not None and any(r() is None for r in old["refs"]):
            self.never_replayed = getattr(self, "never_replayed", 0) + 1

        self._plan_seen[key] = {"refs": tuple(weakref.ref(t._base if t._base is not None else t) for t in tensors),
                                "state": "seen"}

    def _merge_rows(self, comp_imgs, flows, r0, r1, max_flow_y, ref_dev):
        flows = flows.contiguous()
        b = SlabWork(self, ref_dev, r0, r1, max_flow_y, int(flows.shape[1]))
        n = len(comp_imgs)
        b.front([comp_imgs[i] for i in range(n)], [flows[i] for i in range(n)])
        return b.finish()

    # ---- strategy "reduce": frame-sharded merge + one reduce-scatter of the accumulators ---------------------------------
    def partial(self, ref_img, my_frames, bounds, rows):
        """This rank's frames through the whole single-GPU chain and into raw accumulators (no reference frame, no
        normalisation), laid out for ONE reduce-scatter: acc float32 [world, 2, rows, sW, 3] — acc[j, 0] / acc[j, 1] are
        num / den of output rows [bounds[j], bounds[j + 1]) (one merge launch per slab; the padding rows of the equal
        chunks stay zero) — and the rank's accumulated robustness [H, W] or None.  Replayed from a HIP graph for
        device-resident inputs."""
        from .graph import GraphRunner, capturable

        world = len(bounds) - 1

        def fn(ref, *frames):
            from .super_resolution import BurstPipeline
            from .merge import merge_burst, can_fuse_acc_r

            cfg = self.config
            self.pipe = pipe = BurstPipeline(cfg).init_ref(ref)
            self.device = pipe.device
            H, W = pipe.ref.shape
            sH, sW = pipe.output_size()
            acc = torch.zeros((world, 2, rows, sW, 3), dtype=torch.float32, device=self.device)
            # (denoiser: the ranks' partial sums are float64 like the reference's sum — robustness.RobustnessSum; they are
            # all-reduced in float64 and only then turned into the map the reference frame's merge decides on)
            acc_r = (torch.zeros((H, W), dtype=torch.float64 if self.denoiser_on else torch.float32, device=self.device)
                     if self.accumulate_r else None)
            if frames:
                fuse_acc = acc_r is not None and not self.denoiser_on and can_fuse_acc_r(cfg)
                fuse_min = pipe.fuses_local_min() and (fuse_acc or acc_r is None)
                fr = pipe.process_frames(list(frames), None if (fuse_acc or self.denoiser_on) else acc_r, fuse_local_min=fuse_min)
                if self.denoiser_on:
                    for f in fr:
                        acc_r.add_(f[3])
                for j in range(world):
                    b0, b1 = bounds[j], bounds[j + 1]
                    if b1 > b0:
                        merge_burst(fr, None, None, acc[j, 0, : b1 - b0], acc[j, 1, : b1 - b0], pipe.cfa, cfg, do_ref=False,
                                    divide=False, store_den=True, acc_r=acc_r if fuse_acc else None, rows=(b0, b1 - b0),
                                    out_height=sH, local_min=fuse_min)
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

    def finish_rows(self, acc_slab, r0, r1, ref_dev, ref_covs, acc_r=None):
        """Reference frame + normalisation of output rows [r0, r1) on top of the reduced accumulators
        acc_slab [2, rows, sW, 3] (merge.py:83-233, utils.py:85): slab float32 [r1 - r0, sW, 3].  With the
        accumulated-robustness denoiser (merge.py:223-228) `acc_r` is the REDUCED robustness [H, W] and the reference
        frame goes through the sequential operator on the sub-image of the slab (like SlabWork)."""
        from .graph import GraphRunner, capturable
        from .merge import merge_burst, merge_ref
        from .utils import divide

        cfg = self.config
        H, W = ref_dev.shape
        sH = round(cfg.scale * H)
        n = r1 - r0

        def fn(acc, ref, covs, *rob):
            if self.denoiser_on:
                scale, ts = cfg.scale, self.tile_size()
                pow2 = float(scale) in (1.0, 2.0, 4.0, 8.0)
                S0, S1, row0 = sub_image_rows(r0, r1, scale, H, ts, 0.0, from_top=not pow2)
                q = 1 if self.pipe.mono else 2  # covariances: one per Bayer quad (monochrome: per pixel)
                num = torch.zeros((int(round(scale * (S1 - S0))), acc.shape[2], 3), dtype=torch.float32, device=acc.device)
                den = torch.zeros_like(num)
                num[row0:row0 + n] = acc[0, :n]
                den[row0:row0 + n] = acc[1, :n]
                from .robustness import RobustnessSum

                merge_ref(ref[S0:S1], covs[S0 // q:S1 // q], num, den, self.pipe.cfa, cfg,
                          RobustnessSum.decisions_of(rob[0][S0:S1].to(torch.float64),
                                                     cfg.accumulated_robustness_denoiser.merge.max_frame_count))
                divide(num, den)
                return num[row0:row0 + n].contiguous()
            num, den = acc[0, :n], acc[1, :n]
            merge_burst([], ref, covs, num, den, self.pipe.cfa, cfg, load_acc=True, do_ref=True, divide=True,
                        rows=(r0, n), out_height=sH)
            return num

        tensors = (acc_slab, ref_dev, ref_covs) + ((acc_r,) if self.denoiser_on else ())
        if not capturable(cfg, tensors):
            return fn(*tensors)
        key = (int(r0), int(r1))
        runner = self._runners_f.get(key)
        if runner is None:
            runner = self._runners_f[key] = GraphRunner(fn, ref_dev.device)
        return runner(*tensors)


class SlabWork:
    """Step B of one rank as pieces: sub-image setup (reference-frame state of the raw rows [S0, S1) the slab depends on),
    front(frames, flows) for any subset of the frames — raw pass + robustness on the sub-image, or with the denoiser the
    sequential per-frame merge — and finish(): fused merge + reference frame + normalisation of the slab.  The eager path
    calls front() once with every frame; RowsPlan captures setup / one front() per stage / finish() as separate graphs."""

    def __init__(self, eng, ref_dev, r0, r1, max_flow_y, ny_full, ref_wait=True):
        from .super_resolution import BurstPipeline
        from .merge import can_fuse_acc_r
        from .robustness import RobustnessSum

        cfg = self.cfg = eng.config
        self.eng = eng
        H, W = ref_dev.shape
        scale, ts = cfg.scale, eng.tile_size()
        sW = round(scale * W)
        pow2 = float(scale) in (1.0, 2.0, 4.0, 8.0)
        # (the per-frame operator path of the denoiser has no row offset: for scales whose positions idx / scale are
        # not exact in float32 its sub-image starts at the top of the frame, where sub-image = full-frame coordinates)
        S0, S1, row0 = sub_image_rows(r0, r1, scale, H, ts, max_flow_y, from_top=eng.denoiser_on and not pow2)
        self.S0, self.S1, self.row0, self.nrows = S0, S1, row0, r1 - r0
        Hs = S1 - S0
        self.sHs = int(round(scale * Hs))
        self.t0, self.t1 = S0 // ts, -(-S1 // ts)
        dev = ref_dev.device
        sub = self.sub = BurstPipeline(cfg, dev)
        sub.ref_wait = ref_wait
        sub.init_ref(ref_dev[S0:S1], alignment=False)  # device-resident rows of the replicated reference frame
        # row slices = VIEWS of the full fields + the tile rows around them: the flow-irregularity weight S of the
        # sub-image's first / last tile row is evaluated on the full field (module docstring)
        sub.flow_rows = (self.t0, int(ny_full) - self.t1)
        self.out = torch.empty((self.nrows, sW, 3), dtype=torch.float32, device=dev)
        # (the denoiser decides on the sum: float64 like the reference's, robustness.RobustnessSum — as in main())
        self.acc_sum = RobustnessSum((Hs, W), dev) if eng.denoiser_on else None
        self.acc_r = torch.zeros((Hs, W), dtype=torch.float32, device=dev) if (eng.accumulate_r and not eng.denoiser_on) else None
        self.L0 = int(math.ceil(r0 / scale)) - S0
        self.L1 = min(Hs, int(math.ceil(r1 / scale)) - S0)
        self.frames = []
        if eng.denoiser_on:
            # the accumulated-robustness denoiser (merge.py:223-228) needs sum_n r_n before the reference frame is
            # merged: sequential operator path on the sub-image
            self.num = torch.zeros((self.sHs, sW, 3), dtype=torch.float32, device=dev)
            self.den = torch.zeros_like(self.num)
        else:
            self.fuse_acc = self.acc_r is not None and can_fuse_acc_r(cfg)
            self.fuse_min = sub.fuses_local_min() and (self.fuse_acc or self.acc_r is None) and row0 % slab_align(scale) == 0

    def front(self, imgs, flows):
        """`imgs`: full frames (their rows [S0, S1) are used); `flows`: their FULL flow fields [ny, nx, 2] (views welcome)."""
        from .merge import merge

        sub, S0, S1 = self.sub, self.S0, self.S1
        sub_flows = [f[self.t0:self.t1] for f in flows]
        if self.eng.denoiser_on:
            for img, fl in zip(imgs, sub_flows):
                raw, flow, covs, r = sub.process_frame(img[S0:S1], None, flow=fl)
                self.acc_sum.add(r)
                merge(raw, flow, covs, r, self.num, self.den, sub.cfa, self.cfg)
        elif imgs:
            self.frames += sub.process_frames([img[S0:S1] for img in imgs], None if self.fuse_acc else self.acc_r,
                                              fuse_local_min=self.fuse_min, flows=sub_flows)

    def finish(self):
        from .merge import merge_ref, merge_burst
        from .utils import divide

        sub, cfg = self.sub, self.cfg
        if self.eng.denoiser_on:
            merge_ref(sub.ref, sub.ref_covs, self.num, self.den, sub.cfa, cfg,
                      self.acc_sum.for_decisions(cfg.accumulated_robustness_denoiser.merge.max_frame_count))
            divide(self.num, self.den)
            self.out.copy_(self.num[self.row0:self.row0 + self.nrows])
            return self.out, self.acc_sum.mask((self.L0, self.L1))
        else:
            merge_burst(self.frames, sub.ref, sub.ref_covs, self.out, None, sub.cfa, cfg, do_ref=True, divide=True,
                        acc_r=self.acc_r if (self.fuse_acc and self.frames) else None, rows=(self.row0, self.nrows),
                        out_height=self.sHs, local_min=self.fuse_min, lr_row_offset=self.S0)
        return self.out, (self.acc_r[self.L0:self.L1] if self.acc_r is not None else None)


class EagerRows:
    """Strategy "rows" without graphs, for any engine with init_ref / align_frames / merge_rows (HipEngine on host frames,
    first bursts, timers, debug; the NumPy engine of the CPU tests): step A stage by stage, then — the flow bound is
    MEASURED here, so every flow has to be known — step B in one piece."""

    streams = None  # collectives on the current stream

    def __init__(self, eng, ref_img, comp_imgs, stages, rank, world, rows, max_flow, remember=None):
        self.eng, self.ref_img, self.comp_imgs = eng, ref_img, comp_imgs
        self.stages, self.rank, self.world, self.rows, self.max_flow = stages, rank, world, rows, max_flow
        self.remember = remember
        self.gathered = []
        self.ready = False

    def _ensure_ref(self):
        if not self.ready:
            self.eng.init_ref(self.ref_img)
            self.ready = True

    def align(self, s):
        self._ensure_ref()
        mine = stage_frames(self.stages[s], len(self.comp_imgs), self.world, self.rank)
        local = self.eng.align_frames([self.comp_imgs[i] for i in mine])  # [len(mine), ny, nx, 2]
        padded = torch.zeros((self.stages[s][1], *local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
        return padded

    def gather_buffer(self, s, like):
        return torch.empty((self.world, *like.shape), dtype=like.dtype, device=like.device)

    def front(self, s, gathered):
        self.gathered.append(gathered)

    def finish(self):
        self._ensure_ref()
        eng, n = self.eng, len(self.comp_imgs)
        H, W = eng.shape()
        r0, r1 = self.rows
        flows, extra, max_flow = None, {}, self.max_flow
        if n:
            # frame i was aligned by rank i % world in round i // world: slot (i // world - first round) of its stage
            flows = torch.cat([g.transpose(0, 1).reshape(-1, *g.shape[2:]) for g in self.gathered])[:n].contiguous()
            if max_flow is None:
                max_flow = float(flows[..., 1].abs().max())                     # the one host read of the eager step
                if not math.isfinite(max_flow):
                    max_flow = float(H)
                if self.remember is not None:  # the bound the captured plan of this input set will use
                    eng._bounds[self.remember] = min(float(H), 1.25 * max_flow + 2.0)
            else:  # caller's bound: no host read; the check stays on the device
                extra["flow_bound_exceeded"] = ~(flows[..., 1].abs().amax() <= float(max_flow))
        max_flow = 0.0 if max_flow is None else float(max_flow)
        slab, acc_r = (eng.merge_rows(self.comp_imgs, flows, r0, r1, max_flow) if r1 > r0 else (None, None))
        self.device = flows.device if flows is not None else (slab.device if slab is not None else torch.device("cpu"))
        return slab, acc_r, extra, False


def _flow_grid_of(shape, config):
    """(ny, nx) of the flow field of an [H, W] frame: tiles of the reference frame padded up to whole tiles
    (alignment.py:27-37; BurstPipeline.flow_grid once the reference state exists)."""
    ts = int(config.block_matching.tuning.tile_size)
    return -(-int(shape[0]) // ts), -(-int(shape[1]) // ts)


_plan_streams = {}  # device index -> the step-A stream of every RowsPlan of the process (step B: graph.shared_streams()[0])


class RowsPlan:
    """Strategy "rows" of one rank for one set of device-resident float32 inputs as HIP graphs on TWO streams:

        s_a:  g_ref_a (reference alignment state)   g_a[0]   g_a[1]   g_a[2] ...          (step A of the rank's frames)
        RCCL:                                          all-gather 0  all-gather 1 ...       (flows of a stage: G x k x 376 kB)
        s_b:  g_ref_b (reference state of the slab's rows)      g_b[0]    g_b[1] ...  g_fin (raw pass + robustness of a
                                                                                           stage's frames on the slab; merge)

    so the latency-bound step-A kernels of stage s + 1 (barrier-bound FFT phases, coarse pyramid levels) run next to the
    VALU-bound step-B kernels of stage s, and no stream ever waits for the host.  The sub-image extent is fixed at capture
    (`bound` on |flow_y|, rounded up to 8 rows); every g_b checks its gathered flows against it on the device and
    finish() reads the flag once the merge has been queued — exceeded: the caller recomputes the burst eagerly."""

    def __init__(self, eng, ref, comp_imgs, stages, rank, world, rows, bound, check, key):
        from . import _lib
        from .graph import shared_streams, capture
        from .super_resolution import BurstPipeline, _stream_pool
        from .utils_image import _grey_plan, _grey_plans

        cfg, dev = eng.config, ref.device
        self.eng, self.key, self.check, self.stages, self.world, self.rank = eng, key, check, stages, world, rank
        n, G = len(comp_imgs), world
        comps = [comp_imgs[i] for i in range(n)]
        self.keep = (ref, comp_imgs)
        self.bound = float(math.ceil(float(bound) / 8.0) * 8)
        self.device = dev
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.s_b = shared_streams(dev)[0]
        if idx not in _plan_streams:
            _plan_streams[idx] = torch.cuda.Stream(dev)
        self.s_a = _plan_streams[idx]
        r0, r1 = rows
        H, W = ref.shape
        with torch.cuda.device(dev):
            pipe = self.pipe = BurstPipeline(cfg, dev)
            pipe.ref_wait = False  # the pieces are separate captures: their order is the order of the replays on s_a / s_b
            ns = pipe._n_streams(None)
            pool = _stream_pool.setdefault(idx, [])
            if len(pool) < ns:
                pool += [torch.cuda.Stream(dev) for _ in range(ns - len(pool))]
            if cfg.grey_method == "FFT" and not pipe.mono:  # plans allocate: they have to exist before their stream is captured
                for st in (self.s_a, *pool[:ns]):
                    with torch.cuda.stream(st):
                        _grey_plan(H, W, dev, _lib.MAX_BATCH)
            torch.cuda.synchronize(dev)
            # Round 5: the reference alignment state and step A of the rank's first stage are ONE graph — the frames' own
            # grey images and pyramids (side streams forked at the start of the reference precompute,
            # BurstPipeline._on_streams) run next to the single-frame, latency-bound reference kernels instead of behind
            # them: at G = 8 the replicated reference state was 0.27 ms of a rank's 1.65 ms.  (A rank without frames in
            # stage 0 keeps the separate reference graph.)
            ny_nx = _flow_grid_of(ref.shape, cfg)
            eng.pipe, eng.device = pipe, dev
            first = stage_frames(stages[0], n, G, rank) if stages else []
            fuse_ref = bool(first) and os.environ.get("HHSR_ROWS_SPLIT_REF") is None  # (A/B switch, read at capture)
            self.g_ref_a = None
            if not fuse_ref:
                self.g_ref_a = torch.cuda.CUDAGraph()
                with capture(self.g_ref_a, self.s_a):
                    pipe.init_ref(ref, robustness=False)
            ny, nx = ny_nx
            self.local, self.gath, self.g_a = [], [], []
            for k, st in enumerate(stages):
                mine = stage_frames(st, n, G, rank)
                loc = torch.zeros((st[1], ny, nx, 2), dtype=torch.float32, device=dev)
                self.local.append(loc)
                self.gath.append(torch.zeros((G, st[1], ny, nx, 2), dtype=torch.float32, device=dev))
                g = None
                if mine:
                    g = torch.cuda.CUDAGraph()
                    with capture(g, self.s_a):
                        if k == 0 and fuse_ref:
                            pipe.ref_wait = True
                            pipe.init_ref(ref, robustness=False)
                        loc[: len(mine)].copy_(eng.align_frames([comps[i] for i in mine]))
                        pipe.ref_wait = False
                self.g_a.append(g)
            assert tuple(pipe.flow_grid()) == (ny, nx)
            self.flag = torch.zeros((1,), dtype=torch.bool, device=dev)
            self.flag_host = torch.zeros((1,), dtype=torch.bool).pin_memory()
            self.g_ref_b = torch.cuda.CUDAGraph()
            with capture(self.g_ref_b, self.s_b):
                self.flag.zero_()
                work = self.work = SlabWork(eng, ref, r0, r1, self.bound, ny, ref_wait=False)
            self.g_b = []
            for st, gat in zip(stages, self.gath):
                fr = stage_frames(st, n, G)
                g = torch.cuda.CUDAGraph()
                with capture(g, self.s_b):
                    self.flag.logical_or_(~(gat[..., 1].abs().amax() <= self.bound))
                    work.front([comps[i] for i in fr], [gat[i % G, i // G - st[0]] for i in fr])
                self.g_b.append(g)
            self.g_fin = torch.cuda.CUDAGraph()
            with capture(self.g_fin, self.s_b):
                self.out, self.acc_r = work.finish()
            self.e_flag = torch.cuda.Event()
            self.plans = list(_grey_plans.values())  # the graphs hold the FFT plans' spectrum buffers: keep them alive
        self.streams = (self.s_a, self.s_b)

    def open(self):
        self.cur = cur = torch.cuda.current_stream(self.device)
        self.s_a.wait_stream(cur)
        self.s_b.wait_stream(cur)
        if self.g_ref_a is not None:  # (else: part of the first stage's step-A graph)
            with torch.cuda.stream(self.s_a):
                self.g_ref_a.replay()
        with torch.cuda.stream(self.s_b):
            self.g_ref_b.replay()
        self.eng.pipe, self.eng.device = self.pipe, self.device
        return self

    def align(self, s):
        if self.g_a[s] is not None:
            with torch.cuda.stream(self.s_a):
                self.g_a[s].replay()
        return self.local[s]

    def gather_buffer(self, s, like):
        return self.gath[s]

    def front(self, s, gathered):
        with torch.cuda.stream(self.s_b):
            self.g_b[s].replay()

    def replay_compute_only(self):
        """This rank's whole step WITHOUT its collectives: every graph of the plan in order, the gather buffers still
        holding the flows of the last real burst — per-rank compute of the G-rank job (what tools/debug/emulate_ranks.py
        measures on one GPU, here on the rank's own GPU; bench.py reports max over ranks next to the real step time)."""
        ctx = self.open()
        for s in range(len(self.stages)):
            ctx.align(s)
            with torch.cuda.stream(self.s_b):
                self.s_b.wait_stream(self.s_a)  # (the all-gather's dependency: step B of a stage follows its step A)
            ctx.front(s, None)
        return ctx.finish()

    def finish(self):
        with torch.cuda.stream(self.s_b):
            if self.check:
                self.flag_host.copy_(self.flag, non_blocking=True)
                self.e_flag.record(self.s_b)
            self.g_fin.replay()
        self.cur.wait_stream(self.s_a)
        self.cur.wait_stream(self.s_b)
        if self.check:
            self.e_flag.synchronize()  # (the flag is known when the last stage's flows are there: the merge is still running)
            if bool(self.flag_host[0]):
                return None, None, {}, True
            return self.out, self.acc_r, {}, False
        return self.out, self.acc_r, {"flow_bound_exceeded": self.flag[0]}, False


def _staged(t, group):
    """Host-only backends (gloo, used by the CPU tests) get CPU copies of device tensors."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _all_gather(t, world, group):
    """Equal-size all-gather: [world, *t.shape].  RCCL ("nccl") gathers device tensors in place over xGMI."""
    src = t.cpu() if _staged(t, group) else t.contiguous()
    out = torch.empty((world, *src.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group) if src.is_cuda else dist.all_gather(list(out.unbind(0)), src, group=group)
    return out.to(t.device)


def _all_gather_stage(out, src, world, group, streams=None):
    """out [world, *src.shape] <- every rank's src (one stage's flow fields).  `streams` = (producer, consumer) HIP
    streams of a RowsPlan: RCCL reads `src` once the producer stream has written it and only the CONSUMER stream waits
    for the collective — the producer goes on aligning the next stage.  None: everything on the current stream.
    Host-only backends (gloo: CPU tests, ranks sharing one GPU) stage through host memory."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):  # (force_sharded without a process group)
        if streams is None or not src.is_cuda:
            out.copy_(src.unsqueeze(0))
        else:  # the "gather" runs on the consumer stream, behind the producer's work — like the collective would
            with torch.cuda.stream(streams[1]):
                streams[1].wait_stream(streams[0])
                out.copy_(src.unsqueeze(0))
        return
    if src.is_cuda and dist.get_backend(group) == "nccl":
        if streams is None:
            dist.all_gather_into_tensor(out, src, group=group)
            return
        with torch.cuda.stream(streams[0]):
            work = dist.all_gather_into_tensor(out, src, group=group, async_op=True)
        with torch.cuda.stream(streams[1]):
            work.wait()  # (a stream-level wait: the host does not block)
        return
    if src.is_cuda and streams is not None:
        with torch.cuda.stream(streams[0]):
            host = src.cpu()
    else:
        host = src.cpu() if src.is_cuda else src
    parts = torch.empty((world, *host.shape), dtype=host.dtype)
    dist.all_gather(list(parts.unbind(0)), host.contiguous(), group=group)
    if out.is_cuda and streams is not None:
        with torch.cuda.stream(streams[1]):
            out.copy_(parts)
    else:
        out.copy_(parts)


def _gather(t, dst, world, group):
    """Equal-size gather of `t` to global rank `dst`; returns the stacked tensor there, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()):
        return t.unsqueeze(0)
    me = dist.get_rank()
    staged = _staged(t, group)
    src = t.cpu() if staged else t.contiguous()
    out = torch.empty((world, *src.shape), dtype=src.dtype, device=src.device) if me == dst else None
    dist.gather(src, list(out.unbind(0)) if me == dst else None, dst=dst, group=group)  # received in place
    if me != dst:
        return None
    return out.to(t.device) if staged else out


def _reduce_scatter_rows(acc, world, group, buffers=None):
    """acc [world, 2, rows, sW, 3] (chunk j = num / den of slab j) -> this rank's summed slab [2, rows, sW, 3].  RCCL: ONE
    reduce-scatter over xGMI into a buffer that is kept across bursts (`buffers`: the engine's dict — the finishing step
    is a HIP graph bound to its address); host backends (gloo, CPU tests) have no reduce-scatter: all-reduce + slice."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    if not (dist.is_available() and dist.is_initialized()):
        return acc[0]
    if acc.is_cuda and dist.get_backend(group) == "nccl":
        shape = tuple(acc.shape[1:])
        out = buffers.get("rs_out") if buffers is not None else None
        if out is None or tuple(out.shape) != shape or out.device != acc.device:
            out = torch.empty(shape, dtype=acc.dtype, device=acc.device)
            if buffers is not None:
                buffers["rs_out"] = out
        dist.reduce_scatter_tensor(out, acc, op=dist.ReduceOp.SUM, group=group)
        return out
    host = acc.cpu() if acc.is_cuda else acc.clone()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    return host[rank].contiguous().to(acc.device)


def _all_reduce(t, group):
    if not (dist.is_available() and dist.is_initialized()):
        return t
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


def main_sharded(ref_img, comp_imgs, config, group=None, engine=None, gather=True, max_flow=None, strategy=None,
                 force_sharded=False):
    """Multi-GPU equivalent of main() (see the module docstring).

    gather=True : returns (output [sH, sW, 3], debug_dict) on rank 0 and (None, {}) on the other ranks;
    gather=False: every rank returns (its slab [rows_j, sW, 3], {"rows": (r0, r1), ...}) — the output stays sharded.
    `strategy`: "rows" (default; config.hip.strategy) or "reduce".
    `max_flow` (strategy "rows"; also config.hip.max_flow): bound on |flow_y| in pixels that sizes the sub-image halo.
    Given, the step runs WITHOUT any host read (debug["flow_bound_exceeded"] is a 0-dim device tensor the caller may
    inspect later: True = some flow left the halo, rows near slab seams are then not the single-GPU result); default:
    measured from the gathered flow fields — the one device-to-host read of a scalar of the step, after which the
    sub-image extent is rounded up to multiples of 8 rows so that the captured step-B graph is reused from burst to
    burst.  Works un-initialised / with world_size 1 (then it IS main(), unless `force_sharded` sends the one rank through
    the sharded code path — stages, collectives, slab finish — e.g. to run the RCCL call sites on a one-GPU box)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    eng = engine if engine is not None else HipEngine(config)
    n = len(comp_imgs)
    if world == 1 and not force_sharded:
        out, debug = eng.single(ref_img, comp_imgs)
        if not gather:
            debug = dict(debug, rows=(0, int(out.shape[0])))
        return out, debug
    root = dist.get_global_rank(group, 0) if group is not None else 0
    strategy = _strategy(config, strategy)
    if config.mode != "bayer":
        # the reference's one-channel robustness reads the statistics of row y / 2 for row y (robustness.py:358 with the
        # same-size map of :337-343): a row slab is not self-contained there.  Monochrome bursts therefore always shard by
        # FRAMES ("reduce": every rank runs whole frames; only the reference frame's merge and the normalisation, which
        # are row-local in this mode too, are done per slab)
        strategy = "reduce"
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
        acc, acc_r_part, ref_dev, ref_covs = eng.partial(ref_img, [comp_imgs[i] for i in mine], bounds, rows)
        red = _reduce_scatter_rows(acc, world, group, getattr(eng, "_buffers", None))   # [2, rows, sW, 3]: this rank's slab, summed
        if want_acc:
            acc_full = _all_reduce(acc_r_part, group)                       # [H, W], 48 MB at 12 MP
        if r1 > r0:
            # (the accumulated-robustness denoiser reads the REDUCED robustness when it merges the reference frame)
            slab = (eng.finish_rows(red, r0, r1, ref_dev, ref_covs, acc_full) if getattr(eng, "denoiser_on", False)
                    else eng.finish_rows(red, r0, r1, ref_dev, ref_covs))
        if want_acc:
            a0, a1 = int(math.ceil(r0 / config.scale)), min(H, int(math.ceil(r1 / config.scale)))
            if acc_full.dtype != torch.float32:  # (the denoiser's float64 sum: reported as the float32 map the API returns)
                acc_full = acc_full.to(torch.float32)
            acc_r = acc_full[a0:a1] if r1 > r0 else None
        dev = acc.device
    else:
        # ---- A: frame-parallel alignment, stage by stage, each stage's flows all-gathered; B: row-parallel kernels +
        # robustness of the stage's frames on this rank's slab as soon as they are there; then merge + reference frame +
        # normalisation of the slab (HipEngine on device-resident bursts: RowsPlan, pipelined over two streams)
        H, W = tuple(ref_img.shape)
        sH, sW = round(config.scale * H), round(config.scale * W)
        hip = config.get("hip", None) if hasattr(config, "get") else None
        if max_flow is None and hip is not None:
            max_flow = hip.get("max_flow", None)
        cost = align_cost(config.scale)
        if hip is not None and hip.get("align_cost", None) is not None:
            cost = float(hip.get("align_cost"))
        sf = int(hip.get("stage_frames", STAGE_FRAMES)) if hip is not None else STAGE_FRAMES
        stages = stage_plan(n, world, sf if sf > 0 else max(n, 1))
        bounds = slab_bounds(sH, world, n, cost, slab_align(config.scale))
        rows = max(b1 - b0 for b0, b1 in zip(bounds[:-1], bounds[1:]))  # (padded chunk of the optional gather)
        r0, r1 = bounds[rank], bounds[rank + 1]

        def run(ctx):
            for st in range(len(stages)):
                local = ctx.align(st)
                out_buf = ctx.gather_buffer(st, local)
                _all_gather_stage(out_buf, local, world, group, ctx.streams)
                ctx.front(st, out_buf)
            return ctx.finish()

        open_ = getattr(eng, "rows_open", None)
        nonempty = all(b1 > b0 for b0, b1 in zip(bounds[:-1], bounds[1:]))
        ctx = open_(ref_img, comp_imgs, stages, rank, world, (r0, r1), max_flow, group, nonempty) if open_ is not None else \
            EagerRows(eng, ref_img, comp_imgs, stages, rank, world, (r0, r1), max_flow)
        slab, acc_r, extra, retry = run(ctx)
        if retry:
            # some flow of this burst left the halo the captured plan was sized for (every rank sees the same gathered
            # flows, so every rank gets here): recompute eagerly with the measured bound; the next capture takes it over
            key = ctx.key
            eng._plans.pop(key, None)
            packed = torch.is_tensor(comp_imgs)
            eng._mark_seen(key, (ref_img, comp_imgs) if packed else (ref_img, *comp_imgs))
            ctx = EagerRows(eng, ref_img, comp_imgs, stages, rank, world, (r0, r1), None, remember=key)
            slab, acc_r, extra, _ = run(ctx)
            extra = dict(extra, flow_bound_recomputed=True)
        debug.update(extra)
        dev = getattr(ctx, "device", None) or (slab.device if slab is not None else torch.device("cpu"))
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
    if all(bounds[j + 1] - bounds[j] == rows for j in range(world - 1)):  # equal slabs: the padded chunks are the image
        out = gathered.view(world * rows, sW, 3)[:sH]
    else:
        out = torch.cat([gathered[j, : bounds[j + 1] - bounds[j]] for j in range(world)])
    debug = {"robustness": [], "flow": [], **{k: v for k, v in debug.items() if k.startswith("flow_bound")}}
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
