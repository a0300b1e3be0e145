"""Frame-sharded multi-GPU merge.  The reference is single-GPU; this is the one
parallelism the build adds.

Every comp frame's contribution to num/den depends only on the reference-frame state and that frame
(reference super_resolution.py:133-173) and contributions are summed (merge.py:432-434), so:
  * one process per GPU, rank k takes comp frames k, k+G, k+2G, ... (no data-path collective);
  * every rank replicates the cheap reference-frame precompute;
  * ONE exchange of the float32 accumulators: the output is cut into G row slabs; every rank merges its
    frames slab by slab into a slab-major buffer [G][2][rows][sW][3] (num | den per slab) and a single
    all-to-all (= a reduce-scatter on the fully connected xGMI fabric: 7 concurrent point-to-point
    transfers per GPU, 1/8 of the buffer each — RCCL's ring reduce would cross 7 links in sequence) hands
    slab j of every rank to rank j, which sums the G partials;
  * each rank adds the reference frame (Alg. 11) and normalises ITS slab — so the finish is parallel too — and
    sends the finished [rows][sW][3] slab to rank 0 (half the volume of the accumulators);
  * the [H][W] accumulated robustness, when requested, is a plain sum-reduce to rank 0.
The accumulated-robustness *denoiser* (merge.py:223-228 overwrite rule) needs the full accumulators on one
rank and falls back to one reduce + rank-0 finish.  The engine that does the per-rank compute is injected
so that the sharding / exchange logic can be exercised without a GPU (tests run it on gloo, world_size 2).
"""
import torch
import torch.distributed as dist

SLAB_ALIGN = 32  # slabs start on the 32-row workgroup grid of the x2 merge kernel (16-row grid of the generic one)


def shard_indices(n_frames, rank, world):
    """Indices of the comp frames rank `rank` of `world` processes (round-robin: balanced to +-1)."""
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


class HipEngine:
    """Per-rank compute on the local MI355X."""

    def __init__(self, config):
        from .super_resolution import BurstPipeline, denoiser_enabled

        self.config = config
        self.denoiser_on = denoiser_enabled(config)
        self.accumulate_r = self.denoiser_on or bool(config.robustness.save_mask)
        self.pipe = BurstPipeline(config)

    def init_ref(self, ref_img):
        self.pipe.init_ref(ref_img)
        return self

    def output_shape(self):
        return (*self.pipe.output_size(), 3)

    def partial(self, comp_imgs, world=1):
        """This rank's frames merged into slab-major accumulators: float32 [world][2][rows][sW][3] with
        rows = slab_rows(sH, world) (num | den per slab; rows past the image stay zero).  world = 1: one slab =
        the whole output.  Returns (acc, acc_r)."""
        from .merge import merge_burst, can_fuse_acc_r

        pipe = self.pipe
        sH, sW = pipe.output_size()
        rows = slab_rows(sH, world) if world > 1 else sH
        bounds = slab_bounds(sH, world) if world > 1 else [0, sH]
        padded = world * rows != sH
        alloc = torch.zeros if (padded or not comp_imgs) else torch.empty
        acc = alloc((world, 2, rows, sW, 3), dtype=torch.float32, device=pipe.device)
        acc_r = torch.zeros(tuple(pipe.ref.shape), dtype=torch.float32, device=pipe.device) if self.accumulate_r else None
        fuse_acc = acc_r is not None and can_fuse_acc_r(self.config) and len(comp_imgs) > 0
        fuse_min = pipe.fuses_local_min() and (fuse_acc or acc_r is None)  # 5x5 local minimum inside the merge
        frames = pipe.process_frames(list(comp_imgs), None if fuse_acc else acc_r, fuse_local_min=fuse_min)
        if not frames:
            return acc, acc_r
        for j in range(world):
            r0, r1 = bounds[j], bounds[j + 1]
            if r1 > r0:
                merge_burst(frames, None, None, acc[j, 0, : r1 - r0], acc[j, 1, : r1 - r0], pipe.cfa, self.config,
                            do_ref=False, divide=False, store_den=True, acc_r=acc_r if fuse_acc else None,
                            rows=(r0, r1 - r0), out_height=sH, local_min=fuse_min)
        return acc, acc_r

    def finish_slab(self, acc, row0, acc_r=None):
        """acc [2][rows][sW][3] (summed over ranks) -> finished output slab: reference frame + normalise."""
        from .merge import merge_burst, merge_ref
        from .utils import divide

        pipe = self.pipe
        sH, _ = pipe.output_size()
        self._ref_covs = pipe.ref_covs
        if acc.shape[1] == 0:
            return acc[0]
        if self.denoiser_on:  # whole image only (row0 == 0 and all rows)
            assert row0 == 0 and acc.shape[1] == sH
            merge_ref(pipe.ref, self._ref_covs, acc[0], acc[1], pipe.cfa, self.config, acc_r)
            divide(acc[0], acc[1])
        else:
            merge_burst([], pipe.ref, self._ref_covs, acc[0], acc[1], pipe.cfa, self.config, load_acc=True, do_ref=True,
                        divide=True, rows=(row0, acc.shape[1]), out_height=sH)
        return acc[0]


def _staged(t, group):
    """Host-only backends (gloo, used by the CPU tests) get CPU copies of device tensors."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _reduce_sum(t, dst, group):
    """Sum-reduce to `dst`.  RCCL ("nccl") reduces device tensors in place over xGMI."""
    if _staged(t, group):
        h = t.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return t
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return t


def _all_to_all(recv, send, group):
    """Equal-split all-to-all: chunk j of `send` goes to rank j."""
    if _staged(send, group):
        hr, hs = torch.empty(recv.shape, dtype=recv.dtype), send.cpu()
        dist.all_to_all_single(hr, hs, group=group)
        recv.copy_(hr)
    else:
        dist.all_to_all_single(recv, send, group=group)


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


def main_sharded(ref_img, comp_imgs, config, group=None, engine=None):
    """Frame-sharded equivalent of main().  Returns (output, debug_dict) on rank 0 and (None, {}) on
    the other ranks.  Works un-initialised / with world_size 1 (then it is main() without collectives)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    eng = engine if engine is not None else HipEngine(config)
    eng.init_ref(ref_img)
    mine = [comp_imgs[i] for i in shard_indices(len(comp_imgs), rank, world)]
    sH, sW, _ = eng.output_shape()
    root = dist.get_global_rank(group, 0) if (world > 1 and group is not None) else 0
    debug = {"robustness": [], "flow": []}

    if world == 1 or getattr(eng, "denoiser_on", False):
        # single exchange of the whole accumulators, finish on rank 0
        acc, acc_r = eng.partial(mine, 1)
        if world > 1:
            acc = _reduce_sum(acc, root, group)
            if acc_r is not None:
                acc_r = _reduce_sum(acc_r, root, group)
        if rank != 0:
            return None, {}
        out = eng.finish_slab(acc[0], 0, acc_r)
        if acc_r is not None:
            debug["accumulated robustness"] = acc_r
        return out, debug

    rows = slab_rows(sH, world)
    bounds = slab_bounds(sH, world)
    acc, acc_r = eng.partial(mine, world)              # [world][2][rows][sW][3]
    recv = torch.empty_like(acc)
    _all_to_all(recv, acc, group)                        # recv[k] = rank k's partial of MY slab
    del acc
    summed = recv.sum(dim=0)                             # [2][rows][sW][3]
    del recv
    valid = bounds[rank + 1] - bounds[rank]
    out_slab = torch.zeros((rows, sW, 3), dtype=torch.float32, device=summed.device)
    if valid > 0:
        out_slab[:valid] = eng.finish_slab(summed[:, :valid].contiguous() if valid != rows else summed, bounds[rank])
    if acc_r is not None:
        acc_r = _reduce_sum(acc_r, root, group)
    gathered = _gather(out_slab, root, world, group)     # [world][rows][sW][3] on rank 0
    if rank != 0:
        return None, {}
    out = gathered.view(world * rows, sW, 3)[:sH]
    if acc_r is not None:
        debug["accumulated robustness"] = acc_r
    return out, debug
