"""Frame-sharded multi-GPU merge.  The reference is single-GPU; this is the one
parallelism the build adds.

Every comp frame's contribution to num/den depends only on the reference-frame state and that frame
(reference super_resolution.py:133-173) and contributions are summed (merge.py:432-434), so:
  * one process per GPU, rank k takes comp frames k, k+G, k+2G, ... (no data-path collective);
  * every rank replicates the cheap reference-frame precompute;
  * ONE sum-reduce to rank 0 of the packed float32 buffer [2, sH, sW, 3] (num | den) — RCCL over xGMI
    with backend "nccl" (+ the [H, W] accumulated robustness when the mask is requested);
  * rank 0 adds the reference frame (Alg. 11, which may overwrite rather than add) and normalises.
The engine that does the per-rank compute is injected so that the sharding / reduction logic can be
exercised without a GPU (tests run it with world_size 2 on gloo).
"""
import torch
import torch.distributed as dist


def shard_indices(n_frames, rank, world):
    """Indices of the comp frames rank `rank` of `world` processes (round-robin: balanced to +-1)."""
    return list(range(rank, n_frames, world))


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

    def partial(self, comp_imgs):
        """Packed accumulators [2, sH, sW, 3] of this rank's frames (+ accumulated robustness or None)."""
        from .merge import merge_burst, can_fuse_acc_r

        pipe = self.pipe
        sH, sW = pipe.output_size()
        acc = torch.empty((2, sH, sW, 3), dtype=torch.float32, device=pipe.device)
        acc_r = torch.zeros(tuple(pipe.ref.shape), dtype=torch.float32, device=pipe.device) if self.accumulate_r else None
        fuse_acc = acc_r is not None and can_fuse_acc_r(self.config) and len(comp_imgs) > 0
        frames = pipe.process_frames(list(comp_imgs), None if fuse_acc else acc_r)
        if frames:
            merge_burst(frames, None, None, acc[0], acc[1], pipe.cfa, self.config, do_ref=False, divide=False,
                        store_den=True, acc_r=acc_r if fuse_acc else None)
        else:
            acc.zero_()
        return acc, acc_r

    def finish(self, acc, acc_r):
        """Rank 0: reference-frame merge + normalisation on the reduced accumulators."""
        from .kernels import estimate_kernels
        from .merge import merge_burst, merge_ref
        from .utils import divide

        pipe = self.pipe
        ref_covs = estimate_kernels(pipe.ref, self.config)
        if self.denoiser_on:
            merge_ref(pipe.ref, ref_covs, acc[0], acc[1], pipe.cfa, self.config, acc_r)
            divide(acc[0], acc[1])
        else:
            merge_burst([], pipe.ref, ref_covs, acc[0], acc[1], pipe.cfa, self.config, load_acc=True, do_ref=True,
                        divide=True)
        return acc[0]


def _reduce_sum(t, dst, group):
    """Sum-reduce to `dst`.  RCCL ("nccl") reduces device tensors in place over xGMI; a host-only backend
    (gloo, used by the CPU tests) gets a staged copy."""
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return t
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return t


def main_sharded(ref_img, comp_imgs, config, group=None, engine=None):
    """Frame-sharded equivalent of main().  Returns (output, debug_dict) on rank 0 and (None, {}) on
    the other ranks.  Works un-initialised / with world_size 1 (then it is main() without collectives)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    eng = engine if engine is not None else HipEngine(config)
    eng.init_ref(ref_img)
    mine = [comp_imgs[i] for i in shard_indices(len(comp_imgs), rank, world)]
    acc, acc_r = eng.partial(mine)
    if world > 1:
        dst = dist.get_global_rank(group, 0) if group is not None else 0
        acc = _reduce_sum(acc, dst, group)
        if acc_r is not None:
            acc_r = _reduce_sum(acc_r, dst, group)
    if rank != 0:
        return None, {}
    out = eng.finish(acc, acc_r)
    debug = {"robustness": [], "flow": []}
    if acc_r is not None:
        debug["accumulated robustness"] = acc_r
    return out, debug
