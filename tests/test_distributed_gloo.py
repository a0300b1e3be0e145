"""Multi-GPU path over torch.distributed with world_size 2 and 3 on gloo (CPU): frame-parallel alignment, the single
all-gather of the flow fields, row-parallel merge on sub-images and the gather of the finished slabs are exercised
with the NumPy oracle as the per-rank engine, and must reproduce the sequential result — which also checks the
sub-image halo rule (distributed.sub_image_rows) against the reference algorithm itself."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from helpers import base_config
from handheld_super_resolution import distributed as hdist
from handheld_super_resolution import synthetic as synth


class OracleEngine:
    """Test double for distributed.HipEngine with the same methods (NumPy oracle as the compute)."""

    denoiser_on = False
    accumulate_r = True

    def __init__(self, config):
        self.cfg = config
        self.cfa = np.array(config.exif.cfa_pattern)
        self.wb = np.array(config.exif.white_balance, dtype=np.float64)
        self.curves = (np.array(config.noise_model.std_curve), np.array(config.noise_model.diff_curve))

    def single(self, ref, comps):
        out, dbg = oracle.main(ref, comps, self.cfg)
        return torch.from_numpy(out), {k: (torch.from_numpy(np.asarray(v, np.float32)) if k == "accumulated robustness" else v)
                                       for k, v in dbg.items()}

    def init_ref(self, ref):
        self.ref = np.asarray(ref, np.float32)
        grey = oracle.compute_grey_images(self.ref, "FFT")
        self.al = oracle.init_alignment(grey, self.cfg)
        return self

    def shape(self):
        return self.ref.shape

    def output_shape(self):
        H, W = self.ref.shape
        s = self.cfg.scale
        return (round(s * H), round(s * W), 3)

    def tile_size(self):
        return int(self.cfg.block_matching.tuning.tile_size)

    def align_frames(self, comps):
        ts = self.tile_size()
        H, W = self.ref.shape
        fl = [oracle.align(*self.al, oracle.compute_grey_images(img, "FFT"), self.cfg) for img in comps]
        if not fl:
            return torch.empty((0, -(-H // ts), -(-W // ts), 2), dtype=torch.float32)
        return torch.from_numpy(np.stack(fl))

    def merge_rows(self, comps, flows, r0, r1, max_flow_y):
        cfg, ts = self.cfg, self.tile_size()
        H, W = self.ref.shape
        s = cfg.scale
        S0, S1, row0 = hdist.sub_image_rows(r0, r1, s, H, ts, max_flow_y)
        Hs = S1 - S0
        t0, t1 = S0 // ts, -(-S1 // ts)
        ref_s = self.ref[S0:S1]
        stats = oracle.init_robustness(ref_s, self.cfa, self.wb, cfg)
        num = np.zeros((round(s * Hs), round(s * W), 3), np.float32)
        den = np.zeros_like(num)
        acc_r = np.zeros((Hs, W), np.float64 if self.denoiser_on else np.float32)  # (the denoiser decides on a float64 sum)
        t = cfg.robustness.tuning
        for i, img in enumerate(comps):
            img_s = np.asarray(img, np.float32)[S0:S1]
            flow_s = flows[i, t0:t1].numpy()
            # the flow-irregularity weight is the one stage that is not row-local: full field, then the sub-image's rows
            S = oracle.compute_s(flows[i].numpy(), t.Mt, t.s1, t.s2)[t0:t1]
            r = oracle.compute_robustness(img_s, *stats, flow_s, self.cfa, self.wb, self.curves, cfg, S=S)
            acc_r += r
            oracle.merge(img_s, flow_s, oracle.estimate_kernels(img_s, cfg), r, num, den, self.cfa, cfg)
        oracle.merge_ref(ref_s, oracle.estimate_kernels(ref_s, cfg), num, den, self.cfa, cfg,
                         acc_rob=acc_r if self.denoiser_on else None)
        oracle.divide(num, den)
        L0 = int(math.ceil(r0 / s)) - S0
        L1 = min(Hs, int(math.ceil(r1 / s)) - S0)
        return (torch.from_numpy(np.ascontiguousarray(num[row0:row0 + (r1 - r0)])),
                torch.from_numpy(acc_r[L0:L1].astype(np.float32)))


    # ---- strategy "reduce" --------------------------------------------------------------------------------------------
    def partial(self, ref, my_frames, bounds, rows):
        cfg = self.cfg
        self.init_ref(ref)
        H, W = self.ref.shape
        sH, sW, _ = self.output_shape()
        stats = oracle.init_robustness(self.ref, self.cfa, self.wb, cfg)
        num = np.zeros((sH, sW, 3), np.float32)
        den = np.zeros_like(num)
        acc_r = np.zeros((H, W), np.float64 if self.denoiser_on else np.float32)  # (HipEngine.partial: float64 partial sums)
        for img in my_frames:
            img = np.asarray(img, np.float32)
            flow = oracle.align(*self.al, oracle.compute_grey_images(img, "FFT"), cfg)
            r = oracle.compute_robustness(img, *stats, flow, self.cfa, self.wb, self.curves, cfg)
            acc_r += r
            oracle.merge(img, flow, oracle.estimate_kernels(img, cfg), r, num, den, self.cfa, cfg)
        world = len(bounds) - 1
        acc = np.zeros((world, 2, rows, sW, 3), np.float32)  # chunk j = num / den of slab j (one reduce-scatter)
        for j in range(world):
            b0, b1 = bounds[j], bounds[j + 1]
            acc[j, 0, : b1 - b0], acc[j, 1, : b1 - b0] = num[b0:b1], den[b0:b1]
        return torch.from_numpy(acc), torch.from_numpy(acc_r), self.ref, oracle.estimate_kernels(self.ref, cfg)

    def finish_rows(self, acc_slab, r0, r1, ref, ref_covs, acc_r=None):
        cfg = self.cfg
        sH, sW, _ = self.output_shape()
        num = np.zeros((sH, sW, 3), np.float32)
        den = np.zeros_like(num)
        num[r0:r1], den[r0:r1] = acc_slab[0, : r1 - r0].numpy(), acc_slab[1, : r1 - r0].numpy()
        assert (acc_r is not None) == self.denoiser_on
        oracle.merge_ref(ref, ref_covs, num, den, self.cfa, cfg,  # (whole image: only rows [r0, r1) are kept)
                         acc_rob=acc_r.numpy() if self.denoiser_on else None)
        oracle.divide(num, den)
        return torch.from_numpy(np.ascontiguousarray(num[r0:r1]))


def _burst(seam=False, denoiser=False):
    ref, comp, _ = synth.make_burst(128, 128, 4, seed=9, max_shift=0.0 if seam else 1.5, occluder=denoiser)
    if seam:  # frame 1 is brighter around raw rows 80-100: its robustness there sits in the band where S decides
        comp = comp.copy()
        yy = np.arange(128, dtype=np.float32)[:, None]
        comp[1] = np.clip(comp[1] + 0.06 * np.exp(-(((yy - 90) / 8.0) ** 2)), 0, 1).astype(np.float32)
    cfg = base_config(ts=16, scale=2)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    if denoiser:  # merge.py:223-228 with both branches alive: 3 comp frames, "every frame fully accepted" adds, the rest overwrites
        cfg.accumulated_robustness_denoiser.enabled = True
        cfg.accumulated_robustness_denoiser.merge.enabled = True
        cfg.accumulated_robustness_denoiser.merge.max_frame_count = 3
    return ref, comp, cfg


class DenoiserEngine(OracleEngine):
    denoiser_on = True


class SeamEngine(OracleEngine):
    """OracleEngine whose "alignment" returns a crafted flow field: zero everywhere except 3 px in y in tile row 4 of
    frame 1 — irregular flow directly ABOVE the sub-image of slab 1 (world 2: output rows [192, 256) -> raw rows from
    96 - (3 + HALO) -> tile row 5 on).  The flow-irregularity weight S of tile row 5 depends on tile row 4
    (robustness.py:587-612), which a row slice of the flow field does not contain."""

    def align_frames(self, comps):
        H, W = self.ref.shape
        ts = self.tile_size()
        fl = np.zeros((len(comps), -(-H // ts), -(-W // ts), 2), np.float32)
        if len(comps) and getattr(self, "_has_frame1", True):
            fl[self._frame1, 4, :, 1] = 3.0
        return torch.from_numpy(fl)


def _worker(rank, world, port, out_path, strategy="rows", seam=False, n_comp=3, hip=None, denoiser=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref, comp, cfg = _burst(seam, denoiser)
        comp = comp[:n_comp]
        if hip:
            cfg.hip = hip
        eng = DenoiserEngine(cfg) if denoiser else OracleEngine(cfg)
        if seam:  # frame 1 of the burst is aligned by rank 1 % world as its (1 // world)-th frame
            eng = SeamEngine(cfg)
            eng._has_frame1, eng._frame1 = (1 % world == rank), 1 // world
        out, dbg = hdist.main_sharded(ref, comp, cfg, engine=eng, strategy=strategy)
        if rank == 0:
            np.savez(out_path, out=out.numpy(), acc_r=dbg["accumulated robustness"].numpy())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_slab_bounds():
    b = hdist.slab_bounds(6000, 8)
    assert b[0] == 0 and b[-1] == 6000 and all(x % 96 == 0 for x in b[:-1]) and b == sorted(b)
    assert hdist.slab_rows(6000, 8) == 768 and hdist.slab_rows(256, 3) == 96
    assert hdist.slab_bounds(100, 8) == [0, 96, 100, 100, 100, 100, 100, 100, 100]  # trailing slabs may be empty


def test_uneven_slabs_and_stages():
    """Strategy "rows": ranks that align one frame fewer take more rows; stages are whole rounds of >= 4 frames."""
    b = hdist.slab_bounds(6000, 8, 19, 0.9)
    assert b[0] == 0 and b[-1] == 6000 and all(x % 96 == 0 for x in b[:-1]) and b == sorted(b)
    sizes = [b1 - b0 for b0, b1 in zip(b[:-1], b[1:])]
    assert max(sizes[:3]) < min(sizes[3:])          # ranks 0-2 align 3 frames, ranks 3-7 two
    # the modelled per-rank cost a_j rho sH + rows_j n is level to within one slab-alignment step
    cost = [len(range(j, 19, 8)) * 0.9 * 6000 + sizes[j] * 19 for j in range(8)]
    assert max(cost) - min(cost) <= 2 * 96 * 19
    # the scale's own tile grid (32 output rows at x2, 48 at x3, 96 at fractional scales) levels the ranks further
    assert (hdist.slab_align(2), hdist.slab_align(3), hdist.slab_align(1.5), hdist.slab_align(1)) == (32, 48, 96, 16)
    for scale, sH in ((2, 6000), (3, 18000)):
        rho, al = hdist.align_cost(scale), hdist.slab_align(scale)
        bf = hdist.slab_bounds(sH, 8, 19, rho, al)
        assert bf[0] == 0 and bf[-1] == sH and all(x % al == 0 for x in bf[:-1]) and bf == sorted(bf)
        sz = [b1 - b0 for b0, b1 in zip(bf[:-1], bf[1:])]
        cost = [len(range(j, 19, 8)) * rho * sH + sz[j] * 19 for j in range(8)]
        assert max(cost) - min(cost) <= 2 * al * 19 and max(sz[:3]) < min(sz[3:])
    assert hdist.slab_bounds(6000, 8, 19, 0.0) == hdist.slab_bounds(6000, 8)
    assert hdist.slab_bounds(6000, 1, 19, 0.9) == [0, 6000]
    assert hdist.stage_plan(19, 8) == [(0, 1), (1, 1), (2, 1)]
    assert hdist.stage_plan(19, 2) == [(0, 2), (2, 2), (4, 2), (6, 2), (8, 2)]
    assert hdist.stage_plan(19, 8, 100) == [(0, 3)] and hdist.stage_plan(0, 4) == []
    assert hdist.stage_frames((2, 1), 19, 8) == [16, 17, 18] and hdist.stage_frames((0, 2), 19, 2, 1) == [1, 3]
    for n, g in ((19, 8), (19, 3), (3, 8), (7, 2)):     # every frame in exactly one stage, aligned by exactly one rank
        st = hdist.stage_plan(n, g)
        assert sorted(i for s_ in st for i in hdist.stage_frames(s_, n, g)) == list(range(n))
        assert sorted(i for s_ in st for r in range(g) for i in hdist.stage_frames(s_, n, g, r)) == list(range(n))


def test_shard_indices():
    assert hdist.shard_indices(19, 0, 8) == [0, 8, 16]
    assert hdist.shard_indices(19, 7, 8) == [7, 15]
    assert sorted(sum((hdist.shard_indices(19, r, 8) for r in range(8)), [])) == list(range(19))
    assert hdist.shard_indices(3, 5, 8) == []  # more ranks than frames: empty shard contributes zeros


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_sequential(tmp_path, world):
    """world 3: uneven slabs (256 output rows -> 96 + 96 + 64) and uneven frame shards (3 frames)."""
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = np.load(out_path)
    ref, comp, cfg = _burst()
    want, dbg = oracle.main(ref, comp, cfg)
    with np.errstate(all="ignore"):
        d = np.abs(got["out"] - want)
    assert (np.isnan(got["out"]) == np.isnan(want)).all()
    # every pixel a slab depends on lies >= 8 rows inside its sub-image: the row-sharded result is the sequential one,
    # bit for bit (no partial sums are exchanged, so there is no summation-order difference either)
    assert np.nanmax(d) == 0.0
    assert np.array_equal(got["acc_r"], dbg["accumulated robustness"].astype(np.float32))


@pytest.mark.timeout(300)
def test_staged_gathers_equal_sequential(tmp_path):
    """config.hip.stage_frames = 1: every round is its own stage (two all-gathers for 3 frames on 2 ranks, the second
    one with a rank that has no frame left) — same bits as one gather of everything."""
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), out_path, "rows", False, 3, {"stage_frames": 1, "align_cost": 0.0}),
             nprocs=2, join=True)
    got = np.load(out_path)
    ref, comp, cfg = _burst()
    want, dbg = oracle.main(ref, comp, cfg)
    assert np.array_equal(got["out"], want, equal_nan=True)
    assert np.array_equal(got["acc_r"], dbg["accumulated robustness"].astype(np.float32))


@pytest.mark.timeout(300)
def test_moving_object_at_seam(tmp_path):
    """S of a tile is the flow spread over its 3 x 3 TILE neighbourhood — not row-local.  A burst
    whose flow is irregular directly above a sub-image must still reproduce the sequential result bit for bit — and the
    case must be real: with S recomputed from the row slice (round 2) the slab differs."""
    ref, comp, cfg = _burst(seam=True)
    flows = np.zeros((3, 8, 8, 2), np.float32)
    flows[1, 4, :, 1] = 3.0
    # sequential result with the same injected flows (oracle stages as in oracle.main)
    cfa, wb = np.array(cfg.exif.cfa_pattern), np.array(cfg.exif.white_balance, np.float64)
    curves = (np.array(cfg.noise_model.std_curve), np.array(cfg.noise_model.diff_curve))

    def sequential(slice_rows=None):
        r0s = 0 if slice_rows is None else slice_rows[0] * 16
        r1s = 128 if slice_rows is None else min(128, slice_rows[1] * 16)
        rs, imgs = ref[r0s:r1s], comp[:, r0s:r1s]
        stats = oracle.init_robustness(rs, cfa, wb, cfg)
        num = np.zeros((2 * (r1s - r0s), 256, 3), np.float32)
        den = np.zeros_like(num)
        for i in range(3):
            f = flows[i] if slice_rows is None else flows[i, slice_rows[0]:slice_rows[1]]
            r = oracle.compute_robustness(imgs[i], *stats, f, cfa, wb, curves, cfg)
            oracle.merge(imgs[i], f, oracle.estimate_kernels(imgs[i], cfg), r, num, den, cfa, cfg)
        oracle.merge_ref(rs, oracle.estimate_kernels(rs, cfg), num, den, cfa, cfg)
        oracle.divide(num, den)
        return num

    want = sequential()
    S0, S1, row0 = hdist.sub_image_rows(192, 256, 2, 128, 16, 3.0)
    assert S0 == 80 and S1 == 128 and row0 == 32
    naive = sequential((5, 8))[row0:row0 + 64]  # round 2's sub-image: S from the slice
    with np.errstate(all="ignore"):
        assert np.nanmax(np.abs(naive - want[192:256])) > 1e-4, "the burst does not exercise the seam case"
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), out_path, "rows", True), nprocs=2, join=True)
    got = np.load(out_path)
    assert np.array_equal(got["out"], want, equal_nan=True)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_reduce_strategy_matches_sequential(tmp_path, world):
    """strategy="reduce" (north star: frames sharded one per rank, ONE reduce of the float32 accumulators, reference
    merge.py:432-434): partial sums are added in a different order than the sequential merge, so the result agrees to
    float32 summation noise instead of bit for bit."""
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), out_path, "reduce"), nprocs=world, join=True)
    got = np.load(out_path)
    ref, comp, cfg = _burst()
    want, dbg = oracle.main(ref, comp, cfg)
    assert (np.isnan(got["out"]) == np.isnan(want)).all()
    with np.errstate(all="ignore"):
        d = np.abs(got["out"] - want)
    assert np.nanmax(d) < 2e-6  # measured 2.4e-7
    np.testing.assert_allclose(got["acc_r"], dbg["accumulated robustness"].astype(np.float32), rtol=0, atol=2e-6)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("strategy", ["rows", "reduce"])
def test_denoiser_sharded(tmp_path, strategy):
    """The accumulated-robustness denoiser of the reference frame's merge (merge.py:223-228) across ranks: "rows" sums the
    robustness of ALL frames on the rank's sub-image; "reduce" all-reduces the ranks' float64 partial sums before the slab is
    finished (robustness.RobustnessSum: the decision `acc_rob < max_frame_count` is taken on the float64 sum).  Both
    branches of the decision must occur in the burst."""
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), out_path, strategy, False, 3, None, True), nprocs=2, join=True)
    got = np.load(out_path)
    ref, comp, cfg = _burst(denoiser=True)
    want, dbg = oracle.main(ref, comp, cfg)
    acc = dbg["accumulated robustness"]
    assert 0.02 < float((acc < 3).mean()) < 0.98, "the burst does not exercise both branches of the decision"
    assert (np.isnan(got["out"]) == np.isnan(want)).all()
    with np.errstate(all="ignore"):
        d = np.nanmax(np.abs(got["out"] - want))
    assert d == 0.0 if strategy == "rows" else d < 2e-6
    np.testing.assert_allclose(got["acc_r"], np.asarray(acc, np.float32), rtol=0, atol=0 if strategy == "rows" else 2e-6)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("strategy", ["rows", "reduce"])
def test_more_ranks_than_frames(tmp_path, strategy):
    """3 ranks, 2 comp frames: one rank aligns / merges nothing (zero contribution to the all-gather / reduce-scatter)."""
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(3, _free_port(), out_path, strategy, False, 2), nprocs=3, join=True)
    got = np.load(out_path)
    ref, comp, cfg = _burst()
    want, _ = oracle.main(ref, comp[:2], cfg)
    assert (np.isnan(got["out"]) == np.isnan(want)).all()
    with np.errstate(all="ignore"):
        d = np.nanmax(np.abs(got["out"] - want))
    assert d == 0.0 if strategy == "rows" else d < 2e-6


def _burst20():
    """20 frames: 19 comp frames on 8 ranks = the frame distribution of BASELINE.json's C3 / C5 (3, 3, 3, 2, 2, 2, 2, 2)."""
    ref, comp, _ = synth.make_burst(192, 128, 20, seed=13, max_shift=1.5)
    cfg = base_config(ts=16, scale=2)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    return ref, comp, cfg


class FastOracleEngine(OracleEngine):
    """OracleEngine with the C form of the accumulation (oracle.cfast: bit-identical to oracle/merge.py on what
    tests/test_oracle_kat.py compares) — 19 frames x 8 ranks of the NumPy merge would take minutes."""

    def merge_rows(self, comps, flows, r0, r1, max_flow_y):
        from oracle import cfast

        keep = oracle.merge, oracle.merge_ref
        oracle.merge, oracle.merge_ref = cfast.merge, cfast.merge_ref
        try:
            return super().merge_rows(comps, flows, r0, r1, max_flow_y)
        finally:
            oracle.merge, oracle.merge_ref = keep

    def partial(self, ref, my_frames, bounds, rows):
        from oracle import cfast

        keep = oracle.merge, oracle.merge_ref
        oracle.merge, oracle.merge_ref = cfast.merge, cfast.merge_ref
        try:
            return super().partial(ref, my_frames, bounds, rows)
        finally:
            oracle.merge, oracle.merge_ref = keep

    def finish_rows(self, acc_slab, r0, r1, ref, ref_covs, acc_r=None):
        from oracle import cfast

        keep = oracle.merge, oracle.merge_ref
        oracle.merge, oracle.merge_ref = cfast.merge, cfast.merge_ref
        try:
            return super().finish_rows(acc_slab, r0, r1, ref, ref_covs, acc_r)
        finally:
            oracle.merge, oracle.merge_ref = keep


def _worker8(rank, world, port, out_path, strategy):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref, comp, cfg = _burst20()
        eng = FastOracleEngine(cfg)
        assert len(hdist.shard_indices(len(comp), rank, world)) == (3 if rank < 3 else 2)
        out, dbg = hdist.main_sharded(ref, comp, cfg, engine=eng, strategy=strategy)
        if rank == 0:
            np.savez(out_path, out=out.numpy(), acc_r=dbg["accumulated robustness"].numpy())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("strategy", ["rows", "reduce"])
def test_world_8_nineteen_frames(tmp_path, strategy):
    """VERDICT r4 #2b: the world size the target names, EXECUTED — 8 ranks, 19 comp frames (3, 3, 3, 2, 2, 2, 2, 2), uneven
    slabs of the x2 tile grid, both strategies, against the sequential result: "rows" bit for bit, "reduce" to float32
    summation order."""
    ref, comp, cfg = _burst20()
    sH = 2 * ref.shape[0]
    b = hdist.slab_bounds(sH, 8, len(comp), hdist.align_cost(2), hdist.slab_align(2))
    sizes = [b1 - b0 for b0, b1 in zip(b[:-1], b[1:])]
    assert b[-1] == sH and min(sizes) >= 32 and max(sizes[:3]) <= min(sizes[3:]) and len(set(sizes)) > 1  # uneven, none empty
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker8, args=(8, _free_port(), out_path, strategy), nprocs=8, join=True)
    got = np.load(out_path)
    want, dbg = oracle.main(ref, comp, cfg, fast=True)
    assert (np.isnan(got["out"]) == np.isnan(want)).all()
    with np.errstate(all="ignore"):
        d = np.nanmax(np.abs(got["out"] - want))
    assert d == 0.0 if strategy == "rows" else d < 2e-6
    # (the engines sum the robustness maps in float32 like the product, D15; the sequential oracle in float64: 19 terms)
    np.testing.assert_allclose(got["acc_r"], dbg["accumulated robustness"].astype(np.float32), rtol=0, atol=4e-6)


def test_single_process_path_is_main():
    ref, comp, cfg = _burst()
    out, _ = hdist.main_sharded(ref, comp[:2], cfg, engine=OracleEngine(cfg))
    want, _ = oracle.main(ref, comp[:2], cfg)
    assert np.array_equal(out.numpy(), want, equal_nan=True)


def test_sub_image_rows():
    # 12 MP x2, 8 ranks: slab 3 = output rows [2304, 3072) -> raw rows [1152, 1536) +- (4 + HALO), tile aligned, even
    h = 4 + hdist.HALO
    S0, S1, row0 = hdist.sub_image_rows(2304, 3072, 2, 3000, 16, 3.2)
    assert S0 == ((1152 - h) // 16) * 16 and S1 == 1536 + h + ((1536 + h) & 1) and row0 == 2304 - 2 * S0
    assert S0 % 16 == 0 and S1 % 2 == 0 and row0 % 32 == 0
    assert hdist.sub_image_rows(0, 768, 2, 3000, 16, 3.2)[:2] == (0, 384 + h)      # clipped at the top
    assert hdist.sub_image_rows(5376, 6000, 2, 3000, 16, 3.2)[1] == 3000           # ... and at the bottom
    # non-integer scale: S0 * scale must be an integer output row
    S0, S1, row0 = hdist.sub_image_rows(300, 450, 1.5, 400, 16, 2.0)
    assert S0 % 16 == 0 and (S0 * 1.5).is_integer() and row0 == 300 - int(S0 * 1.5)


@pytest.mark.timeout(600)
def test_bench_launches_ranks_itself():
    """`python bench.py --gpus 2` run by hand re-executes itself under torch.distributed.run: two ranks rendezvous
    (gloo here; RCCL on the GPU box), go through main_sharded and rank 0 prints ONE JSON line with n_gpus = 2.  The
    per-rank engine is this file's NumPy test double — launch plumbing only, not a measurement."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--engine",
                        os.path.abspath(__file__) + ":OracleEngine", "--height", "512", "--width", "512", "--frames", "3",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=550)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["steps"] == 1 and rec["value"] > 0
    assert rec["scaling"] == "strong" and "sharded by rows" in rec["config"]["parallelism"]
    # both strategies in the one record (the driver passes --gpus N only), the launcher's and the group's rank counts agree
    assert rec["ranks_agree"] and set(rec["strategies"]) == {"rows", "reduce"} and not rec.get("errors")
    assert rec["strategies"]["rows"]["headline"] and rec["strategies"]["reduce"]["value"] > 0
