"""Frame-sharded merge over torch.distributed with world_size 2 on gloo (CPU): the sharding, the single
sum-reduce of the packed accumulators and the rank-0 finish are exercised with the oracle as the
per-rank engine, and must reproduce the sequential result."""
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

    def __init__(self, config):
        self.cfg = config
        self.cfa = np.array(config.exif.cfa_pattern)
        self.wb = np.array(config.exif.white_balance, dtype=np.float64)
        self.curves = (np.array(config.noise_model.std_curve), np.array(config.noise_model.diff_curve))

    def init_ref(self, ref):
        self.ref = np.asarray(ref, np.float32)
        grey = oracle.compute_grey_images(self.ref, "FFT")
        self.al = oracle.init_alignment(grey, self.cfg)
        self.stats = oracle.init_robustness(self.ref, self.cfa, self.wb, self.cfg)
        return self

    def output_shape(self):
        H, W = self.ref.shape
        s = self.cfg.scale
        return (round(s * H), round(s * W), 3)

    def partial(self, comps, world=1):
        H, W = self.ref.shape
        sH, sW, _ = self.output_shape()
        rows = hdist.slab_rows(sH, world) if world > 1 else sH
        acc = np.zeros((2, world * rows, sW, 3), np.float32)
        acc_r = np.zeros((H, W), np.float32)
        for img in comps:
            flow = oracle.align(*self.al, oracle.compute_grey_images(img, "FFT"), self.cfg)
            r = oracle.compute_robustness(img, *self.stats, flow, self.cfa, self.wb, self.curves, self.cfg)
            acc_r += r
            oracle.merge(img, flow, oracle.estimate_kernels(img, self.cfg), r, acc[0, :sH], acc[1, :sH], self.cfa, self.cfg)
        slabs = np.ascontiguousarray(acc.reshape(2, world, rows, sW, 3).transpose(1, 0, 2, 3, 4))
        return torch.from_numpy(slabs), torch.from_numpy(acc_r)

    def finish_slab(self, acc, row0, acc_r=None):
        sH, sW, _ = self.output_shape()
        a = acc.numpy()
        rows = a.shape[1]
        full = np.zeros((2, sH, sW, 3), np.float32)
        full[:, row0:row0 + rows] = a
        oracle.merge_ref(self.ref, oracle.estimate_kernels(self.ref, self.cfg), full[0], full[1], self.cfa, self.cfg)
        oracle.divide(full[0], full[1])
        return torch.from_numpy(np.ascontiguousarray(full[0, row0:row0 + rows]))


def _burst():
    ref, comp, _ = synth.make_burst(128, 128, 4, seed=9, max_shift=1.5)
    cfg = base_config(ts=16, scale=2)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    return ref, comp, cfg


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref, comp, cfg = _burst()
        out, dbg = hdist.main_sharded(ref, comp, cfg, engine=OracleEngine(cfg))
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
    assert b[0] == 0 and b[-1] == 6000 and all(x % 32 == 0 for x in b[:-1]) and b == sorted(b)
    assert hdist.slab_rows(6000, 8) == 768 and hdist.slab_rows(256, 3) == 96
    assert hdist.slab_bounds(100, 8) == [0, 32, 64, 96, 100, 100, 100, 100, 100]  # trailing slabs may be empty


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
    assert np.nanmax(d) < 1e-5  # only the float32 summation order differs
    np.testing.assert_allclose(got["acc_r"], dbg["accumulated robustness"], atol=1e-6)


def test_single_process_path_is_main():
    ref, comp, cfg = _burst()
    out, _ = hdist.main_sharded(ref, comp[:2], cfg, engine=OracleEngine(cfg))
    want, _ = oracle.main(ref, comp[:2], cfg)
    with np.errstate(all="ignore"):
        assert np.nanmax(np.abs(out.numpy() - want)) < 1e-6
