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
        acc_r = np.zeros((Hs, W), np.float32)
        for i, img in enumerate(comps):
            img_s = np.asarray(img, np.float32)[S0:S1]
            flow_s = flows[i, t0:t1].numpy()
            r = oracle.compute_robustness(img_s, *stats, flow_s, self.cfa, self.wb, self.curves, cfg)
            acc_r += r
            oracle.merge(img_s, flow_s, oracle.estimate_kernels(img_s, cfg), r, num, den, self.cfa, cfg)
        oracle.merge_ref(ref_s, oracle.estimate_kernels(ref_s, cfg), num, den, self.cfa, cfg)
        oracle.divide(num, den)
        L0 = int(math.ceil(r0 / s)) - S0
        L1 = min(Hs, int(math.ceil(r1 / s)) - S0)
        return torch.from_numpy(np.ascontiguousarray(num[row0:row0 + (r1 - r0)])), torch.from_numpy(acc_r[L0:L1].copy())


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
    assert b[0] == 0 and b[-1] == 6000 and all(x % 96 == 0 for x in b[:-1]) and b == sorted(b)
    assert hdist.slab_rows(6000, 8) == 768 and hdist.slab_rows(256, 3) == 96
    assert hdist.slab_bounds(100, 8) == [0, 96, 100, 100, 100, 100, 100, 100, 100]  # trailing slabs may be empty


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
