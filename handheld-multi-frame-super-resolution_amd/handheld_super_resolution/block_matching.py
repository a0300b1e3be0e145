"""Integer block matching per pyramid level (reference block_matching.py)."""
from . import _lib


def _check(ref_lvl, moving_lvl, alignment):
    assert ref_lvl.is_contiguous() and moving_lvl.is_contiguous() and alignment.is_contiguous()
    ny, nx, _ = alignment.shape
    return ny, nx


def align_lvl_block_matching_L2(ref_lvl, ref_fft_lvl, moving_lvl, alignment, l, config):
    """L2 tile search (block_matching.py:20-76).  The reference correlates zero-padded reference tiles
    with gathered moving windows by FFT and adds a box-filtered window energy; the HIP kernel stages
    the same clamp-to-edge window in LDS and evaluates the (2r+1)^2 SSDs directly (identical argmin up
    to float32 near-ties).  `ref_lvl` is the reference pyramid level (the reference passes its tiled
    copy here); `ref_fft_lvl` is accepted for signature parity and ignored."""
    ts = config.block_matching.tuning.tile_sizes[l]
    r = config.block_matching.tuning.search_radii[l]
    if ts not in (8, 16, 32, 64):
        raise NotImplementedError("Box filter for tile size {} not implemented".format(ts))
    ny, nx = _check(ref_lvl, moving_lvl, alignment)
    mh, mw = moving_lvl.shape
    _lib.call("hhsr_bm_l2", _lib.ptr(ref_lvl), ref_lvl.shape[1], _lib.ptr(moving_lvl), mh, mw, mw,
              _lib.ptr(alignment), ny, nx, ts, r, _lib.stream())


def align_lvl_block_matching_L1(ref_lvl, moving_lvl, alignments, l, config, effective=False):
    """L1 tile search (block_matching.py:78-345).  The upstream kernels are undefined behaviour
    (uninitialised shifts + write race); this implements their INTENDED semantics
    (SAD, zero outside, flow <- round(flow) + shift).  effective=True (`metrics: L1_ref_effective`)
    gives the most likely on-hardware outcome of the reference instead: flow <- round(flow)."""
    ts = config.block_matching.tuning.tile_sizes[l]
    r = config.block_matching.tuning.search_radii[l]
    if ts == 8 or ts not in (16, 32, 64):
        raise NotImplementedError("L1 local search kernel for tile size {} not implemented".format(ts))
    if ts == 16:
        assert 2 * r + 16 <= 32, "L1 local search kernel only implemented for search windows up to size 32"
    if ts == 64:
        assert 2 * r <= 16, f"Cant handle search radius {r} with tile size {ts} in L1 local search kernel."
    ny, nx = _check(ref_lvl, moving_lvl, alignments)
    mh, mw = moving_lvl.shape
    _lib.call("hhsr_bm_l1", _lib.ptr(ref_lvl), ref_lvl.shape[1], _lib.ptr(moving_lvl), mh, mw, mw,
              _lib.ptr(alignments), ny, nx, ts, r, 1 if effective else 0, _lib.stream())
