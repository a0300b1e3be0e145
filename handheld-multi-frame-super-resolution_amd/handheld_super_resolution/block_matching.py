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
    to float32 near-ties).  `ref_lvl` is the reference pyramid level [H_l, W_l] (what init_alignment of this
    package keeps) or the reference's tiled, zero-padded copy [ny, nx, ts + 2r, ts + 2r] (what upstream callers
    pass: it is un-tiled on the fly); anything else raises TypeError.  `ref_fft_lvl` is accepted for signature
    parity and ignored."""
    ts = config.block_matching.tuning.tile_sizes[l]
    r = config.block_matching.tuning.search_radii[l]
    if ts not in (8, 16, 32, 64):
        raise NotImplementedError("Box filter for tile size {} not implemented".format(ts))
    if ref_lvl.dim() == 4:
        # a reference caller's `tyled_pyr_lvl`: [ny, nx, ts + 2r, ts + 2r], every tile zero-padded by r on each side
        # (alignment.py:56-60 upstream) -> back to the level the kernel reads, [ny ts, nx ts]
        ny4, nx4, p4, q4 = ref_lvl.shape
        if p4 != ts + 2 * r or q4 != ts + 2 * r or (ny4, nx4) != tuple(alignment.shape[:2]):
            raise TypeError(f"tiled reference level of shape {tuple(ref_lvl.shape)} does not match tile size {ts}, "
                            f"search radius {r} and a flow field of shape {tuple(alignment.shape)}")
        ref_lvl = ref_lvl[:, :, r:r + ts, r:r + ts].permute(0, 2, 1, 3).reshape(ny4 * ts, nx4 * ts).contiguous()
    elif ref_lvl.dim() != 2:
        raise TypeError(f"reference level must be [H, W] or the tiled [ny, nx, ts + 2r, ts + 2r], got {tuple(ref_lvl.shape)}")
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
