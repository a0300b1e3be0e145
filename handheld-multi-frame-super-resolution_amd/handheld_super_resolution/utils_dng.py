"""Burst front end (reference utils_dng.py:50-164): sensor counts -> normalised, white-balanced
float32 RAW on the GPU.  DNG *decoding* needs rawpy + exifread like the reference; bursts that are already in
memory (or in an .npz file) as integer arrays + metadata take the same normalisation without them."""
import numpy as np
import torch

from . import _lib


def normalize_burst(raw, black_levels, white_level, white_balance, cfa_pattern, device=None):
    """uint16 counts [n, H, W] (or [H, W]) -> float32 GPU tensor of the same shape:
    (count - black[c]) / (white - black[c]) * wb[c] / wb[1] per CFA colour c, in the reference's float32
    arithmetic (utils_dng.py:149-160).  black_levels / white_balance: per colour (R, G, B; a 4th rawpy entry for
    the second green is ignored)."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    arr = raw if isinstance(raw, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(raw))
    if arr.dtype not in (torch.uint16, torch.int16):
        if arr.dtype.is_floating_point:
            raise TypeError("normalize_burst expects integer sensor counts (got %s): float data is taken as already "
                            "normalised" % arr.dtype)
        if int(arr.min()) < 0 or int(arr.max()) > 65535:
            raise ValueError("sensor counts outside the uint16 range")
        arr = arr.to(torch.int32).to(torch.uint16) if hasattr(torch, "uint16") else arr
    squeeze = arr.dim() == 2
    if squeeze:
        arr = arr[None]
    if arr.dim() != 3:
        raise ValueError("raw must be [H, W] or [n, H, W]")
    arr = arr.contiguous().to(dev, non_blocking=True)  # current stream; does not block the host for pinned memory
    n, H, W = arr.shape
    out = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    bl = [float(v) for v in list(black_levels)[:3]]
    wb = [float(v) for v in list(white_balance)[:3]]
    if len(bl) < 3 or len(wb) < 3:
        raise ValueError("black_levels and white_balance need one entry per colour (R, G, B)")
    _lib.call("hhsr_normalize_raw_u16", _lib.ptr(arr), n, H, W, W, _lib.cfa_bytes(cfa_pattern), _lib.doubles(bl),
              float(white_level), _lib.doubles(wb), _lib.ptr(out), _lib.stream())
    return out[0] if squeeze else out


def load_dng_burst(burst_path):
    """Folder of .dng files -> (ref_raw, raw_comp, ISO, tags, CFA, xyz2cam, white_balance, ref_path) like the
    reference (utils_dng.py:50-164), with the normalisation done on the GPU.  Needs rawpy + exifread."""
    try:
        import rawpy
        import exifread
    except ImportError as e:
        raise ImportError("reading .dng bursts needs rawpy and exifread (not installed); pass an in-memory burst "
                          "or an .npz file instead") from e
    import glob
    import os

    paths = sorted(glob.glob(os.path.join(str(burst_path), "*.dng")))
    if not paths:
        raise FileNotFoundError("At least one raw .dng file must be present in the burst folder.")
    frames = []
    for p in paths:
        with rawpy.imread(p) as ro:
            frames.append(ro.raw_image.copy())
    with rawpy.imread(paths[0]) as raw:
        white_level = int(raw.white_level)
        black_levels = list(raw.black_level_per_channel)
        white_balance = list(raw.camera_whitebalance)
        cfa = raw.raw_pattern.copy()
    cfa[cfa == 3] = 1
    with open(paths[0], "rb") as f:
        tags = exifread.process_file(f)
    if "EXIF ISOSpeedRatings" in tags:
        iso = int(str(tags["EXIF ISOSpeedRatings"]))
    elif "Image ISOSpeedRatings" in tags:
        iso = int(str(tags["Image ISOSpeedRatings"]))
    else:
        raise AttributeError("ISO value could not be found in both EXIF and Image type.")
    iso = min(3200, max(100, iso))
    xyz2cam = None
    if "Image Tag 0xC621" in tags:  # DNG ColorMatrix1 (raw2rgb.py:11-26)
        xyz2cam = np.array([x.decimal() for x in tags["Image Tag 0xC621"].values]).reshape(3, 3).astype(np.float32)
    stack = normalize_burst(np.stack(frames), black_levels, white_level, white_balance, cfa)
    return stack[0], stack[1:], iso, tags, cfa, xyz2cam, white_balance, paths[0]
