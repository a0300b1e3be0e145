"""ctypes binding of libhhsr_hip.so (C ABI declared in include/hhsr.h).

There is no CPU fallback: if the library is missing or a launch fails, a RuntimeError is raised.
torch tensors are passed as raw device pointers; every call enqueues on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HHSR_LIB") or os.path.join(_HERE, "libhhsr_hip.so")  # HHSR_LIB: A/B builds of the library

P, I, L, D, F = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float
U8P = C.POINTER(C.c_uint8)
DP = C.POINTER(C.c_double)
FP = C.POINTER(C.c_float)
PP = C.POINTER(C.c_void_p)

# name -> argtypes (all return int)
_SIGS = {
    "hhsr_lowpass_mask_c2c": [P, I, I, L, L, P],
    "hhsr_lowpass_mask_r2c": [P, I, I, L, L, P],
    "hhsr_grey_plan_create": [I, I, I, PP],
    "hhsr_grey_lowpass": [P, P, P, P],
    "hhsr_grey_lowpass_batch": [P, PP, PP, I, P],
    "hhsr_grey_plan_destroy": [P],
    "hhsr_pad_circular": [P, I, I, I, P, I, I, I, P],
    "hhsr_gauss_decimate": [P, I, I, I, P, I, I, FP, I, P],
    "hhsr_gauss_decimate_batch": [PP, I, I, I, I, PP, I, I, FP, I, P],
    "hhsr_grad_hessian": [P, I, I, I, I, P, P, P, P],
    "hhsr_bm_l2": [P, I, P, I, I, I, P, I, I, I, I, P],
    "hhsr_bm_l1": [P, I, P, I, I, I, P, I, I, I, I, I, P],
    "hhsr_ica": [P, P, P, I, P, P, I, I, I, P, I, I, I, I, I, P],
    "hhsr_align_level": [P, I, I, I, P, P, I, I, I, P, I, I, I, I, I, I, P, I, I, I, F, P],
    "hhsr_align_level_batch": [P, I, I, I, P, PP, I, I, I, I, PP, I, I, I, I, I, I, PP, I, I, I, F, P],
    "hhsr_flow_upscale_nearest": [P, I, I, P, I, I, I, F, P],
    "hhsr_cov_from_raw": [P, I, I, I, P, D, D, D, D, D, D, D, D, I, P],
    "hhsr_rob_stats": [P, I, I, I, U8P, DP, P, P, P],
    "hhsr_frame_stats": [P, I, I, I, U8P, DP, P, P, P, D, D, D, D, D, D, D, D, I, P],
    "hhsr_frame_stats_batch": [PP, I, I, I, I, U8P, DP, PP, PP, D, D, D, D, D, D, D, D, I, P],
    "hhsr_normalize_raw_u16": [P, I, I, I, I, U8P, DP, D, DP, P, P],
    "hhsr_rob_upscale": [P, I, I, P, I, I, I, P, P],
    "hhsr_rob_s": [P, I, I, D, F, F, P, I, I, P],
    "hhsr_rob_sigma": [P, P, I, I, P, I, P, P, P],
    "hhsr_ref_planes": [P, P, I, I, P, I, P, P, P, P],
    "hhsr_rob_frame": [P, I, I, P, P, P, P, I, I, I, P, P, I, D, P, P],
    "hhsr_rob_frames": [PP, I, I, I, P, P, P, PP, I, I, I, PP, D, F, F, P, I, D, PP, I, I, P],
    "hhsr_local_min5": [P, I, I, P, P, P],
    "hhsr_rob_sum": [PP, I, I, I, I, D, P, P, P, P],
    "hhsr_mono_frame_stats": [P, I, I, I, P, P, P, D, D, D, D, D, D, D, D, I, P],
    "hhsr_mono_rob_upscale": [P, I, I, P, I, I, I, P, P],
    "hhsr_mono_rob_sigma": [P, P, I, I, P, I, P, P],
    "hhsr_mono_rob_frame": [P, I, I, P, P, P, I, I, I, P, P, I, D, P, P],
    "hhsr_accumulate": [P, I, I, I, P, I, I, I, P, P, U8P, D, I, P, P, I, I, P],
    "hhsr_accumulate_ref": [P, I, I, I, P, U8P, D, I, P, I, D, D, P, P, I, I, P],
    "hhsr_divide": [P, P, L, P],
    "hhsr_add": [P, P, L, P],
    "hhsr_frame_count_denoise": [P, P, I, I, P, I, I, D, I, D, D, I, P],
    "hhsr_postprocess": [P, P, P, I, I, FP, I, D, P, I, I, I, I, P],
    "hhsr_orient_plane": [P, P, I, I, I, P],
    "hhsr_merge_burst": [PP, PP, PP, PP, I, I, I, I, I, I, I, P, P, U8P, D, I, I, P, P, P, I, I, I, I, I, P],
    "hhsr_clock_probe": [P, L, P],
    "hhsr_merge_burst_chain": [PP, PP, PP, PP, I, I, I, I, I, I, I, P, P, U8P, D, I, I, P, P, P, I, I, P, I, P],
}

MERGE_LOAD_ACC, MERGE_DO_REF, MERGE_DIVIDE, MERGE_STORE_DEN = 1, 2, 4, 8
MERGE_LOCAL_MIN, MERGE_STORE_CLASSES, MERGE_LOAD_CLASSES = 16, 32, 64
MAX_FRAMES = 64
MAX_BATCH = 8  # HHSR_MAX_BATCH: frames per launch of the batched front-end entry points

_lib = None


def exported_symbols():
    """Every entry point include/hhsr.h declares (used by the symbol-export test)."""
    return ["hhsr_version", "hhsr_last_error", "hhsr_merge_chain_bytes", *_SIGS]


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python handheld-multi-frame-super-resolution_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU fallback for the HIP hot path.")
    lib = C.CDLL(LIB_PATH)
    lib.hhsr_version.restype = C.c_char_p
    lib.hhsr_last_error.restype = C.c_char_p
    lib.hhsr_merge_chain_bytes.argtypes = [I, I]
    lib.hhsr_merge_chain_bytes.restype = C.c_size_t
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def version():
    return load().hhsr_version().decode()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = getattr(torch._C, "_cuda_getDevice", None)


def stream(device=None):
    """torch's current HIP stream on `device` (default: the current device) as a hipStream_t.  Called once per kernel
    launch: torch.cuda.current_stream() costs ~7 us (device-index normalisation, availability and environment checks,
    a Stream object) — a fifth of the host time of a burst; the raw-handle getter costs 0.2 us."""
    if _raw_stream is not None and _current_device is not None:
        idx = device.index if (device is not None and getattr(device, "index", None) is not None) else _current_device()
        return C.c_void_p(_raw_stream(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Raw device pointer of a CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("hhsr: expected a tensor on the GPU (the HIP path has no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def call(name, *args):
    lib = _lib if _lib is not None else load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib.hhsr_last_error().decode()}")


def cfa_bytes(cfa):
    flat = [int(v) for row in cfa for v in (row if hasattr(row, "__iter__") else [row])]
    if len(flat) != 4:
        raise ValueError(f"CFA pattern must be 2x2, got {cfa}")
    return (C.c_uint8 * 4)(*flat)


def doubles(vals):
    vals = [float(v) for v in vals]
    return (C.c_double * len(vals))(*vals)


def floats(vals):
    vals = [float(v) for v in vals]
    return (C.c_float * len(vals))(*vals)


def ptr_array(tensors):
    arr = (C.c_void_p * max(1, len(tensors)))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else 0
    return arr


def f32c(t, device=None):
    """float32 contiguous GPU tensor from a tensor / ndarray.  Host data is uploaded on torch's CURRENT stream without
    blocking the host when it is page-locked (pinned tensors / arrays registered by the caller): the frame pipeline
    calls this on the frame's side stream, so the upload of one frame overlaps the kernels of the others (the
    reference uploads synchronously, twice per frame: super_resolution.py:141,145).  Pageable host
    memory goes through the runtime's staging copy — correct, but the host thread waits for it."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    dev = device if device is not None else (t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    if not t.is_cuda and t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        t = t.to(torch.float32)  # (float64 host data too: half the PCIe bytes, the kernels work on float32)
    return t.to(device=dev, non_blocking=True).to(dtype=torch.float32).contiguous()
