"""Oracle: the accumulation (Alg. 4 / Alg. 11) compiled from ``oracle/csrc/merge.c`` with gcc — the same float64 /
float32 operation sequence as ``oracle/merge.py`` pixel by pixel, ~40 x faster than the NumPy form, so that a sweep can
afford TWO oracle runs per burst (its own flows, and the flows of the implementation under test) and a full-size burst
takes minutes instead of an hour.

TEST INFRASTRUCTURE — only ``tests/``, ``__graft_entry__`` and ``bench.py``'s checker legs use it.  ``oracle/merge.py``
stays the form that is pinned against the reference's own outputs (``tests/golden/merge.npz``); this one is pinned
against THAT (``tests/test_oracle_kat.py::test_c_merge_equals_numpy_merge``: every scale family, iso kernels, grey mode,
the accumulated-robustness rules, NaN covariances, frames pushed over the border) and against the same goldens
(``tests/test_oracle_golden.py``).  The only operation that is not bit-identical by construction is ``exp`` (libm here,
NumPy's own SIMD routine there: both within an ulp of float64)."""
import ctypes
import importlib
import os
import subprocess

import numpy as np

_np_merge = importlib.import_module(__package__ + ".merge")  # the MODULE (the package rebinds the name to the function)

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "merge.c")
LIB = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def build(force=False):
    """gcc -O2, no fast-math, no FMA contraction (every operation rounds like the NumPy expression it restates)."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    tmp = LIB + f".{os.getpid()}.tmp"
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
                           "-Wall", "-Wextra", "-o", tmp, SRC, "-lm"])
    os.replace(tmp, LIB)  # atomic: forked workers may race to build
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        p, i64, f64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int
        lib.oracle_merge.argtypes = [p, i64, i64, p, i64, i64, p, i64, i64, p, p, p, i64, i64, p, f64, f64, i32, i32, i64,
                                     i64, i32]
        lib.oracle_merge.restype = None
        lib.oracle_merge_ref.argtypes = [p, i64, i64, p, i64, i64, p, p, i64, i64, p, f64, i32, i32, p, i64, i64, f64, f64,
                                         i64, i64, i64, i32]
        lib.oracle_merge_ref.restype = None
        _lib = lib
    return _lib


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data


def _inplace(a):
    return a.dtype == F32 and a.flags.c_contiguous and a.flags.writeable


def merge(comp, flow, covs, r, num, den, cfa, config, threads=1):
    """Same contract and result as ``oracle.merge.merge`` (num / den updated in place)."""
    if not (_inplace(num) and _inplace(den)):
        return _np_merge.merge(comp, flow, covs, r, num, den, cfa, config)
    lib = load()
    comp, pc = _c(comp, F32)
    flow, pf = _c(flow, F32)
    covs, pk = _c(covs, F32)
    r, pr = _c(r, F32)
    cfa, pa = _c(np.asarray(cfa), np.int64)
    hr_h, hr_w, _ = num.shape
    lib.oracle_merge(pc, comp.shape[0], comp.shape[1], pf, flow.shape[0], flow.shape[1], pk, covs.shape[0], covs.shape[1],
                     pr, num.ctypes.data, den.ctypes.data, hr_h, hr_w, pa, float(config.scale),
                     float(config.block_matching.tuning.tile_size), int(config.mode == "bayer"),
                     int(config.merging.kernel == "iso"), 0, hr_h, int(threads))


def merge_ref(ref, covs, num, den, cfa, config, acc_rob=None, threads=1):
    """Same contract and result as ``oracle.merge.merge_ref``."""
    denoise = bool(config.accumulated_robustness_denoiser.enabled)
    if not (_inplace(num) and _inplace(den)) or (denoise and np.asarray(acc_rob).dtype != np.float64):
        return _np_merge.merge_ref(ref, covs, num, den, cfa, config, acc_rob)
    lib = load()
    ref, pr = _c(ref, F32)
    covs, pk = _c(covs, F32)
    cfa, pa = _c(np.asarray(cfa), np.int64)
    oh, ow, _ = num.shape
    pacc, arh, arw, mfc, mm, rad = None, 0, 0, 0.0, 1.0, 1
    if denoise:
        d = config.accumulated_robustness_denoiser.merge
        acc, pacc = _c(acc_rob, np.float64)
        arh, arw = acc.shape
        mfc, mm, rad = float(d.max_frame_count), float(d.max_multiplier), int(d.rad_max)
    lib.oracle_merge_ref(pr, ref.shape[0], ref.shape[1], pk, covs.shape[0], covs.shape[1], num.ctypes.data, den.ctypes.data,
                         oh, ow, pa, float(config.scale), int(config.mode == "bayer"), int(config.merging.kernel == "iso"),
                         pacc, arh, arw, mfc, mm, rad, 0, oh, int(threads))
