"""dtype constants, divide / add and the verbose timers (reference utils.py:16-146)."""
import time

import numpy as np
import torch

from . import _lib

DEFAULT_NUMPY_FLOAT_TYPE = np.float32
DEFAULT_TORCH_FLOAT_TYPE = torch.float32
DEFAULT_TORCH_COMPLEX_TYPE = torch.complex64
EPSILON_DIV = 1e-10
DEFAULT_THREADS = 16  # kept for API parity; the HIP kernels choose their own 256-thread workgroups


def getTime(currentTime, labelName, printTime=True, spaceSize=50):
    """Print the elapsed time since currentTime; return the new current time (utils.py:26-30)."""
    if printTime:
        print(labelName, " " * (spaceSize - len(labelName)), ": ", round((time.perf_counter() - currentTime) * 1000, 2), "milliseconds")
    return time.perf_counter()


def divide(num, den):
    """num = num / den in place (utils.py:62-90).  0/0 stays NaN like the reference."""
    assert num.shape == den.shape
    _lib.call("hhsr_divide", _lib.ptr(num), _lib.ptr(den), num.numel(), _lib.stream())


def add(A, B):
    """A += B for 2-D arrays (utils.py:92-120); the accumulated robustness is float32 here."""
    assert A.shape == B.shape
    _lib.call("hhsr_add", _lib.ptr(A), _lib.ptr(B), A.numel(), _lib.stream())


def round_iso(iso):
    import math

    return int(100 * (2 ** round(math.log2(iso / 100))))


def timer(func, enabled, start_s=None, end_s=None, spaceSize=50):
    """Synchronising wall-clock wrapper gated by config.verbose (utils.py:128-146)."""
    if not enabled:
        return func

    def wrapper(*args, **kwargs):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if start_s is not None:
            print(start_s)
        out = func(*args, **kwargs)
        torch.cuda.synchronize()
        if end_s is not None:
            print(end_s, " " * (spaceSize - len(end_s)), ": ", round((time.perf_counter() - t1) * 1000, 2), "milliseconds")
        return out

    return wrapper
