"""A CPU stand-in for the slice of ``numba.cuda`` the reference's kernels use.

TEST INFRASTRUCTURE ONLY.  This lets the reference's own ``@cuda.jit`` kernel
*bodies* (plain Python) execute in this GPU-less, numba-less container so that
golden vectors can be captured from the reference itself (SURVEY.md §8c,
App. E).  It runs only where ``/root/reference`` exists; nothing here ships to
the GPU box except the small ``.npz`` fixtures it produced.

Execution model
---------------
* One OS thread.  A kernel launch iterates blocks × threads.
* Kernels whose source calls ``cuda.syncthreads`` / ``cuda.shfl_down_sync``
  are rewritten by ``loader.py`` into *generators*: ``syncthreads()`` becomes
  ``yield ("sync",)`` and ``shfl_down_sync(m, v, o)`` becomes
  ``(yield ("shfl", v, o))``.  The launcher steps all threads of a block
  round-robin from barrier to barrier, which gives CUDA barrier semantics
  deterministically.
* ``cuda.shared.array`` returns the same buffer to every thread of a block
  (keyed by per-thread call order), ``cuda.local.array`` a private one.
* Python ``float`` kernel arguments are passed as ``np.float64`` and float
  literals inside kernels are wrapped to ``np.float64`` by the loader, so that
  NumPy-2 promotion (float32 ⊕ float64 → float64) mirrors Numba's typing
  (SURVEY.md App. B).
"""
from __future__ import annotations

import inspect
import sys
import types

import numpy as np

try:  # torch tensors are accepted as kernel arguments
    import torch
except Exception:  # pragma: no cover
    torch = None

WARP = 32


class DeviceArray(np.ndarray):
    """ndarray view with the two device-array methods the reference calls."""

    def copy_to_host(self):
        return np.array(self)

    def copy_to_device(self, src):
        self[...] = np.asarray(src)


def _as_dev(a):
    return np.asarray(a).view(DeviceArray)


class _Ctx:
    __slots__ = ("tid3", "bid3", "bdim3", "gdim3", "shared_idx", "block", "lin")


_cur = _Ctx()


class _Block:
    def __init__(self):
        self.shared = []


class _Dim:
    def __init__(self, which):
        self._w = which

    @property
    def x(self):
        return getattr(_cur, self._w)[0]

    @property
    def y(self):
        return getattr(_cur, self._w)[1]

    @property
    def z(self):
        return getattr(_cur, self._w)[2]


def _grid(ndim):
    v = tuple(_cur.bid3[i] * _cur.bdim3[i] + _cur.tid3[i] for i in range(ndim))
    return v[0] if ndim == 1 else v


class _Shared:
    @staticmethod
    def array(shape, dtype):
        k = _cur.shared_idx
        _cur.shared_idx = k + 1
        blk = _cur.block
        if k == len(blk.shared):
            blk.shared.append(np.zeros(shape, dtype=dtype))
        return blk.shared[k]


class _Local:
    @staticmethod
    def array(shape, dtype):
        return np.zeros(shape, dtype=dtype)


def _conv_arg(a):
    if torch is not None and isinstance(a, torch.Tensor):
        return a.detach().numpy()
    if isinstance(a, bool):
        return a
    if isinstance(a, float):
        return np.float64(a)
    if isinstance(a, int):
        return np.int64(a)
    return a


def _t3(t):
    if isinstance(t, (int, np.integer)):
        t = (int(t),)
    t = tuple(int(v) for v in t)
    return t + (1,) * (3 - len(t))


class _Launcher:
    def __init__(self, fn, grid, block):
        self.fn, self.grid, self.block = fn, _t3(grid), _t3(block)

    def __call__(self, *args):
        args = [_conv_arg(a) for a in args]
        fn = self.fn
        gx, gy, gz = self.grid
        bx, by, bz = self.block
        is_gen = inspect.isgeneratorfunction(fn)
        _cur.bdim3 = self.block
        _cur.gdim3 = self.grid
        I = np.int64
        tids = [(I(tx), I(ty), I(tz)) for tz in range(bz) for ty in range(by) for tx in range(bx)]
        _cur.bdim3 = tuple(I(v) for v in self.block)
        _cur.gdim3 = tuple(I(v) for v in self.grid)
        for bzz in range(gz):
            for byy in range(gy):
                for bxx in range(gx):
                    _cur.bid3 = (I(bxx), I(byy), I(bzz))
                    _cur.block = _Block()
                    if not is_gen:
                        for t in tids:
                            _cur.tid3 = t
                            _cur.shared_idx = 0
                            fn(*args)
                    else:
                        self._run_block_gen(fn, args, tids)

    @staticmethod
    def _run_block_gen(fn, args, tids):
        n = len(tids)
        gens = [None] * n
        sidx = [0] * n
        alive = [True] * n
        pending = [None] * n  # value to send on next resume
        for i, t in enumerate(tids):
            _cur.tid3 = t
            _cur.shared_idx = 0
            gens[i] = fn(*args)
        first = True
        while True:
            reqs = [None] * n
            any_alive = False
            for i, t in enumerate(tids):
                if not alive[i]:
                    continue
                _cur.tid3 = t
                _cur.shared_idx = sidx[i]
                try:
                    if first:
                        reqs[i] = next(gens[i])
                    else:
                        reqs[i] = gens[i].send(pending[i])
                    any_alive = True
                except StopIteration:
                    alive[i] = False
                sidx[i] = _cur.shared_idx
            first = False
            if not any_alive:
                return
            # resolve the collective each live thread is parked on
            for i in range(n):
                pending[i] = None
            for i in range(n):
                r = reqs[i]
                if r is None or r[0] == "sync":
                    continue
                # ("shfl", value, offset): lane i reads lane i+offset of its warp
                _, val, off = r
                lane = i % WARP
                j = i + off
                if lane + off < WARP and j < n and reqs[j] is not None and reqs[j][0] == "shfl":
                    pending[i] = reqs[j][1]
                else:
                    pending[i] = val


class _Kernel:
    def __init__(self, fn):
        self.fn = fn
        self.__name__ = getattr(fn, "__name__", "kernel")

    def __getitem__(self, cfg):
        grid, block = cfg[0], cfg[1]
        return _Launcher(self.fn, grid, block)


def _jit(*a, **kw):
    device = kw.get("device", False)
    if a and callable(a[0]) and len(a) == 1 and not kw:
        return _Kernel(a[0])

    def deco(fn):
        return fn if device else _Kernel(fn)

    return deco


class _Stream:
    def synchronize(self):
        pass


def _to_device(obj, to=None, stream=None, copy=True):
    if to is not None:
        to[...] = np.asarray(obj)
        return to
    return _as_dev(np.array(obj))


def _device_array(shape, dtype=np.float64, stream=None):
    return _as_dev(np.zeros(shape, dtype=dtype))


def _device_array_like(a, stream=None):
    return _as_dev(np.zeros(np.shape(a), dtype=np.asarray(a).dtype))


def _as_cuda_array(t):
    if torch is not None and isinstance(t, torch.Tensor):
        return _as_dev(t.detach().numpy())
    return _as_dev(t)


def install():
    """Register fake ``numba`` and ``numba.cuda`` modules in ``sys.modules``."""
    numba = types.ModuleType("numba")
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = _jit
    cuda.grid = _grid
    cuda.threadIdx = _Dim("tid3")
    cuda.blockIdx = _Dim("bid3")
    cuda.blockDim = _Dim("bdim3")
    cuda.gridDim = _Dim("gdim3")
    cuda.shared = _Shared
    cuda.local = _Local
    cuda.to_device = _to_device
    cuda.device_array = _device_array
    cuda.device_array_like = _device_array_like
    cuda.as_cuda_array = _as_cuda_array
    cuda.synchronize = lambda: None
    cuda.stream = lambda: _Stream()
    # these two only exist un-rewritten in kernels the goldens never run
    cuda.syncthreads = lambda: (_ for _ in ()).throw(RuntimeError("syncthreads outside generator kernel"))
    cuda.shfl_down_sync = lambda *a: (_ for _ in ()).throw(RuntimeError("shfl outside generator kernel"))
    numba.cuda = cuda
    numba.float32 = np.float32
    numba.float64 = np.float64
    numba.complex64 = np.complex64
    numba.uint8 = np.uint8
    numba.int32 = np.int32
    numba.int64 = np.int64
    sys.modules["numba"] = numba
    sys.modules["numba.cuda"] = cuda
    return numba
