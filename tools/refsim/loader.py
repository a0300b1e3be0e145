"""Load the reference package from /root/reference on top of ``fake_numba``.

TEST INFRASTRUCTURE ONLY (see fake_numba.py).  Nothing is copied: the
reference's sources are read where they lie, rewritten *in memory* and
executed:

* the literal device string ``"cuda"`` → ``"cpu"`` (torch tensors live on the
  host here);
* ``1/0`` (used as +inf in the robustness kernels, reference
  robustness.py:390,582-585,678) → ``float("inf")``;
* inside ``@cuda.jit`` functions: float literals are wrapped in ``np.float64``
  (Numba types them float64), ``cuda.syncthreads()`` → ``yield ("sync",)`` and
  ``cuda.shfl_down_sync(m, v, o)`` → ``(yield ("shfl", v, o))``;
* inside ``@cuda.jit`` functions int literals, ``int()``, ``round()`` and
  ``range()`` produce ``np.int64`` (Numba: int64; int64 (+) float32 -> float64),
  ``math.*`` returns float32 for float32 input and float64 otherwise,
  and ``round`` saturates non-finite input to 0 (the reference rounds +inf at
  robustness.py:519, SURVEY.md App. A D6 — an out-of-bounds curve read on real
  hardware whose result is discarded).

``handheld_super_resolution/__init__.py`` is bypassed (it imports rawpy /
omegaconf through ``super_resolution``); third-party modules that are absent
from this image and unused on the hot path are stubbed.
"""
from __future__ import annotations

import ast
import importlib.abc
import importlib.util
import os
import sys
import types

import numpy as np

from . import fake_numba

REF_ROOT = os.environ.get("HHSR_REFERENCE", "/root/reference")
PKG = "handheld_super_resolution"


def _is_cuda_jit(dec):
    node = dec.func if isinstance(dec, ast.Call) else dec
    return isinstance(node, ast.Attribute) and node.attr == "jit"


def _is_cuda_call(node, name):
    return (
        isinstance(node, ast.Call)
        and isinstance(node.func, ast.Attribute)
        and node.func.attr == name
        and isinstance(node.func.value, ast.Name)
        and node.func.value.id == "cuda"
    )


AUDIT_CALLS = bool(os.environ.get("HHSR_REFSIM_AUDIT"))            # any value: max / min / abs calls, with their call sites
AUDIT = AUDIT_CALLS and os.environ.get("HHSR_REFSIM_AUDIT") != "calls"  # "calls": ONLY those (minutes instead of hours)
_BINOPS = {ast.Add: "add", ast.Sub: "sub", ast.Mult: "mul", ast.Div: "truediv", ast.FloorDiv: "floordiv", ast.Mod: "mod",
           ast.Pow: "pow", ast.LShift: "lshift", ast.RShift: "rshift", ast.BitAnd: "and", ast.BitOr: "or", ast.BitXor: "xor"}


class _KernelRewriter(ast.NodeTransformer):
    """(HHSR_REFSIM_AUDIT=1 additionally routes every arithmetic operator, every max / min / abs call and every assignment
    to a plain local name through tools.refsim.audit, which records the operand and result types and returns the value
    unchanged: tests/test_refsim_typing.py checks the recorded classes against Numba's typing rules.)"""

    def __init__(self, fname="?"):
        self.fname = fname

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if AUDIT and type(node.op) in _BINOPS:
            return ast.copy_location(ast.Call(func=ast.Name(id="__refsim_op", ctx=ast.Load()),
                                              args=[ast.Constant(_BINOPS[type(node.op)]), node.left, node.right], keywords=[]), node)
        return node

    def visit_UnaryOp(self, node):
        self.generic_visit(node)
        if AUDIT and isinstance(node.op, ast.USub):
            return ast.copy_location(ast.Call(func=ast.Name(id="__refsim_neg", ctx=ast.Load()), args=[node.operand], keywords=[]), node)
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if AUDIT and type(node.op) in _BINOPS:
            import copy

            load = copy.deepcopy(node.target)
            for n in ast.walk(load):
                if hasattr(n, "ctx"):
                    n.ctx = ast.Load()
            val = ast.Call(func=ast.Name(id="__refsim_op", ctx=ast.Load()),
                           args=[ast.Constant("i" + _BINOPS[type(node.op)]), load, node.value], keywords=[])
            if isinstance(node.target, ast.Name):
                val = ast.Call(func=ast.Name(id="__refsim_assign", ctx=ast.Load()),
                               args=[ast.Constant(self.fname), ast.Constant(node.target.id), val], keywords=[])
            return ast.copy_location(ast.Assign(targets=[node.target], value=val), node)
        return node

    def visit_Assign(self, node):
        self.generic_visit(node)
        if AUDIT and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            node.value = ast.Call(func=ast.Name(id="__refsim_assign", ctx=ast.Load()),
                                  args=[ast.Constant(self.fname), ast.Constant(node.targets[0].id), node.value], keywords=[])
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        if AUDIT_CALLS and isinstance(node.func, ast.Name) and node.func.id in ("max", "min", "abs") and not node.keywords:
            site = f"{self.fname}:{getattr(node, 'lineno', 0)}"  # kernel (or device function) and source line of the call
            return ast.copy_location(ast.Call(func=ast.Name(id="__refsim_call", ctx=ast.Load()),
                                              args=[ast.Constant(node.func.id), ast.Constant(site), *node.args], keywords=[]), node)
        if _is_cuda_call(node, "syncthreads"):
            return ast.Yield(value=ast.Tuple(elts=[ast.Constant("sync")], ctx=ast.Load()))
        if _is_cuda_call(node, "shfl_down_sync"):
            return ast.Yield(
                value=ast.Tuple(elts=[ast.Constant("shfl"), node.args[1], node.args[2]], ctx=ast.Load())
            )
        return node

    def visit_Constant(self, node):
        # Numba types float literals float64 and int literals int64 (int64 (+) float32 -> float64)
        if isinstance(node.value, float):
            return ast.Call(func=ast.Name(id="__refsim_f64", ctx=ast.Load()), args=[node], keywords=[])
        if isinstance(node.value, int) and not isinstance(node.value, bool):
            return ast.Call(func=ast.Name(id="__refsim_i64", ctx=ast.Load()), args=[node], keywords=[])
        return node

    def visit_Name(self, node):
        # int()/round()/range() produce int64 values inside kernels, like Numba's
        if isinstance(node.ctx, ast.Load) and node.id in ("int", "round", "range", "math"):
            return ast.copy_location(ast.Name(id="__refsim_" + node.id, ctx=ast.Load()), node)
        return node


class _ModuleRewriter(ast.NodeTransformer):
    def visit_FunctionDef(self, node):
        if any(_is_cuda_jit(d) for d in node.decorator_list):
            decs = node.decorator_list
            node.decorator_list = []
            node = _KernelRewriter(node.name).visit(node)
            node.decorator_list = decs
        return node


def _k_round(x, nd=None):
    """Kernel-side round(): half-to-even, int64 result; tolerates non-finite input (the reference
    rounds +inf at robustness.py:519, D6 — on hardware an out-of-bounds read whose value is discarded)."""
    import builtins

    if nd is None:
        xf = float(x)
        if xf != xf or xf in (float("inf"), float("-inf")):
            return np.int64(0)
        return np.int64(builtins.round(xf))
    return builtins.round(x, nd)


def _k_int(x):
    return np.int64(int(x))


class _KMath:
    """``math`` as Numba types it inside kernels: float32 in -> float32 out, otherwise float64;
    floor/ceil give int64 (plain ``math`` returns weakly-typed Python scalars under NumPy 2)."""

    import math as _m

    @staticmethod
    def _ty(x):
        return np.float32 if isinstance(x, np.float32) else np.float64

    @classmethod
    def _un(cls, name, x):
        t = cls._ty(x)
        try:
            return t(getattr(cls._m, name)(float(x)))
        except (ValueError, OverflowError):  # sqrt(<0) -> nan, exp overflow -> inf like the device
            return t(getattr(np, name)(np.float64(x)))

    @classmethod
    def sqrt(cls, x):
        return cls._un("sqrt", x)

    @classmethod
    def exp(cls, x):
        return cls._un("exp", x)

    @classmethod
    def modf(cls, x):
        t = cls._ty(x)
        f, i = cls._m.modf(float(x))
        return t(f), t(i)

    @classmethod
    def floor(cls, x):
        return np.int64(cls._m.floor(float(x)))

    @classmethod
    def ceil(cls, x):
        return np.int64(cls._m.ceil(float(x)))

    @classmethod
    def copysign(cls, a, b):
        return np.float64(cls._m.copysign(float(a), float(b)))


def _k_range(*a):
    for t in a:  # Numba has no range(float): "No implementation of function Function(<class 'range'>) ... (float64, ...)"
        if isinstance(t, (float, np.floating)):
            raise TypeError(f"range() of a float ({type(t).__name__}) does not type under Numba")
    for v in range(*(int(t) for t in a)):
        yield np.int64(v)


class _RefLoader(importlib.abc.Loader):
    def __init__(self, path, name):
        self.path, self.name = path, name

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        with open(self.path, "r", encoding="utf-8") as f:
            src = f.read()
        src = src.replace('"cuda"', '"cpu"').replace("'cuda'", "'cpu'").replace("1/0", 'float("inf")')
        tree = ast.parse(src, filename=self.path)
        tree = _ModuleRewriter().visit(tree)
        ast.fix_missing_locations(tree)
        module.__dict__["__refsim_f64"] = np.float64
        module.__dict__["__refsim_i64"] = np.int64
        module.__dict__["__refsim_int"] = _k_int
        module.__dict__["__refsim_round"] = _k_round
        module.__dict__["__refsim_range"] = _k_range
        module.__dict__["__refsim_math"] = _KMath
        if AUDIT_CALLS:
            from . import audit

            module.__dict__["__refsim_op"] = audit.op
            module.__dict__["__refsim_neg"] = audit.neg
            module.__dict__["__refsim_call"] = audit.call
            module.__dict__["__refsim_assign"] = audit.assign
        module.__dict__["__file__"] = self.path
        exec(compile(tree, self.path, "exec"), module.__dict__)


class _RefFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(PKG + "."):
            return None
        leaf = fullname.split(".", 1)[1]
        p = os.path.join(REF_ROOT, PKG, leaf + ".py")
        if not os.path.exists(p):
            return None
        return importlib.util.spec_from_loader(fullname, _RefLoader(p, fullname))


class _AnyModule(types.ModuleType):
    """Stub for a module that is absent here and unused on the hot path: any attribute resolves."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return None


def _stub(name, **attrs):
    m = _AnyModule(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load_config_module():
    here = os.path.dirname(os.path.abspath(__file__))
    p = os.path.join(here, "..", "..", "handheld-multi-frame-super-resolution_amd", "handheld_super_resolution", "config.py")
    spec = importlib.util.spec_from_file_location("_refsim_hhsr_config", os.path.abspath(p))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_refsim_hhsr_config"] = mod
    spec.loader.exec_module(mod)
    return mod


_installed = False
cfgmod = None


def install():
    """Make ``import handheld_super_resolution.<module>`` resolve to the reference."""
    global _installed, cfgmod
    if _installed:
        return cfgmod
    if not os.path.isdir(os.path.join(REF_ROOT, PKG)):
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    fake_numba.install()
    cfgmod = _load_config_module()
    # modules absent from this image; none is used by main()
    _stub("omegaconf", OmegaConf=cfgmod.OmegaConf, DictConfig=cfgmod.Config)
    for name in ("rawpy", "exifread", "cv2", "imageio", "colour_demosaicing"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    for name in ("skimage", "skimage.filters", "skimage.transform", "skimage.color", "skimage.exposure"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name, gaussian=None, resize=None)
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        _stub("matplotlib")
        _stub("matplotlib.pyplot")
    pkg = types.ModuleType(PKG)
    pkg.__path__ = [os.path.join(REF_ROOT, PKG)]
    pkg.__package__ = PKG
    # a stale product package of the same name must not shadow the reference here
    for k in [k for k in sys.modules if k == PKG or k.startswith(PKG + ".")]:
        del sys.modules[k]
    sys.modules[PKG] = pkg
    sys.meta_path.insert(0, _RefFinder())
    _installed = True
    return cfgmod


def ref(module):
    """Import one reference module, e.g. ``ref('merge')``."""
    install()
    return importlib.import_module(f"{PKG}.{module}")
