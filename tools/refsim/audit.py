"""Type audit of the reference kernels as tools/refsim executes them (HHSR_REFSIM_AUDIT=1).

TEST INFRASTRUCTURE ONLY.  The goldens under tests/golden/ were produced by running the reference's @cuda.jit kernel bodies
as plain Python on NumPy scalars (loader.py, fake_numba.py): NumPy-2 scalar promotion stands in for Numba's typing.  This
module records, while the goldens are regenerated, every CLASS of typed operation the kernels execute —

    op      (operator, type of the left operand, type of the right operand) -> type of the result
    call    max / min / abs with the types of the arguments -> type of the result
    var     (kernel, local variable) -> the set of types assigned to it during one launch configuration

— with counts, and writes them to tests/golden/typing_audit.json.  tests/test_refsim_typing.py holds Numba's rules (with the
Numba source location of each) and fails on any recorded class the table does not cover or types differently.  The values
pass through unchanged: the goldens of an audited run are bit-identical to those of a plain run (make_goldens checks).
"""
import collections
import json
import operator
import os

import numpy as np

OPS = {"add": operator.add, "sub": operator.sub, "mul": operator.mul, "truediv": operator.truediv,
       "floordiv": operator.floordiv, "mod": operator.mod, "pow": operator.pow, "lshift": operator.lshift,
       "rshift": operator.rshift, "and": operator.and_, "or": operator.or_, "xor": operator.xor}
ops = collections.Counter()    # (op, lhs, rhs, result) -> count
calls = collections.Counter()  # (name, (arg types...), result) -> count
varts = collections.defaultdict(collections.Counter)  # (kernel, variable) -> {type: count}


def tname(v):
    if isinstance(v, np.generic):
        return v.dtype.name                      # float32, float64, int64, bool, uint8 ...
    if isinstance(v, np.ndarray):
        return f"array[{v.dtype.name}]"
    if isinstance(v, bool):
        return "pybool"
    if isinstance(v, int):
        return "pyint"
    if isinstance(v, float):
        return "pyfloat"
    if isinstance(v, tuple):
        return "tuple"
    return type(v).__name__


def op(name, a, b):
    r = OPS[name[1:] if name[0] == "i" and name[1:] in OPS else name](a, b)
    ops[(name[1:] if name[0] == "i" and name[1:] in OPS else name, tname(a), tname(b), tname(r))] += 1
    return r


def neg(a):
    r = -a
    ops[("neg", tname(a), "-", tname(r))] += 1
    return r


call_sites = collections.defaultdict(set)  # (name, (arg types...), result) -> {"kernel:line", ...}


def call(name, site, *args):
    r = {"max": max, "min": min, "abs": abs}[name](*args)
    k = (name, tuple(tname(a) for a in args), tname(r))
    calls[k] += 1
    call_sites[k].add(site)
    return r


def assign(kernel, var, value):
    varts[(kernel, var)][tname(value)] += 1
    return value


def dump(path):
    rec = {
        "ops": sorted([list(k) + [n] for k, n in ops.items()]),
        "calls": sorted([[k[0], list(k[1]), k[2], n, sorted(call_sites[k])] for k, n in calls.items()]),
        "vars": sorted([[k[0], k[1], sorted(v)] for k, v in varts.items() if len(v) > 1]),
        "vars_single_type": sum(1 for v in varts.values() if len(v) == 1),
        "note": "classes of typed operations the reference's kernels executed while tools/refsim/make_goldens.py regenerated "
                "every fixture (HHSR_REFSIM_AUDIT=1): [operator, lhs, rhs, result, count]; [call, [args], result, count]; "
                "[kernel, variable, types] for local variables that were assigned more than one type",
    }
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(rec, f, indent=1)
    return rec
